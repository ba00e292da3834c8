// membench2.hip -- which streaming shape moves the channelizer's 1:2 byte mix fastest?
// Every kernel reads nd dwords (1 dword = one int16 I/Q sample) and writes 2 dwords per dword read.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include <functional>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));

// Wave-run kernel: wave w owns rows [w*spw, (w+1)*spw) of 64 dwords; U rows in flight.
// MODE 0: store 8 B/lane row-contiguous (512 B / instr); MODE 1: 16 B/lane (two rows -> 1 KB / instr)
// NT: nontemporal load+store. REMAP: consecutive runs stay on one XCD (bid%8 picks the eighth of the stream).
template <int U, int MODE, bool NT, bool REMAP, int WPB>
__global__ void __launch_bounds__(64 * WPB) k_run(const unsigned* in, unsigned* out, long long rows_total, int spw) {
  long long nwaves = (long long)gridDim.x * WPB;
  long long wv = (long long)blockIdx.x * WPB + (threadIdx.x >> 6);
  if (REMAP) { long long per = nwaves / 8; long long b = blockIdx.x; long long blk = (b % 8) * (gridDim.x / 8) + b / 8; wv = blk * WPB + (threadIdx.x >> 6); (void)per; }
  int lane = threadIdx.x & 63;
  long long r0 = wv * spw;
  if (r0 >= rows_total) return;
  for (int s = 0; s < spw; s += U) {
    unsigned v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned* p = in + (r0 + s + u) * 64 + lane;
      v[u] = NT ? __builtin_nontemporal_load(p) : *p;
    }
    if (MODE == 0) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        u2* q = (u2*)out + (r0 + s + u) * 64 + lane;
        u2 w = {v[u], v[u] + 1};
        if (NT) __builtin_nontemporal_store(w, q); else *q = w;
      }
    } else {
#pragma unroll
      for (int u = 0; u < U; u += 2) {
        u4* q = (u4*)out + (r0 + s + u) * 32 + lane;  // two rows = 1 KB = 64 lanes x 16 B
        u4 w = {v[u], v[u] + 1, v[u + 1], v[u + 1] + 1};
        if (NT) __builtin_nontemporal_store(w, q); else *q = w;
      }
    }
  }
}

// write-only variants: 16 B/lane, wave-run
template <bool NT>
__global__ void __launch_bounds__(64) k_wrun(u4* out, long long rows16, int spw) {
  long long r0 = (long long)blockIdx.x * spw; int lane = threadIdx.x;
  u4 w = {1, 2, 3, (unsigned)lane};
  for (int s = 0; s < spw; ++s) { u4* q = out + (r0 + s) * 64 + lane; if (NT) __builtin_nontemporal_store(w, q); else *q = w; }
}
__global__ void __launch_bounds__(256) k_wgs(u4* out, long long n) {
  long long stride = (long long)gridDim.x * blockDim.x; u4 v = {1, 2, 3, 4};
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = v;
}

struct Case { std::string name; std::function<void()> run; double bytes; std::vector<double> ms; };

int main(int argc, char** argv) {
  long long mib = argc > 1 ? atoll(argv[1]) : 4096;
  int rounds = argc > 2 ? atoi(argv[2]) : 3;
  long long bytes_in = mib << 20, nd = bytes_in / 4, rows = nd / 64;
  void *in, *out; CK(hipMalloc(&in, bytes_in)); CK(hipMalloc(&out, 2 * bytes_in));
  CK(hipMemset(in, 1, bytes_in)); CK(hipMemset(out, 0, 2 * bytes_in));
  std::vector<Case> cases;
  auto add = [&](std::string n, std::function<void()> f, double bytes) { cases.push_back({n, f, bytes, {}}); };
#define RUN(U, MODE, NT, REMAP, WPB, SPW) add("run U=" #U " mode=" #MODE " nt=" #NT " remap=" #REMAP " wpb=" #WPB " spw=" #SPW, [=] { \
    long long waves = rows / (SPW); hipLaunchKernelGGL((k_run<U, MODE, NT, REMAP, WPB>), dim3((unsigned)(waves / (WPB))), dim3(64 * (WPB)), 0, 0, (const unsigned*)in, (unsigned*)out, rows, SPW); }, 12.0 * nd)
  RUN(8, 0, false, false, 1, 512);
  RUN(8, 0, false, false, 1, 64);
  RUN(8, 0, false, false, 1, 16);
  RUN(8, 0, false, false, 1, 8);
  RUN(8, 1, false, false, 1, 512);
  RUN(8, 1, false, false, 1, 64);
  RUN(8, 1, false, false, 1, 8);
  RUN(8, 1, true, false, 1, 512);
  RUN(8, 1, true, false, 1, 8);
  RUN(8, 0, false, true, 1, 512);
  RUN(8, 0, false, true, 1, 64);
  RUN(8, 1, false, true, 1, 64);
  RUN(8, 1, false, true, 1, 8);
  RUN(8, 0, false, false, 4, 512);
  RUN(8, 0, false, false, 4, 64);
  RUN(8, 1, false, false, 4, 64);
  RUN(8, 1, false, false, 4, 8);
  RUN(16, 1, false, false, 1, 512);
  RUN(16, 1, false, false, 4, 64);
  RUN(4, 1, false, false, 4, 8);
  RUN(2, 1, false, false, 4, 8);
  RUN(2, 1, false, false, 4, 2);
  long long rows16 = (2 * bytes_in / 16) / 64;
  for (int spw : {1, 8, 64, 1024})
    add("write-only wave-run spw=" + std::to_string(spw), [=] { hipLaunchKernelGGL(k_wrun<false>, dim3((unsigned)(rows16 / spw)), dim3(64), 0, 0, (u4*)out, rows16, spw); }, 2.0 * bytes_in);
  add("write-only wave-run nt spw=8", [=] { hipLaunchKernelGGL(k_wrun<true>, dim3((unsigned)(rows16 / 8)), dim3(64), 0, 0, (u4*)out, rows16, 8); }, 2.0 * bytes_in);
  for (int g : {2048, 16384, 65536})
    add("write-only grid-stride g=" + std::to_string(g), [=] { hipLaunchKernelGGL(k_wgs, dim3(g), dim3(256), 0, 0, (u4*)out, (long long)(2 * bytes_in / 16)); }, 2.0 * bytes_in);

  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (auto& c : cases) { c.run(); }
  CK(hipDeviceSynchronize());
  for (int r = 0; r < rounds; ++r)
    for (auto& c : cases) {
      CK(hipEventRecord(a)); for (int i = 0; i < 5; ++i) c.run(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); c.ms.push_back(ms / 5);
    }
  CK(hipGetLastError());
  for (auto& c : cases) {
    std::sort(c.ms.begin(), c.ms.end());
    printf("%-62s min %7.3f med %7.3f ms  best %7.1f GB/s\n", c.name.c_str(), c.ms.front(), c.ms[c.ms.size() / 2], c.bytes / c.ms.front() / 1e6);
  }
  return 0;
}
