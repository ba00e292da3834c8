// membench.hip -- HBM streaming yardsticks for the channelizer's traffic mix (1 B read : 2 B written).
// Build: hipcc -O3 --offload-arch=gfx950 tools/membench.hip -o tools/membench ; run on the GPU box.
// Each variant moves `n` 16-byte vectors in and 2n out (or read-only / write-only), timed with hipEvents.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));

// grid-stride, out split in two halves
__global__ void __launch_bounds__(256) k_split(const u4* in, u4* out, long long n) {
  long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    u4 v = in[i]; out[i] = v; out[n + i] = v;
  }
}
// grid-stride, out contiguous 32 B per thread as two 16 B stores (lane stride 32 B)
__global__ void __launch_bounds__(256) k_contig32(const u4* in, u4* out, long long n) {
  long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    u4 v = in[i]; out[2 * i] = v; out[2 * i + 1] = v;
  }
}
// grid-stride, wave writes two fully coalesced 1 KB rows per 1 KB read
__global__ void __launch_bounds__(256) k_rows(const u4* in, u4* out, long long n) {
  long long stride = (long long)gridDim.x * blockDim.x;
  int lane = threadIdx.x & 63;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    u4 v = in[i];
    long long w = i - lane;  // wave base
    out[2 * w + lane] = v; out[2 * w + 64 + lane] = v;
  }
}
// nontemporal versions of k_rows
__global__ void __launch_bounds__(256) k_rows_nt(const u4* in, u4* out, long long n) {
  long long stride = (long long)gridDim.x * blockDim.x;
  int lane = threadIdx.x & 63;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    u4 v = __builtin_nontemporal_load(&in[i]);
    long long w = i - lane;
    __builtin_nontemporal_store(v, &out[2 * w + lane]); __builtin_nontemporal_store(v, &out[2 * w + 64 + lane]);
  }
}
// PFB-like: each wave owns a contiguous run; per step reads 256 B (dword/lane), writes 512 B (8 B/lane);
// unrolled by U steps so U loads are in flight.  n counted in dwords here.
template <int U, bool RUN64>
__global__ void __launch_bounds__(64) k_pfb_like(const unsigned* in, u2* out, long long nd, int steps_per_wave) {
  long long wave = blockIdx.x;
  int lane = threadIdx.x;
  long long base = wave * (long long)steps_per_wave * 64;
  for (int s = 0; s < steps_per_wave; s += U) {
    unsigned v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = in[base + (long long)(s + u) * 64 + lane];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      long long row = base + (long long)(s + u) * 64;
      if (RUN64) {  // 8 runs of 64 B per store instruction, like the {8,8} last pass
        int f = lane >> 3, j = lane & 7;
        // emulate: instruction k (=u) writes frames f of this 8-frame chunk at channel j+8k
        long long chunk = base + (long long)(s) * 64;
        out[chunk + (long long)f * 64 + j + 8 * u] = (u2){v[u], v[u]};
      } else {
        out[row + lane] = (u2){v[u], v[u]};
      }
    }
  }
}
__global__ void __launch_bounds__(256) k_read(const u4* in, u4* out, long long n) {
  long long stride = (long long)gridDim.x * blockDim.x;
  u4 acc = {0, 0, 0, 0};
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) acc ^= in[i];
  if (acc.x == 0x12345678u) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_write(u4* out, long long n) {
  long long stride = (long long)gridDim.x * blockDim.x;
  u4 v = {1, 2, 3, 4};
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = v;
}
__global__ void __launch_bounds__(256) k_copy(const u4* in, u4* out, long long n) {
  long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = in[i];
}

template <class F>
double time_ms(F f, int iters) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int i = 0; i < iters; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / iters;
}

int main(int argc, char** argv) {
  long long bytes_in = (argc > 1 ? atoll(argv[1]) : 4096) * (1ll << 20);  // MiB
  long long n = bytes_in / 16;
  void *in, *out;
  CK(hipMalloc(&in, bytes_in)); CK(hipMalloc(&out, 2 * bytes_in));
  CK(hipMemset(in, 1, bytes_in)); CK(hipMemset(out, 0, 2 * bytes_in));
  const int it = 10;
  auto rep = [&](const char* name, double ms, double bytes) { printf("%-34s %8.3f ms  %8.1f GB/s\n", name, ms, bytes / ms / 1e6); fflush(stdout); };
  for (int g : {1024, 2048, 4096, 8192, 16384}) {
    char nm[64];
    snprintf(nm, 64, "read       grid=%d", g);  rep(nm, time_ms([&] { hipLaunchKernelGGL(k_read, dim3(g), dim3(256), 0, 0, (const u4*)in, (u4*)out, n); }, it), 16.0 * n);
    snprintf(nm, 64, "write(2n)  grid=%d", g);  rep(nm, time_ms([&] { hipLaunchKernelGGL(k_write, dim3(g), dim3(256), 0, 0, (u4*)out, 2 * n); }, it), 32.0 * n);
    snprintf(nm, 64, "copy 1:1   grid=%d", g);  rep(nm, time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, 0, (const u4*)in, (u4*)out, n); }, it), 32.0 * n);
    snprintf(nm, 64, "1:2 split  grid=%d", g);  rep(nm, time_ms([&] { hipLaunchKernelGGL(k_split, dim3(g), dim3(256), 0, 0, (const u4*)in, (u4*)out, n); }, it), 48.0 * n);
    snprintf(nm, 64, "1:2 contig32 grid=%d", g); rep(nm, time_ms([&] { hipLaunchKernelGGL(k_contig32, dim3(g), dim3(256), 0, 0, (const u4*)in, (u4*)out, n); }, it), 48.0 * n);
    snprintf(nm, 64, "1:2 rows   grid=%d", g);  rep(nm, time_ms([&] { hipLaunchKernelGGL(k_rows, dim3(g), dim3(256), 0, 0, (const u4*)in, (u4*)out, n); }, it), 48.0 * n);
    snprintf(nm, 64, "1:2 rows nt grid=%d", g); rep(nm, time_ms([&] { hipLaunchKernelGGL(k_rows_nt, dim3(g), dim3(256), 0, 0, (const u4*)in, (u4*)out, n); }, it), 48.0 * n);
  }
  long long nd = bytes_in / 4;  // dwords
  for (int spw : {512, 2048, 8192}) {
    long long waves = nd / 64 / spw;
    char nm[64];
    snprintf(nm, 64, "pfb-like U=8 spw=%d", spw);
    rep(nm, time_ms([&] { hipLaunchKernelGGL((k_pfb_like<8, false>), dim3((unsigned)waves), dim3(64), 0, 0, (const unsigned*)in, (u2*)out, nd, spw); }, it), 12.0 * nd);
    snprintf(nm, 64, "pfb-like U=8 run64 spw=%d", spw);
    rep(nm, time_ms([&] { hipLaunchKernelGGL((k_pfb_like<8, true>), dim3((unsigned)waves), dim3(64), 0, 0, (const unsigned*)in, (u2*)out, nd, spw); }, it), 12.0 * nd);
    snprintf(nm, 64, "pfb-like U=16 spw=%d", spw);
    rep(nm, time_ms([&] { hipLaunchKernelGGL((k_pfb_like<16, false>), dim3((unsigned)waves), dim3(64), 0, 0, (const unsigned*)in, (u2*)out, nd, spw); }, it), 12.0 * nd);
  }
  return 0;
}
