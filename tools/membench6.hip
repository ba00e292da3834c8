// membench6.hip -- what does MI355X deliver for the OTHER byte mixes of the channelizer?  1 B read : 4 B written (cfg3:
// int8 I/Q in, complex64 out; cfg5: int16 in, two frames out per D samples) and 1 : 2 (cfg2, cfg4), each in the two
// access shapes that matter: short-lived workgroups sweeping the buffers in dispatch order (tools/membench2's winner) and
// one long run per wave.  Reads one dword per lane, writes RATIO x 4 bytes per lane and dword read (16-byte stores).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// wave w owns rows [w*spw, (w+1)*spw) of 64 dwords (256 B); per row it writes RATIO rows of 64 dwords
template <int RATIO>
__global__ void __launch_bounds__(256) k(const unsigned* in, uint4* out, long long rows, int spw) {
  const long long wv = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const long long r0 = wv * spw;
  if (r0 >= rows) return;
  for (int s = 0; s < spw; s += 4) {   // 4 rows = 1 KB in flight per wave
    unsigned v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = in[(r0 + s + u) * 64 + lane];
    // 4 rows in -> 4*RATIO rows out = RATIO KB: RATIO 16-byte stores per lane, each instruction 1 KB contiguous
    uint4* o = out + (r0 + s) * 16 * RATIO + lane;
#pragma unroll
    for (int j = 0; j < RATIO; ++j) o[j * 64] = make_uint4(v[0] + j, v[1], v[2], v[3]);
  }
}

// U rows in flight per wave, one 16-byte store per lane per (row group, j): U = 1 writes 4 dwords of ONE input dword
template <int RATIO, int U>
__global__ void __launch_bounds__(256) ku(const unsigned* in, uint4* out, long long rows, int spw) {
  const long long wv = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const long long r0 = wv * spw;
  if (r0 >= rows) return;
  for (int s = 0; s < spw; s += U) {
    unsigned v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = in[(r0 + s + u) * 64 + lane];
    uint4* o = out + (r0 + s) * 16 * RATIO + lane;   // U rows in -> U * RATIO * 256 B out = U * RATIO / 4 instructions of 1 KB
#pragma unroll
    for (int j = 0; j < U * RATIO / 4; ++j) o[j * 64] = make_uint4(v[0] + j, v[U > 1 ? 1 : 0], v[U > 2 ? 2 : 0], v[U > 3 ? 3 : 0]);
  }
}

template <int RATIO, int U>
void runu(const char* name, const unsigned* in, uint4* out, long long rows, int spw, double bytes) {
  const long long waves = rows / spw;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  std::vector<float> ms;
  for (int r = 0; r < 6; ++r) {
    CK(hipEventRecord(a));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((ku<RATIO, U>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, 0, in, out, rows, spw);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float t; CK(hipEventElapsedTime(&t, a, b)); if (r) ms.push_back(t / 5);
  }
  std::sort(ms.begin(), ms.end());
  printf("%-36s U=%d spw=%5d  min %7.3f med %7.3f ms  best %7.1f GB/s = %.3f of 8 TB/s\n", name, U, spw, ms.front(), ms[ms.size() / 2],
         bytes / ms.front() / 1e6, bytes / ms.front() / 1e6 / 8000.0);
}

template <int RATIO>
void run(const char* name, const unsigned* in, uint4* out, long long rows, int spw, double bytes) {
  const long long waves = rows / spw;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  std::vector<float> ms;
  for (int r = 0; r < 6; ++r) {
    CK(hipEventRecord(a));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<RATIO>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, 0, in, out, rows, spw);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float t; CK(hipEventElapsedTime(&t, a, b)); if (r) ms.push_back(t / 5);
  }
  std::sort(ms.begin(), ms.end());
  printf("%-44s spw=%5d  min %7.3f med %7.3f ms  best %7.1f GB/s = %.3f of 8 TB/s\n", name, spw, ms.front(), ms[ms.size() / 2],
         bytes / ms.front() / 1e6, bytes / ms.front() / 1e6 / 8000.0);
}

int main() {
  const long long in_bytes = 2ll << 30, rows = in_bytes / 256;   // 2 GiB in; up to 8 GiB out
  void *in, *out; CK(hipMalloc(&in, in_bytes)); CK(hipMalloc(&out, in_bytes * 4));
  CK(hipMemset(in, 1, in_bytes)); CK(hipMemset(out, 0, in_bytes * 4));
  for (int spw : {4, 8, 32, 512}) run<4>("1 read : 4 written (cfg3, cfg5)", (const unsigned*)in, (uint4*)out, rows, spw, 5.0 * in_bytes);
  for (int spw : {4, 8, 32, 512}) run<2>("1 read : 2 written (cfg2, cfg4)", (const unsigned*)in, (uint4*)out, rows, spw, 3.0 * in_bytes);
  for (int spw : {1, 2, 4}) runu<4, 1>("1:4, one row per step", (const unsigned*)in, (uint4*)out, rows, spw, 5.0 * in_bytes);
  for (int spw : {2, 4, 8}) runu<4, 2>("1:4, two rows per step", (const unsigned*)in, (uint4*)out, rows, spw, 5.0 * in_bytes);
  for (int spw : {2, 4, 8}) runu<2, 2>("1:2, two rows per step", (const unsigned*)in, (uint4*)out, rows, spw, 3.0 * in_bytes);
  return 0;
}
