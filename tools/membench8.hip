// tools/membench8.hip -- what a SCREEN-based PDW bracket pass could cost (route 3): F x M uint16 screens
// (upper 16 bits of |y|^2) read 8 bytes per lane (four channels), two limits per channel; a sample whose screen falls between
// the limits (about `frac` of them) is fetched from the F x M complex64 matrix (8 bytes, scattered) and counted.  No staging,
// no float64: the bare access shape, like membench7 for the pass that exists.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/membench8 tools/membench8.hip ; run: tools/membench8 [log2 frames] [M]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int ROWS>
__global__ void __launch_bounds__(256) screen_pass(const uint2* screen, const float2* y, long long F, int M, unsigned lo, unsigned hi,
                                                   unsigned long long* out) {
  const int lanes_per_row = M / 4;                       // 4 channels per lane
  const int rows_per_wave = 64 / lanes_per_row;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / lanes_per_row, c4 = lane % lanes_per_row;
  const long long rows_per_block = (long long)ROWS * rows_per_wave * 4;
  const long long r_base = (long long)blockIdx.x * rows_per_block + (long long)wave * ROWS * rows_per_wave;
  unsigned long long acc = 0;
  uint2 v[ROWS];
#pragma unroll
  for (int i = 0; i < ROWS; ++i) {
    const long long r = r_base + (long long)i * rows_per_wave + sub;
    v[i] = r < F ? screen[r * lanes_per_row + c4] : make_uint2(0, 0);
  }
#pragma unroll
  for (int i = 0; i < ROWS; ++i) {
    const long long r = r_base + (long long)i * rows_per_wave + sub;
    const unsigned s[4] = {v[i].x & 0xffffu, v[i].x >> 16, v[i].y & 0xffffu, v[i].y >> 16};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      acc += s[c] < lo;
      if (s[c] >= lo && s[c] <= hi && r < F) {  // undecided: the exact sample
        const float2 z = y[r * M + c4 * 4 + c];
        acc += (unsigned long long)(z.x * z.x + z.y * z.y > 1.0f) << 32;
      }
    }
  }
  if (acc == 0x123456789abcdefull) out[0] = acc;  // keep everything alive
  if ((acc & 0xffffffffull) == 0x7fffffffull) out[1] = acc;
}

int main(int argc, char** argv) {
  const int l2 = argc > 1 ? atoi(argv[1]) : 22, M = argc > 2 ? atoi(argv[2]) : 128;
  const long long F = 1ll << l2, n = F * M;
  uint16_t* d_s; float2* d_y; unsigned long long* d_o;
  CK(hipMalloc(&d_s, n * 2)); CK(hipMalloc(&d_y, n * 8)); CK(hipMalloc(&d_o, 128));
  std::vector<uint16_t> h(n);
  unsigned x = 12345u;
  for (long long i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; h[i] = (uint16_t)(x >> 16); }
  CK(hipMemcpy(d_s, h.data(), n * 2, hipMemcpyHostToDevice));
  CK(hipMemset(d_y, 0, n * 8)); CK(hipMemset(d_o, 0, 128));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("# screen pass over %lld x %d (screen %.2f GB, matrix %.2f GB)\n", F, M, n * 2 / 1e9, n * 8 / 1e9);
  for (double frac : {0.0, 0.02, 0.04, 0.08}) {
    const unsigned lo = 30000u, hi = lo + (unsigned)(frac * 65536.0);
    const int rows_per_wave = 64 / (M / 4);
    constexpr int ROWS = 16;
    const long long rows_per_block = (long long)ROWS * rows_per_wave * 4;
    const unsigned grid = (unsigned)((F + rows_per_block - 1) / rows_per_block);
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(screen_pass<ROWS>, dim3(grid), dim3(256), 0, 0, (const uint2*)d_s, d_y, F, M, lo, frac > 0 ? hi : 0u, d_o);
    CK(hipEventRecord(e0));
    for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(screen_pass<ROWS>, dim3(grid), dim3(256), 0, 0, (const uint2*)d_s, d_y, F, M, lo, frac > 0 ? hi : 0u, d_o);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("undecided fraction %.2f: %.3f ms per pass\n", frac, ms / 10);
  }
  return 0;
}
