// membench4.hip -- does the SHAPE of a store instruction matter?  Every kernel writes the same 8 GiB in the same
// order of 4 KB chunks (8 rows of 512 B, like one 8-frame chunk of the M=64 channelizer output); a workgroup of 8
// waves covers 8 runs of 8 chunks (256 KB), workgroups in dispatch order.  What differs is how a wave's eight
// 8-byte-per-lane store instructions tile its 4 KB chunk:
//   A  8 segments of  64 B per instruction (lane = (row, 8 B slot); instruction k = 64-byte column k)   <- the kernel today
//   B  4 segments of 128 B per instruction (lane = (row pair member, 16 slots))
//   C  1 segment  of 512 B per instruction (instruction k = row k)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int MODE>
__global__ void __launch_bounds__(512) k_store(float2* out, long long chunks) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long run = (long long)blockIdx.x * 8 + wave;
  for (int c = 0; c < 8; ++c) {
    const long long chunk = run * 8 + c;
    if (chunk >= chunks) return;
    float2* base = out + chunk * 512;  // 4 KB = 512 float2
    const float2 v = make_float2((float)lane, (float)c);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      long long idx;
      if (MODE == 0) idx = (lane >> 3) * 64 + k * 8 + (lane & 7);                          // row lane/8, column block k
      else if (MODE == 1) idx = ((lane >> 4) + 4 * (k >> 2)) * 64 + (k & 3) * 16 + (lane & 15);  // rows {l/16, l/16+4}, 128-B blocks
      else idx = k * 64 + lane;                                                           // row k
      base[idx] = v;
    }
  }
}

int main() {
  const long long bytes = 8ll << 30, chunks = bytes / 4096;
  float2* out;
  CK(hipMalloc(&out, bytes));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const unsigned grid = (unsigned)((chunks + 63) / 64);
  const char* names[3] = {"A 8 x 64 B per instruction", "B 4 x 128 B per instruction", "C 1 x 512 B per instruction"};
  for (int rep = 0; rep < 2; ++rep)
    for (int m = 0; m < 3; ++m) {
      std::vector<float> t;
      for (int i = 0; i < 12; ++i) {
        CK(hipEventRecord(a));
        if (m == 0) hipLaunchKernelGGL(k_store<0>, dim3(grid), dim3(512), 0, 0, out, chunks);
        else if (m == 1) hipLaunchKernelGGL(k_store<1>, dim3(grid), dim3(512), 0, 0, out, chunks);
        else hipLaunchKernelGGL(k_store<2>, dim3(grid), dim3(512), 0, 0, out, chunks);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (i >= 4) t.push_back(ms);
      }
      std::sort(t.begin(), t.end());
      printf("%-30s min %.3f med %.3f ms  %.1f GB/s\n", names[m], t[0], t[t.size() / 2], bytes / (t[t.size() / 2] * 1e-3) / 1e9);
    }
  return 0;
}
