// membench5.hip -- what can a CHANNEL-MAJOR store stream reach?  out[c * F + f] (64 channels, F frames, float2), written
// by workgroups of 8 waves that each own a tile of 512 consecutive frames of all 64 channels (the M=64 channelizer's
// channel-major tile).  Same bytes, same tiles, different shapes and orders of the store instructions:
//   A  wave w owns frames [64w, 64w+64) in 8 chunks of 8 frames; per chunk 8 instructions of 8 channels x 64 B  <- today
//   B  same ownership, but per 64 frames 64 instructions of 1 channel x 512 B (a per-wave LDS transpose would give this)
//   C  wave w owns channels [8w, 8w+8): 8 instructions of 512 B walk one channel's 4 KB run, then the next channel
//      (a workgroup-wide transpose)
//   D  as A with chunks of 16 frames: 4 channels x 128 B per instruction
//   E  as B, but 2 channels x 256 B per instruction (32-frame flushes)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int MODE>
__global__ void __launch_bounds__(512) k_store(float2* out, long long F, long long tiles) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long tile = blockIdx.x;
  if (tile >= tiles) return;
  const long long f0 = tile * 512;
  const float2 v = make_float2((float)lane, (float)wave);
  if (MODE == 0) {
    for (int c = 0; c < 8; ++c) {
      const long long f = f0 + wave * 64 + c * 8 + (lane & 7);
#pragma unroll
      for (int k = 0; k < 8; ++k) out[(long long)(k * 8 + (lane >> 3)) * F + f] = v;
    }
  } else if (MODE == 1) {
    const long long f = f0 + wave * 64 + lane;
#pragma unroll 8
    for (int ch = 0; ch < 64; ++ch) out[(long long)ch * F + f] = v;
  } else if (MODE == 2) {
    for (int j = 0; j < 8; ++j) {
      const int ch = wave * 8 + j;
#pragma unroll
      for (int k = 0; k < 8; ++k) out[(long long)ch * F + f0 + k * 64 + lane] = v;
    }
  } else if (MODE == 3) {
    for (int c = 0; c < 4; ++c) {
      const long long f = f0 + wave * 64 + c * 16 + (lane & 15);
#pragma unroll
      for (int k = 0; k < 16; ++k) out[(long long)(k * 4 + (lane >> 4)) * F + f] = v;
    }
  } else {
    for (int c = 0; c < 2; ++c) {
      const long long f = f0 + wave * 64 + c * 32 + (lane & 31);
#pragma unroll 8
      for (int k = 0; k < 32; ++k) out[(long long)(k * 2 + (lane >> 5)) * F + f] = v;
    }
  }
}

int main(int argc, char** argv) {
  const long long F = argc > 1 ? atoll(argv[1]) : 16000000ll;   // a multiple of 512, not a power of two
  const long long tiles = F / 512, bytes = 64 * F * 8;
  float2* out;
  CK(hipMalloc(&out, bytes));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const char* names[5] = {"A 8 ch x 64 B, chunks of 8 frames", "B 1 ch x 512 B, per wave", "C 1 ch x 512 B, 4 KB runs per wave",
                          "D 4 ch x 128 B, chunks of 16", "E 2 ch x 256 B, chunks of 32"};
  for (int rep = 0; rep < 2; ++rep)
    for (int m = 0; m < 5; ++m) {
      std::vector<float> t;
      for (int i = 0; i < 10; ++i) {
        CK(hipEventRecord(a));
        const dim3 g((unsigned)tiles), blk(512);
        if (m == 0) hipLaunchKernelGGL(k_store<0>, g, blk, 0, 0, out, F, tiles);
        else if (m == 1) hipLaunchKernelGGL(k_store<1>, g, blk, 0, 0, out, F, tiles);
        else if (m == 2) hipLaunchKernelGGL(k_store<2>, g, blk, 0, 0, out, F, tiles);
        else if (m == 3) hipLaunchKernelGGL(k_store<3>, g, blk, 0, 0, out, F, tiles);
        else hipLaunchKernelGGL(k_store<4>, g, blk, 0, 0, out, F, tiles);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (i >= 3) t.push_back(ms);
      }
      std::sort(t.begin(), t.end());
      printf("%-36s min %.3f med %.3f ms  %.1f GB/s\n", names[m], t[0], t[t.size() / 2], bytes / (t[t.size() / 2] * 1e-3) / 1e9);
    }
  return 0;
}
