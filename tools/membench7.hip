// membench7: how fast can the PDW bracket pass's ACCESS SHAPE read an F x 128 complex64 matrix?  Bare loads + one max.
//   A  lane = channel, 8-byte loads, 16 rows in flight, a wave walks 4 words of 64 rows, column groups of 64 (the kernel's shape)
//   B  lane = two adjacent channels, 16-byte loads, 8 rows in flight, a wave walks 4 words, one column group of 128
//   C  as B, 16 rows in flight
//   D  flat: thread t reads float4 t, t + stride, ... (the plain streaming read)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int IF>
__global__ void __launch_bounds__(256) kA(const float2* y, long long F, int M, float* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  float mx = 0.f;
  for (int wi = wave; wi < 16; wi += 4) {
    const long long r0 = ((long long)blockIdx.y * 16 + wi) * 64;
    if (r0 + 64 > F) break;
    for (int i = 0; i < 64; i += IF) {
      float2 v[IF];
#pragma unroll
      for (int u = 0; u < IF; ++u) v[u] = y[(r0 + i + u) * M + col];
#pragma unroll
      for (int u = 0; u < IF; ++u) mx = fmaxf(mx, v[u].x * v[u].x + v[u].y * v[u].y);
    }
  }
  if (mx == 123.456f) out[0] = mx;
}
template <int IF>
__global__ void __launch_bounds__(256) kB(const float2* y, long long F, int M, float* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.x * 128 + 2 * lane;
  float mx = 0.f;
  for (int wi = wave; wi < 16; wi += 4) {
    const long long r0 = ((long long)blockIdx.y * 16 + wi) * 64;
    if (r0 + 64 > F) break;
    for (int i = 0; i < 64; i += IF) {
      float4 v[IF];
#pragma unroll
      for (int u = 0; u < IF; ++u) v[u] = *reinterpret_cast<const float4*>(y + (r0 + i + u) * M + col);
#pragma unroll
      for (int u = 0; u < IF; ++u) mx = fmaxf(mx, fmaxf(v[u].x * v[u].x + v[u].y * v[u].y, v[u].z * v[u].z + v[u].w * v[u].w));
    }
  }
  if (mx == 123.456f) out[0] = mx;
}
__global__ void __launch_bounds__(256) kD(const float4* y, long long n4, float* out) {
  float mx = 0.f;
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += 4 * stride) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = (i + u * stride < n4) ? y[i + u * stride] : make_float4(0, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < 4; ++u) mx = fmaxf(mx, fmaxf(v[u].x * v[u].x + v[u].y * v[u].y, v[u].z * v[u].z + v[u].w * v[u].w));
  }
  if (mx == 123.456f) out[0] = mx;
}

int main() {
  const long long F = 1ll << 22; const int M = 128;
  const size_t bytes = (size_t)F * M * 8;
  float2* y; float* out;
  CK(hipMalloc(&y, bytes)); CK(hipMalloc(&out, 4));
  CK(hipMemset(y, 0, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("%-60s %.3f ms  %.2f TB/s\n", name, ms, bytes / ms / 1e9);
  };
  const unsigned gy = (unsigned)(F / 1024);
  time("A  8-byte loads, lane = channel, 16 rows in flight", [&] { hipLaunchKernelGGL(kA<16>, dim3(2, gy), dim3(256), 0, 0, y, F, M, out); });
  time("A' 8-byte loads, lane = channel, 8 rows in flight", [&] { hipLaunchKernelGGL(kA<8>, dim3(2, gy), dim3(256), 0, 0, y, F, M, out); });
  time("B  16-byte loads, lane = 2 channels, 8 rows in flight", [&] { hipLaunchKernelGGL(kB<8>, dim3(1, gy), dim3(256), 0, 0, y, F, M, out); });
  time("C  16-byte loads, lane = 2 channels, 16 rows in flight", [&] { hipLaunchKernelGGL(kB<16>, dim3(1, gy), dim3(256), 0, 0, y, F, M, out); });
  for (int g : {1024, 2048, 4096, 16384})
    time(g == 1024 ? "D  flat float4 stream, grid 1024" : g == 2048 ? "D  grid 2048" : g == 4096 ? "D  grid 4096" : "D  grid 16384",
         [&] { hipLaunchKernelGGL(kD, dim3(g), dim3(256), 0, 0, (const float4*)y, (long long)(bytes / 16), out); });
  return 0;
}
