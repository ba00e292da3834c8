// membench3.hip -- access-shape study for a cooperative-workgroup channelizer schedule.
// One "row" = 64 dwords in (256 B) and 128 dwords out (512 B).  Variants:
//  wgstream: NWG persistent workgroups of NWV waves; each WG owns one long contiguous region and
//            advances through it in steps of NWV*RPS rows (wave w handles rows [w*RPS,(w+1)*RPS) of the
//            step); optional __syncthreads per step.
//  sweep:    same WGs, but step s of WG g is tile (s*NWG + g): the chip sweeps the stream compactly.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include <functional>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef unsigned int u2 __attribute__((ext_vector_type(2)));

template <int NWV, int RPS, bool SWEEP, bool BARRIER, bool REMAP>
__global__ void __launch_bounds__(64 * NWV) k_wg(const unsigned* in, u2* out, long long rows_total) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  long long g = blockIdx.x;
  const long long nwg = gridDim.x;
  if (REMAP) g = (g & 7) * (nwg >> 3) + (g >> 3);
  const long long tile_rows = (long long)NWV * RPS;
  const long long ntiles = rows_total / tile_rows;
  const long long per_wg = ntiles / nwg;
  unsigned v[RPS];
  for (long long s = 0; s < per_wg; ++s) {
    const long long tile = SWEEP ? (s * nwg + g) : (g * per_wg + s);
    const long long r0 = tile * tile_rows + (long long)wave * RPS;
#pragma unroll
    for (int u = 0; u < RPS; ++u) v[u] = in[(r0 + u) * 64 + lane];
    if (BARRIER) __syncthreads();
#pragma unroll
    for (int u = 0; u < RPS; ++u) out[(r0 + u) * 64 + lane] = (u2){v[u], v[u] + 1};
  }
}

struct Case { std::string name; std::function<void()> run; double bytes; std::vector<double> ms; };

int main(int argc, char** argv) {
  long long mib = argc > 1 ? atoll(argv[1]) : 4096;
  int rounds = argc > 2 ? atoi(argv[2]) : 3;
  long long bytes_in = mib << 20, nd = bytes_in / 4, rows = nd / 64;
  void *in, *out; CK(hipMalloc(&in, bytes_in)); CK(hipMalloc(&out, 2 * bytes_in));
  CK(hipMemset(in, 1, bytes_in)); CK(hipMemset(out, 0, 2 * bytes_in));
  std::vector<Case> cases;
#define WG(NWV, RPS, SWEEP, BARRIER, REMAP, NWG) cases.push_back({"wg nwv=" #NWV " rps=" #RPS " sweep=" #SWEEP " bar=" #BARRIER " remap=" #REMAP " nwg=" #NWG, [=] { \
    hipLaunchKernelGGL((k_wg<NWV, RPS, SWEEP, BARRIER, REMAP>), dim3(NWG), dim3(64 * NWV), 0, 0, (const unsigned*)in, (u2*)out, rows); }, 12.0 * nd, {}})
  WG(1, 8, false, false, false, 4096);
  WG(1, 8, true, false, false, 4096);
  WG(1, 8, true, false, true, 4096);
  WG(1, 8, true, false, false, 8192);
  WG(4, 8, false, false, false, 1024);
  WG(4, 8, true, false, false, 1024);
  WG(4, 8, true, false, true, 1024);
  WG(8, 8, false, false, false, 512);
  WG(8, 8, false, true, false, 512);
  WG(8, 8, true, false, false, 512);
  WG(8, 8, true, true, false, 512);
  WG(8, 8, true, true, true, 512);
  WG(16, 8, false, false, false, 256);
  WG(16, 8, false, true, false, 256);
  WG(16, 8, true, true, false, 256);
  WG(16, 8, false, true, false, 512);
  WG(16, 8, true, true, false, 512);
  WG(16, 4, true, true, false, 512);
  WG(8, 16, false, true, false, 512);
  WG(8, 16, true, true, false, 512);
  WG(4, 16, true, true, false, 1024);
  WG(4, 8, true, true, false, 2048);
  WG(8, 8, true, true, false, 1024);
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (auto& c : cases) c.run();
  CK(hipDeviceSynchronize());
  for (int r = 0; r < rounds; ++r)
    for (auto& c : cases) {
      CK(hipEventRecord(a)); for (int i = 0; i < 5; ++i) c.run(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); c.ms.push_back(ms / 5);
    }
  CK(hipGetLastError());
  for (auto& c : cases) {
    std::sort(c.ms.begin(), c.ms.end());
    printf("%-64s min %7.3f med %7.3f ms  best %7.1f GB/s\n", c.name.c_str(), c.ms.front(), c.ms[c.ms.size() / 2], c.bytes / c.ms.front() / 1e6);
  }
  return 0;
}
