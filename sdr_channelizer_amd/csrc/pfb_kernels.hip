// pfb_kernels.hip -- gfx950 kernels of the channelizer library and their launchers.
//
//   pfb_fast_kernel<...>   the hot path (pfb_fast.hpp), one instantiation per
//                          (M, P, D, sample format) in kFastTable below
//   pfb_generic_kernel     any M (2^k by radix-2 in LDS, 2^a 3^b 5^c 7^d by a run-time mixed-radix
//                          Stockham FFT, otherwise a plain DFT),
//                          any P, any 1 <= D <= M, any format / layout: the
//                          correctness net for configurations without a fast
//                          instantiation (e.g. the reference's own M = fs*1e-6 = 56,
//                          /root/reference/matlab/channelizer_example.m:29)
//   pfb_update_history     carries the last M*P + D raw samples to the next call
//                          (the dsp.Channelizer System-object state,
//                          channelizer_example.m:50-56)
//   pfb_stream_copy        1 read : 2 write streaming copy, the measured-HBM yardstick
#include "pfb_table.h"

namespace pfb {

// ---------------------------------------------------------------------------------
// generic kernel

__device__ __forceinline__ void fetch_sample(const KernelParams& p, long long s, float& re, float& im) {
  const void* b;
  long long i;
  if (s >= 0) { b = p.in; i = s; } else { b = p.hist; i = p.hist_samples + s; }
  if (p.fmt == PFB_FMT_INT16_IQ) {
    SampleT<PFB_FMT_INT16_IQ>::cvt(static_cast<const uint32_t*>(b)[i], re, im);
  } else if (p.fmt == PFB_FMT_INT8_IQ) {
    SampleT<PFB_FMT_INT8_IQ>::cvt(static_cast<const uint16_t*>(b)[i], re, im);
  } else {
    SampleT<PFB_FMT_CF32>::cvt(static_cast<const float2*>(b)[i], re, im);
  }
}

__device__ __forceinline__ int bit_reverse(int v, int bits) { return (int)(__brev((unsigned)v) >> (32 - bits)); }

// Radices of the generic kernel's mixed-radix FFT (band counts M = 2^a 3^b 5^c 7^d that have no fused shape): at most
// 12 stages (2^12 = 4096 = the largest M pfb_create accepts); n = 0: M has another prime factor, plain DFT.
struct GenericPlan { int n; unsigned char r[12]; };

// tile_frames frames per workgroup; sm holds tile_frames*M complex (x2 when M is not 2^k)
__global__ void __launch_bounds__(256) pfb_generic_kernel(const KernelParams p, int tile_frames, int log2m, GenericPlan plan) {
  extern __shared__ float2 sm[];
  const int M = p.M, P = p.P, D = p.D;
  const int tid = threadIdx.x, nt = blockDim.x;
  const long long f0 = (long long)blockIdx.x * tile_frames;
  const int eff_off = p.base + D - 1;
  const float im_sign = (p.flags & PFB_FLAG_CONJUGATE_INPUT) ? -1.f : 1.f;

  // polyphase branches u_p[f] = sum_q h[p + Mq] x[f D + eff_off - p - Mq]
  for (int idx = tid; idx < tile_frames * M; idx += nt) {
    const int t = idx / M, br = idx - t * M;
    const long long f = f0 + t;
    float ar = 0.f, ai = 0.f;
    if (f < p.frames) {
      for (int q = 0; q < P; ++q) {
        float xr, xi;
        fetch_sample(p, f * D + eff_off - br - (long long)M * q, xr, xi);
        const float hq = p.taps[br + M * q];
        ar = fmaf(hq, xr, ar);
        ai = fmaf(hq, xi, ai);
      }
    }
    const int pos = (log2m >= 0) ? bit_reverse(br, log2m) : br;
    sm[t * M + pos] = make_float2(ar, ai * im_sign);
  }
  __syncthreads();

  if (log2m >= 0) {  // radix-2 DIT, e^{+j} kernel, in place
    for (int len = 2; len <= M; len <<= 1) {
      const int half = len >> 1, step = M / len;
      for (int idx = tid; idx < tile_frames * (M / 2); idx += nt) {
        const int t = idx / (M / 2), b = idx - t * (M / 2);
        const int j = b % half, i = (b / half) * len + j;
        const float2 w = p.tw[j * step];
        float2* a = &sm[t * M + i];
        const float2 lo = a[0], hi = a[half];
        const float br_ = hi.x * w.x - hi.y * w.y, bi_ = hi.x * w.y + hi.y * w.x;
        a[0] = make_float2(lo.x + br_, lo.y + bi_);
        a[half] = make_float2(lo.x - br_, lo.y - bi_);
      }
      __syncthreads();
    }
  } else if (plan.n > 0) {
    // Stockham autosort, radices 2 ... 7 at run time: O(M sum r) per frame instead of the plain DFT's O(M^2) -- 127 ms
    // -> a few ms per 2^28 samples at M = 560.  Stage with radix r after sub-transforms of length Ns: butterfly j takes
    // a[j + q M/r] e^{+j 2 pi q k / (Ns r)} (k = j mod Ns), an r-point DFT of those (twiddles from the M-entry table:
    // Ns r and r divide M), and leaves u_q at (j / Ns) Ns r + k + q Ns of the other buffer: natural order at the end.
    float2* a = sm;
    float2* b = sm + tile_frames * M;
    int Ns = 1;
    for (int st = 0; st < plan.n; ++st) {
      const int r = plan.r[st], nb = M / r, tstep = M / (Ns * r), rstep = M / r;
      for (int idx = tid; idx < tile_frames * nb; idx += nt) {
        const int t = idx / nb, j = idx - t * nb;
        const int k = j % Ns;
        float2 v[7];
        for (int q = 0; q < r; ++q) {
          const float2 x = a[t * M + j + q * nb], w = p.tw[(q * k * tstep) % M];
          v[q] = make_float2(x.x * w.x - x.y * w.y, x.x * w.y + x.y * w.x);
        }
        float2* dst = b + t * M + (j / Ns) * Ns * r + k;
        for (int pp = 0; pp < r; ++pp) {
          float ur = 0.f, ui = 0.f;
          for (int q = 0; q < r; ++q) {
            const float2 w = p.tw[((pp * q) % r) * rstep];
            ur += v[q].x * w.x - v[q].y * w.y;
            ui += v[q].x * w.y + v[q].y * w.x;
          }
          dst[pp * Ns] = make_float2(ur, ui);
        }
      }
      __syncthreads();
      float2* sw = a; a = b; b = sw;
      Ns *= r;
    }
    if (a != sm + tile_frames * M) {  // the epilogue reads the second half
      for (int idx = tid; idx < tile_frames * M; idx += nt) sm[tile_frames * M + idx] = a[idx];
      __syncthreads();
    }
  } else {  // plain DFT into the second half of sm
    float2* y = sm + tile_frames * M;
    for (int idx = tid; idx < tile_frames * M; idx += nt) {
      const int t = idx / M, k = idx - t * M;
      float ar = 0.f, ai = 0.f;
      int j = 0;  // (k*p) mod M, advanced incrementally
      for (int pp = 0; pp < M; ++pp) {
        const float2 u = sm[t * M + pp], w = p.tw[j];
        ar += u.x * w.x - u.y * w.y;
        ai += u.x * w.y + u.y * w.x;
        j += k; if (j >= M) j -= M;
      }
      y[idx] = make_float2(ar, ai);
    }
    __syncthreads();
  }

  const float2* y = (log2m >= 0) ? sm : sm + tile_frames * M;
  const int half_shift = M / 2;  // fftshift(out,2): dst = (k + floor(M/2)) mod M
  if (p.layout == PFB_LAYOUT_FRAME_MAJOR) {
    for (int idx = tid; idx < tile_frames * M; idx += nt) {
      const int t = idx / M, k = idx - t * M;
      const long long f = f0 + t;
      if (f >= p.frames) continue;
      float2 v = y[idx];
      if (p.flags & PFB_FLAG_DEROTATE) {
        const long long rot = ((p.frame0 + f) * D) % M;
        const float2 w = p.tw[(int)(((long long)k * rot) % M)];
        v = make_float2(v.x * w.x + v.y * w.y, v.y * w.x - v.x * w.y);  // * conj(w)
      }
      const int col = (p.flags & PFB_FLAG_FFTSHIFT) ? (k + half_shift) % M : k;
      if (p.flags & PFB_FLAG_MAGNITUDE) reinterpret_cast<float*>(p.out)[f * M + col] = mag_out(v.x, v.y, p.flags);
      else p.out[f * M + col] = v;
    }
  } else {  // CHANNEL_MAJOR: consecutive threads write consecutive frames of one channel
    for (int idx = tid; idx < tile_frames * M; idx += nt) {
      const int k = idx / tile_frames, t = idx - k * tile_frames;
      const long long f = f0 + t;
      if (f >= p.frames) continue;
      float2 v = y[t * M + k];
      if (p.flags & PFB_FLAG_DEROTATE) {
        const long long rot = ((p.frame0 + f) * D) % M;
        const float2 w = p.tw[(int)(((long long)k * rot) % M)];
        v = make_float2(v.x * w.x + v.y * w.y, v.y * w.x - v.x * w.y);
      }
      const int col = (p.flags & PFB_FLAG_FFTSHIFT) ? (k + half_shift) % M : k;
      const long long o = (long long)col * p.out_ld + p.out_frame0 + f;
      if (p.flags & PFB_FLAG_MAGNITUDE) reinterpret_cast<float*>(p.out)[o] = mag_out(v.x, v.y, p.flags);
      else p.out[o] = v;
    }
  }
}

hipError_t launch_generic(const KernelParams& p, hipStream_t s) {
  if (p.frames <= 0) return hipSuccess;
  const int M = p.M;
  int log2m = -1;
  if ((M & (M - 1)) == 0) { log2m = 0; while ((1 << log2m) < M) ++log2m; }
  int tile = 4096 / M;
  if (tile < 1) tile = 1;
  if (tile > 16) tile = 16;
  const size_t shmem = (size_t)tile * M * sizeof(float2) * (log2m >= 0 ? 1 : 2);
  if (shmem > 64 * 1024) return hipErrorInvalidValue;
  GenericPlan plan{};
  if (log2m < 0) {  // 7, 5, 3 first, then 4s and a 2: every stage length divides M by construction
    int m = M;
    for (int r : {7, 5, 3, 4, 2})
      while (m % r == 0 && plan.n < 12) { plan.r[plan.n++] = (unsigned char)r; m /= r; }
    if (m != 1) plan.n = 0;
  }
  const long long blocks = (p.frames + tile - 1) / tile;
  hipLaunchKernelGGL(pfb_generic_kernel, dim3((unsigned)blocks), dim3(256), shmem, s, p, tile, log2m, plan);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------
// history update: new[i] = stream[n_in - H + i], stream = [old history | in]

template <typename T>
__global__ void pfb_update_history_kernel(const T* old_hist, const T* in, long long n_in, T* new_hist, int H) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= H) return;
  const long long s = n_in - H + i;
  new_hist[i] = (s >= 0) ? in[s] : old_hist[H + s];
}

hipError_t launch_update_history(const void* old_hist, const void* in, long long n_in, void* new_hist,
                                 int hist_samples, int bps, hipStream_t s) {
  const dim3 grid((hist_samples + 255) / 256), block(256);
  if (bps == 2)
    hipLaunchKernelGGL(pfb_update_history_kernel<uint16_t>, grid, block, 0, s, (const uint16_t*)old_hist,
                       (const uint16_t*)in, n_in, (uint16_t*)new_hist, hist_samples);
  else if (bps == 4)
    hipLaunchKernelGGL(pfb_update_history_kernel<uint32_t>, grid, block, 0, s, (const uint32_t*)old_hist,
                       (const uint32_t*)in, n_in, (uint32_t*)new_hist, hist_samples);
  else
    hipLaunchKernelGGL(pfb_update_history_kernel<uint64_t>, grid, block, 0, s, (const uint64_t*)old_hist,
                       (const uint64_t*)in, n_in, (uint64_t*)new_hist, hist_samples);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------
// channel-major by slabs: slab[frames][M] (frame-major, written a moment ago by a channelizer kernel)
// -> out[col * out_ld + out_frame0 + f].  64 x 64 tiles through LDS: rows are read 64 columns at
// a time (512 contiguous bytes per wave instruction) and columns written 64 frames at a time (512 contiguous bytes).

// 64 frames x 64 channels per tile: a column leaves as 64 consecutive frames.  Longer tiles do not help (measured at
// M = 1024, profiles/r02_channel_major_team_routes.txt: 64 / 128 / 256 frames move the 2^30-sample matrix in 6.5 / 6.4 /
// 7.1 ms including the channelizer kernel); with its loads or its stores switched off the kernel costs 0.94 ms less each,
// with both off the route still takes 3.7 ms against 2.66 for the channelizer kernel in one launch: slab by slab, every
// launch is a single wave of workgroups whose ramp-up and drain nothing hides.  The tile goes through LDS in row groups
// of 16 frames per wave (16 loads in flight per lane).
template <typename T>
__global__ void __launch_bounds__(256) pfb_transpose_slab_kernel(const T* slab, long long frames, int M, T* out,
                                                                 long long out_ld, long long out_frame0) {
  constexpr int FT = 64;
  __shared__ T tile[64][FT + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = blockIdx.x * 64;
  const long long f0 = (long long)blockIdx.y * FT;
  constexpr int PER_WAVE = FT / 4;  // frames each wave brings in
  {
    T v[PER_WAVE];
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
      const long long f = f0 + wave * PER_WAVE + i;
      v[i] = (f < frames && c0 + lane < M) ? slab[f * M + c0 + lane] : T{};
    }
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) tile[lane][wave * PER_WAVE + i] = v[i];
  }
  __syncthreads();
  // wave w writes channels c0 + 16 w ... + 15: a store instruction is 64 consecutive frames of one channel
#pragma unroll 1
  for (int c = 0; c < 16; ++c) {
    const int col = c0 + wave * 16 + c;
    if (col >= M) break;
    const long long f = f0 + lane;
    if (f < frames) out[(long long)col * out_ld + out_frame0 + f] = tile[wave * 16 + c][lane];
  }
}

hipError_t launch_transpose_slab(const void* slab, long long frames, int M, void* out, long long out_ld,
                                 long long out_frame0, int elem_bytes, hipStream_t s) {
  if (frames <= 0) return hipSuccess;
  const dim3 grid((unsigned)((M + 63) / 64), (unsigned)((frames + 63) / 64)), block(256);
  if (elem_bytes == 8)
    hipLaunchKernelGGL(pfb_transpose_slab_kernel<float2>, grid, block, 0, s, (const float2*)slab, frames, M, (float2*)out, out_ld, out_frame0);
  else
    hipLaunchKernelGGL(pfb_transpose_slab_kernel<float>, grid, block, 0, s, (const float*)slab, frames, M, (float*)out, out_ld, out_frame0);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------
// 1 read : 2 write streaming copy (same byte mix as int16 -> complex64, D = M): the yardstick next to the
// nominal roofline.  The shape is the fastest of tools/membench2's sweep on MI355X (profiles/r01_membench2_patterns.txt,
// 6.2 TB/s): short-lived 4-wave workgroups in dispatch order, each wave reading two 256-byte rows (one dword per lane
// and row) and writing them back as ONE 16-byte store per lane (1 KB per wave instruction) -- DRAM rows are finished
// while they are open.  A grid-stride loop over the same bytes (what this was in round 1) reaches 4.7 TB/s.

__global__ void __launch_bounds__(256) pfb_stream_copy_kernel(const unsigned* in, uint4* out, long long row_pairs) {
  const long long pair = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= row_pairs) return;
  const int lane = threadIdx.x & 63;
  const unsigned a = in[pair * 128 + lane], b = in[pair * 128 + 64 + lane];
  out[pair * 64 + lane] = make_uint4(a, a + 1u, b, b + 1u);
}

hipError_t launch_stream_copy(const void* in, void* out, long long n_vec16, hipStream_t s) {
  const long long row_pairs = n_vec16 / 32;  // 512 bytes of input per wave
  if (row_pairs <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(pfb_stream_copy_kernel, dim3((unsigned)((row_pairs + 3) / 4)), dim3(256), 0, s, (const unsigned*)in,
                     (uint4*)out, row_pairs);
  return hipGetLastError();
}

// Copy kernels of the channelizer's byte mixes with NO arithmetic (tools/membench6 in the library, so that bench.py can
// print the bound next to every shape's fraction): each wave owns `spw` consecutive 256-byte rows of the input (one dword
// per lane and row) and writes RATIO x 256 bytes per row with 16-byte stores; 4-wave workgroups in dispatch order.
template <int RATIO>
__global__ void __launch_bounds__(256) pfb_mix_copy_kernel(const unsigned* in, uint4* out, long long rows, int spw) {
  const long long wv = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const long long r0 = wv * spw;
  if (r0 >= rows) return;
  for (int s = 0; s < spw; s += 2) {  // two rows per step: 512 B in, RATIO x 512 B out
    const unsigned a = in[(r0 + s) * 64 + lane], b = in[(r0 + s + 1) * 64 + lane];
    uint4* o = out + (r0 + s) * 16 * RATIO + lane;
#pragma unroll
    for (int j = 0; j < RATIO / 2; ++j) o[j * 64] = make_uint4(a, a + j, b, b + j);
  }
}

hipError_t launch_mix_copy(const void* in, void* out, long long rows, int write_ratio, int spw, hipStream_t s) {
  if (rows <= 0 || spw < 2 || (spw & 1) || rows % spw) return hipErrorInvalidValue;
  const long long waves = rows / spw;
  const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
  if (write_ratio == 4) hipLaunchKernelGGL(pfb_mix_copy_kernel<4>, grid, block, 0, s, (const unsigned*)in, (uint4*)out, rows, spw);
  else if (write_ratio == 2) hipLaunchKernelGGL(pfb_mix_copy_kernel<2>, grid, block, 0, s, (const unsigned*)in, (uint4*)out, rows, spw);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------
// fused-kernel table, part 1: the M = 64 shapes (cfg2 and its int8 / cf32 siblings)
//                 M    P   D  CPT FMT               C NP R0 R1 R2 RS0 RS1 RS2 FS  PP     MINW  TW_TABLE
using Cfg64x12i16 = FastCfg<64, 12, 64, 1, PFB_FMT_INT16_IQ, 8, 2, 8, 8, 1, 8, 9, 0, 72, false, 4>;
using Cfg64x12i8  = FastCfg<64, 12, 64, 1, PFB_FMT_INT8_IQ,  8, 2, 8, 8, 1, 8, 9, 0, 72, false, 4>;
using Cfg64x12f32 = FastCfg<64, 12, 64, 1, PFB_FMT_CF32,     8, 2, 8, 8, 1, 8, 9, 0, 72, false, 4>;
// 13 ... 16 taps per band on the cfg2 bank (shorter prototypes are zero-padded onto the 12-tap kernels at create)
using Cfg64x16i16 = FastCfg<64, 16, 64, 1, PFB_FMT_INT16_IQ, 8, 2, 8, 8, 1, 8, 9, 0, 72, false, 4>;

static const FastEntry kRows[] = {
    entry<Cfg64x12i16>("pfb_fast<M64,P12,D64,int16>", 512, 4),
    entry<Cfg64x12i8>("pfb_fast<M64,P12,D64,int8>", 256, 7),  // 8-bit rows are half as long: pairs over long runs beat the shared-halo tiles by 10 %
    entry<Cfg64x12f32>("pfb_fast<M64,P12,D64,cf32>", 512, 4),
    entry<Cfg64x16i16>("pfb_fast<M64,P16,D64,int16>", 512, 4),
};

FastTablePart fast_table_m64() { return FastTablePart{kRows, (int)(sizeof(kRows) / sizeof(kRows[0]))}; }

const FastKernelInfo* find_fast_kernel(int M, int P, int D, int fmt, int variant, bool channel_major) {
  const FastTablePart parts[] = {fast_table_m64(), fast_table_mid(), fast_table_big(), fast_table_mixed()};
  for (const FastTablePart& part : parts)
    for (int i = 0; i < part.count; ++i) {
      const FastEntry& e = part.rows[i];
      if (e.M == M && e.P == P && e.D == D && e.fmt == fmt && (!channel_major || e.info.channel_major_ok) && variant-- == 0)
        return &e.info;
    }
  return nullptr;
}

}  // namespace pfb
