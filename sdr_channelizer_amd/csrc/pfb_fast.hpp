// pfb_fast.hpp -- the hot path: fused polyphase FIR + M-point FFT for gfx950.
//
// Replaces the arithmetic the reference delegates to dsp.Channelizer
// (/root/reference/matlab/channelizer_example.m:31,56,
//  /root/reference/matlab/create_pdws_channelized.m:33,57) together with the
// int->complex normalise (channelizer_example.m:18-21) and fftshift (:58).
//
// One workgroup = NT = D/CPT threads walks a contiguous run of frames.
//
//  FIR   Thread `tid` owns CPT adjacent columns c of the "row" view of the
//        stream (row r = the D samples whose newest member is frame r's newest
//        sample; a wave's load of one row is one contiguous, fully coalesced
//        segment).  Column c feeds branches p = (D-1-c) + D*ph, ph = 0..M/D-1,
//        whose taps h[p + M q] stay in registers, as does a sliding window of
//        the last W = M*P/D rows (already converted to float).  Every input
//        sample is loaded and converted exactly once per run.
//  FFT   A chunk of C frames of branch outputs goes through LDS as a mixed
//        radix (R0 x R1 [x R2]) Cooley-Tukey: each pass is an in-register
//        R-point DFT per thread, with one LDS exchange between passes.
//        Layout of the input of pass i inside a frame:
//            pos_i(n_i, item) = n_i*RS_i + item,  item = kk*S_i + rest
//        (S_i = prod R_{j>i}, K_i = prod R_{j<i}); RS_i / FS are padded so that
//        ds_read_b64 / ds_write_b64 are bank-conflict free (tools/fft_plan_model.py
//        checks the formulas and the padding for every instantiated plan).
//  OUT   The last pass leaves thread `kk` with channels kk + K*k: for each k the
//        threads of a frame store M/R_last contiguous complex64.
//
// No MFMA: 4P + 5 log2 M flop per sample against 12 B of HBM traffic -- the
// kernel is HBM-bound.
#pragma once

#include "pfb_common.h"

namespace pfb {

// ---------------------------------------------------------------------------------
// In-register N-point DFT, kernel e^{+j 2 pi n k / N}, natural order in and out.

#define PFB_DEV static __device__ __forceinline__

constexpr float kSqrtHalf = 0.70710678118654752f;
constexpr float kCosPi8 = 0.92387953251128674f;
constexpr float kSinPi8 = 0.38268343236508977f;

// (r,i) *= e^{+j 2 pi K / N}, K and N compile-time, N <= 16
template <int N, int K>
PFB_DEV void tw_mul(float& r, float& i) {
  static_assert(16 % N == 0, "small DFT sizes only");
  constexpr int q = K * (16 / N);  // sixteenths of a turn, 0..7
  static_assert(q >= 0 && q < 8, "only the upper half plane is needed");
  if constexpr (q == 0) {
  } else if constexpr (q == 4) {  // +j
    const float t = r; r = -i; i = t;
  } else if constexpr (q == 2) {  // (1+j)/sqrt2
    const float t = (r - i) * kSqrtHalf; i = (r + i) * kSqrtHalf; r = t;
  } else if constexpr (q == 6) {  // (-1+j)/sqrt2
    const float t = (-r - i) * kSqrtHalf; i = (r - i) * kSqrtHalf; r = t;
  } else {
    constexpr float c = (q == 1) ? kCosPi8 : (q == 3) ? kSinPi8 : (q == 5) ? -kSinPi8 : -kCosPi8;
    constexpr float s = (q == 1) ? kSinPi8 : (q == 3) ? kCosPi8 : (q == 5) ? kCosPi8 : kSinPi8;
    const float t = r * c - i * s; i = r * s + i * c; r = t;
  }
}

typedef float v2f_t __attribute__((ext_vector_type(2)));

// one complex64 output element; nontemporal = streaming store (output is write-once)
PFB_DEV void store_c64(float2* dst, float re, float im, int nontemporal) {
  v2f_t v = {re, im};
  if (nontemporal) __builtin_nontemporal_store(v, reinterpret_cast<v2f_t*>(dst));
  else *reinterpret_cast<v2f_t*>(dst) = v;
}

template <int N> struct Dft;

template <> struct Dft<2> {
  PFB_DEV void run(float (&re)[2], float (&im)[2]) {
    const float ar = re[0], ai = im[0];
    re[0] = ar + re[1]; im[0] = ai + im[1];
    re[1] = ar - re[1]; im[1] = ai - im[1];
  }
};

template <> struct Dft<4> {
  PFB_DEV void run(float (&re)[4], float (&im)[4]) {
    const float t0r = re[0] + re[2], t0i = im[0] + im[2];
    const float t1r = re[0] - re[2], t1i = im[0] - im[2];
    const float t2r = re[1] + re[3], t2i = im[1] + im[3];
    const float t3r = re[1] - re[3], t3i = im[1] - im[3];
    re[0] = t0r + t2r; im[0] = t0i + t2i;
    re[2] = t0r - t2r; im[2] = t0i - t2i;
    re[1] = t1r - t3i; im[1] = t1i + t3r;  // t1 + j t3
    re[3] = t1r + t3i; im[3] = t1i - t3r;  // t1 - j t3
  }
};

template <int N, int K>
struct DftCombine {  // X[k] = E[k] + W^k O[k],  X[k+N/2] = E[k] - W^k O[k]
  PFB_DEV void run(float (&re)[N], float (&im)[N], const float (&er)[N / 2], const float (&ei)[N / 2],
                   float (&orr)[N / 2], float (&oi)[N / 2]) {
    tw_mul<N, K>(orr[K], oi[K]);
    re[K] = er[K] + orr[K]; im[K] = ei[K] + oi[K];
    re[K + N / 2] = er[K] - orr[K]; im[K + N / 2] = ei[K] - oi[K];
    if constexpr (K + 1 < N / 2) DftCombine<N, K + 1>::run(re, im, er, ei, orr, oi);
  }
};

template <int N> struct Dft {
  PFB_DEV void run(float (&re)[N], float (&im)[N]) {
    float er[N / 2], ei[N / 2], orr[N / 2], oi[N / 2];
#pragma unroll
    for (int k = 0; k < N / 2; ++k) {
      er[k] = re[2 * k]; ei[k] = im[2 * k];
      orr[k] = re[2 * k + 1]; oi[k] = im[2 * k + 1];
    }
    Dft<N / 2>::run(er, ei);
    Dft<N / 2>::run(orr, oi);
    DftCombine<N, 0>::run(re, im, er, ei, orr, oi);
  }
};

// ---------------------------------------------------------------------------------
// Kernel configuration

template <int M_, int P_, int D_, int CPT_, int FMT_, int C_, int NP_, int R0_, int R1_, int R2_, int RS0_,
          int RS1_, int RS2_, int FS_, bool PINGPONG_, int MIN_WAVES_>
struct FastCfg {
  static constexpr int M = M_, P = P_, D = D_, CPT = CPT_, FMT = FMT_, C = C_, NP = NP_;
  static constexpr int NT = D / CPT;   // threads per workgroup
  static constexpr int W = M * P / D;  // window rows = taps per column
  static constexpr int OS = M / D;     // branches per column (1, or 2 when oversampled)
  static constexpr int FS = FS_;       // frame stride in LDS (complex elements)
  static constexpr bool PINGPONG = PINGPONG_;
  static constexpr int MIN_WAVES = MIN_WAVES_;
  static constexpr int R(int i) { return i == 0 ? R0_ : i == 1 ? R1_ : R2_; }
  static constexpr int RS(int i) { return i == 0 ? RS0_ : i == 1 ? RS1_ : RS2_; }
  static constexpr int S(int i) { int s = 1; for (int j = i + 1; j < NP; ++j) s *= R(j); return s; }
  static constexpr int K(int i) { int k = 1; for (int j = 0; j < i; ++j) k *= R(j); return k; }
  static constexpr int BUF = C * FS;   // one chunk buffer (complex elements)
  static constexpr int LDS_ELEMS = BUF * (PINGPONG ? 2 : 1);
  static_assert(D % CPT == 0 && NT % 64 == 0, "whole waves");
  static_assert(M % D == 0 && (M * P) % D == 0, "D divides M");
  static_assert(NP >= 2 && NP <= 3, "2 or 3 passes");
  static_assert(R0_ * R1_ * (NP_ == 3 ? R2_ : 1) == M_, "radices multiply to M");
  static_assert(R(0) * RS(0) <= FS && R(1) * RS(1) <= FS && (NP < 3 || R(2) * RS(2) <= FS), "frame fits");
  // in-place passes are only safe when one wave does the whole pass in one go
  static_assert(PINGPONG || (NT == 64 && C * (M / R(0)) <= 64 && (NP < 3 || C * (M / R(1)) <= 64)),
                "multi-wave or multi-iteration non-final passes need ping-pong buffers");
};

// ---------------------------------------------------------------------------------

template <class K>
struct FastKernel {
  using ST = SampleT<K::FMT>;
  using raw_t = typename ST::raw_t;
  static constexpr int M = K::M, P = K::P, D = K::D, CPT = K::CPT, C = K::C, W = K::W, OS = K::OS, NT = K::NT;
  static constexpr int NW = W - 1 + C;  // window rows held in registers during a chunk

  struct alignas(sizeof(raw_t) * CPT) RawVec { raw_t v[CPT]; };

  // Row r of the stream -> CPT raw samples for this thread.  r is uniform across
  // the workgroup, so the branches below are too.
  PFB_DEV void load_row(const KernelParams& p, long long r, int c0, raw_t (&raw)[CPT]) {
    if (r >= p.frames) {  // padding frames of a partial last chunk
#pragma unroll
      for (int cc = 0; cc < CPT; ++cc) raw[cc] = raw_t{};
      return;
    }
    const long long s0 = r * D + p.base;
    const raw_t* in = static_cast<const raw_t*>(p.in);
    if (s0 >= 0 && p.vec_ok) {
      const RawVec v = *reinterpret_cast<const RawVec*>(in + s0 + c0);
#pragma unroll
      for (int cc = 0; cc < CPT; ++cc) raw[cc] = v.v[cc];
    } else {  // start of the call: part of the row is history
      const raw_t* hist = static_cast<const raw_t*>(p.hist);
#pragma unroll
      for (int cc = 0; cc < CPT; ++cc) {
        const long long s = s0 + c0 + cc;
        raw[cc] = (s >= 0) ? in[s] : hist[p.hist_samples + s];
      }
    }
  }

  template <int I>
  PFB_DEV void pass(const KernelParams& p, float2* src, float2* dst, int tid, long long f0,
                    const float (&twr)[2][16], const float (&twi)[2][16]) {
    constexpr int R = K::R(I), S = K::S(I), KK = K::K(I), RS = K::RS(I);
    constexpr int IPF = M / R, ITEMS = C * IPF, ITERS = (ITEMS + NT - 1) / NT;
    constexpr bool LAST = (I == K::NP - 1);
    constexpr bool TW_REGS = (ITERS == 1);
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int w = tid + it * NT;
      const bool active = (ITEMS % NT == 0) || (w < ITEMS);
      const int fc = w / IPF, item = w % IPF;
      const int kk = item / S, rest = item % S;
      float re[R], im[R];
      if (active) {
#pragma unroll
        for (int n = 0; n < R; ++n) {
          const float2 v = src[fc * K::FS + n * RS + item];
          re[n] = v.x; im[n] = v.y;
        }
      } else {
#pragma unroll
        for (int n = 0; n < R; ++n) { re[n] = 0.f; im[n] = 0.f; }
      }
      Dft<R>::run(re, im);
      if constexpr (!LAST) {
        constexpr int S1 = K::S(I + 1), RS1 = K::RS(I + 1);
        // twiddle e^{+j 2 pi rest k / (R S)} = tw[rest * k * KK]
#pragma unroll
        for (int k = 1; k < R; ++k) {
          float c, s;
          if constexpr (TW_REGS) { c = twr[I][k]; s = twi[I][k]; }
          else { const float2 t = p.tw[rest * k * KK]; c = t.x; s = t.y; }
          const float t = re[k] * c - im[k] * s;
          im[k] = re[k] * s + im[k] * c;
          re[k] = t;
        }
        const int n1 = rest / S1, rest2 = rest % S1;
        if (active) {
#pragma unroll
          for (int k = 0; k < R; ++k) {
            const int item2 = (kk + k * KK) * S1 + rest2;
            dst[fc * K::FS + n1 * RS1 + item2] = make_float2(re[k], im[k]);
          }
        }
      } else {
        const long long f = f0 + fc;
        if (active && f < p.frames) {
          const bool flip_odd = (OS == 2) && (p.flags & PFB_FLAG_DEROTATE) && ((p.frame0 + f) & 1);
          const int shift = (p.flags & PFB_FLAG_FFTSHIFT) ? (M / 2) : 0;
          float2* row = p.out + f * M;
#pragma unroll
          for (int k = 0; k < R; ++k) {
            const int ch = kk + k * KK;
            float vr = re[k], vi = im[k];
            if (flip_odd && (ch & 1)) { vr = -vr; vi = -vi; }
            store_c64(&row[ch ^ shift], vr, vi, p.nontemporal);
          }
        }
      }
    }
  }

  PFB_DEV void run(const KernelParams& p, float2* lds) {
    const int tid = threadIdx.x;
    const long long f_begin = (long long)blockIdx.x * p.frames_per_block;
    if (f_begin >= p.frames) return;
    const long long f_end = (f_begin + p.frames_per_block < p.frames) ? f_begin + p.frames_per_block : p.frames;
    const int c0 = tid * CPT;

    // taps of this thread's columns: h[p_lo + D*j], p_lo = D-1-c
    float h[W][CPT];
#pragma unroll
    for (int j = 0; j < W; ++j)
#pragma unroll
      for (int cc = 0; cc < CPT; ++cc) h[j][cc] = p.taps[(D - 1 - (c0 + cc)) + D * j];
    const float im_sign = (p.flags & PFB_FLAG_CONJUGATE_INPUT) ? -1.f : 1.f;

    // inter-pass twiddles for passes whose item -> thread map is fixed
    float twr[2][16], twi[2][16];
#pragma unroll
    for (int i = 0; i < K::NP - 1; ++i) {
      const int R = K::R(i), S = K::S(i), KK = K::K(i), IPF = M / R;
      if (C * IPF <= NT) {
        const int rest = (tid % IPF) % S;
#pragma unroll
        for (int k = 1; k < 16; ++k) {
          if (k < R) { const float2 t = p.tw[rest * k * KK]; twr[i][k] = t.x; twi[i][k] = t.y; }
        }
      }
    }

    // LDS position of this thread's FIR outputs inside a frame (pass-0 layout)
    int upos[OS][CPT];
#pragma unroll
    for (int ph = 0; ph < OS; ++ph)
#pragma unroll
      for (int cc = 0; cc < CPT; ++cc) {
        const int n = (D - 1 - (c0 + cc)) + D * ph;
        upos[ph][cc] = (n / K::S(0)) * K::RS(0) + (n % K::S(0));
      }

    // window: x[i] is row (chunk_first_frame - (W-1) + i)
    float xr[NW][CPT], xi[NW][CPT];
    raw_t raw[C][CPT];
#pragma unroll
    for (int i = 0; i < W - 1; ++i) {
      raw_t t[CPT];
      load_row(p, f_begin - (W - 1) + i, c0, t);
#pragma unroll
      for (int cc = 0; cc < CPT; ++cc) ST::cvt(t[cc], xr[i][cc], xi[i][cc]);
    }
#pragma unroll
    for (int t = 0; t < C; ++t) load_row(p, f_begin + t, c0, raw[t]);

    float2* buf0 = lds;
    float2* buf1 = K::PINGPONG ? lds + K::BUF : lds;

    for (long long f0 = f_begin; f0 < f_end; f0 += C) {
#pragma unroll
      for (int t = 0; t < C; ++t)
#pragma unroll
        for (int cc = 0; cc < CPT; ++cc) ST::cvt(raw[t][cc], xr[W - 1 + t][cc], xi[W - 1 + t][cc]);
      if (f0 + C < f_end) {  // prefetch the next chunk's rows under this chunk's FFT
#pragma unroll
        for (int t = 0; t < C; ++t) load_row(p, f0 + C + t, c0, raw[t]);
      }

      // FIR: u_{p_lo + D ph}[t] = sum_q h[ph + OS q] * x[row t - ph - OS q]
#pragma unroll
      for (int t = 0; t < C; ++t)
#pragma unroll
        for (int ph = 0; ph < OS; ++ph)
#pragma unroll
          for (int cc = 0; cc < CPT; ++cc) {
            float ar = 0.f, ai = 0.f;
#pragma unroll
            for (int q = 0; q < P; ++q) {
              const int j = ph + OS * q;
              ar = fmaf(h[j][cc], xr[W - 1 + t - j][cc], ar);
              ai = fmaf(h[j][cc], xi[W - 1 + t - j][cc], ai);
            }
            buf0[t * K::FS + upos[ph][cc]] = make_float2(ar, ai * im_sign);
          }
      __syncthreads();

      pass<0>(p, buf0, buf1, tid, f0, twr, twi);
      __syncthreads();
      if constexpr (K::NP == 2) {
        pass<1>(p, buf1, nullptr, tid, f0, twr, twi);
      } else {
        pass<1>(p, buf1, buf0, tid, f0, twr, twi);
        __syncthreads();
        pass<2>(p, buf0, nullptr, tid, f0, twr, twi);
      }
      __syncthreads();  // the next chunk's FIR overwrites buf0

      // slide the window by C rows
#pragma unroll
      for (int i = 0; i < W - 1; ++i)
#pragma unroll
        for (int cc = 0; cc < CPT; ++cc) { xr[i][cc] = xr[i + C][cc]; xi[i][cc] = xi[i + C][cc]; }
    }
  }
};

template <class K>
__global__ void __launch_bounds__(K::NT, K::MIN_WAVES) pfb_fast_kernel(const KernelParams p) {
  __shared__ float2 lds[K::LDS_ELEMS];
  FastKernel<K>::run(p, lds);
}

template <class K>
hipError_t launch_fast(const KernelParams& p, hipStream_t s) {
  const long long blocks = (p.frames + p.frames_per_block - 1) / p.frames_per_block;
  if (blocks <= 0) return hipSuccess;
  hipLaunchKernelGGL(pfb_fast_kernel<K>, dim3((unsigned)blocks), dim3(K::NT), 0, s, p);
  return hipGetLastError();
}

}  // namespace pfb
