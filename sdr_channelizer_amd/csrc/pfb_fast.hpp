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
//
// The arithmetic above is shared by several SCHEDULES (who computes which frames
// when; PFB_OPT_SCHEDULE, bit-identical outputs per shape):
//   0 (A)  one sliding run per workgroup, FIR and FFT by the same threads   (M = 256, small and mixed-radix banks)
//   2 (C)  one chunk per wave, adjacent chunks per workgroup  (access-shape study; channel-major fallback)
//   3 (D)  short runs whose halo rows are shared through LDS
//   4 (F)  D with a FIR wave and an FFT wave per run          (M = 64 default)
//   6 (T)  a FIR team and an FFT team per workgroup           (M = 1024 / 560 default)
//   7 (H)  a FIR wave and an FFT wave per long sliding run    (M = 56 default)
//   8 (C') channel-major only: short sliding runs, each chunk transposed in its LDS buffer, the workgroup's
//          tile written as 256-512-byte runs per channel       (channel-major default of the single-wave plans)
//   9      channel-major only, host side (pfb_api.cpp): frame-major slabs + pfb_transpose_slab_kernel
//  11 (P)  A software-pipelined inside the wave: next chunk's FIR next to this chunk's first FFT pass, two LDS chunk
//          buffers, rows two chunks ahead                      (M = 128 D = 64 default)
//  13 (W)  independent workgroups of a few waves, a frame per wave and chunk, all passes of a frame by one wave
//          (variant 3 of M = 1024 int16)
// Numbers 1, 5, 10 and 12 were studies that lost to the above (persistent strided chunks, persistent wave pairs, the team
// kernel transposing through scratch tiles, the PDW screen fused into the last pass); their measurements are in
// DESIGN.md sections 5 and 9, their code is gone.
#pragma once

#include "pfb_common.h"

#include <utility>

namespace pfb {

// ---------------------------------------------------------------------------------
// Packed-fp32 complex arithmetic.  A complex value is one v2f (re, im) in an aligned VGPR
// pair, so every add / fma below is ONE v_pk_*_f32 issue (gfx950 issues a wave64 VALU op in
// 4 cycles whether it is scalar-fp32 or packed: packing halves the issue count, and the
// kernel is issue-bound long before it is flop-bound).

#define PFB_DEV static __device__ __forceinline__

typedef float v2f __attribute__((ext_vector_type(2)));

constexpr float kSqrtHalf = 0.70710678118654752f;
constexpr float kCosPi8 = 0.92387953251128674f;
constexpr float kSinPi8 = 0.38268343236508977f;

PFB_DEV v2f swp(v2f a) { return __builtin_shufflevector(a, a, 1, 0); }
PFB_DEV v2f splat(float s) { return (v2f){s, s}; }
PFB_DEV v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
// a * (c + j s): pk_mul + pk_fma (the swap and the broadcast ride on op_sel)
PFB_DEV v2f cmul(v2f a, float c, float s) { return fma2(swp(a), (v2f){-s, s}, a * splat(c)); }
// same with the twiddle held as ONE register pair w = (c, s): two instructions, no extra
// register for -s (neg_lo negates s for the real part only)
PFB_DEV v2f cmul_w(v2f a, v2f w) {
  v2f t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]"
      : "=v"(r) : "v"(a), "v"(w), "v"(t));
  return r;
}
// acc += x * h.lo / h.hi (tap broadcast to both halves by op_sel): two taps share one register pair,
// which the compiler will not do by itself (it materialises a splat pair per tap)
PFB_DEV void fma_tap_lo(v2f& acc, v2f x, v2f h, int& tok) {
  (void)tok;
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(x), "v"(h));
}
PFB_DEV void fma_tap_hi(v2f& acc, v2f x, v2f h, int& tok) {
  (void)tok;
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(x), "v"(h));
}
// the first tap of a chain: acc = x * h + 0 with the zero as the instruction's inline constant -- the same operation on
// the same values as an FMA into a zeroed register pair, without the v_mov_b64 that zeroed it (C per column and chunk)
PFB_DEV void fma_tap0_lo(v2f& acc, v2f x, v2f h) {
  asm("v_pk_fma_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(acc) : "v"(x), "v"(h));
}
PFB_DEV void fma_tap0_hi(v2f& acc, v2f x, v2f h) {
  asm("v_pk_fma_f32 %0, %1, %2, 0 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "=v"(acc) : "v"(x), "v"(h));
}
// The same FMAs as builtins: the broadcast is a shufflevector of the tap PAIR, which the backend folds into op_sel /
// op_sel_hi (checked in the ISA: no v_mov, one v_pk_fma_f32 per tap).  Inline asm hides the instruction from the
// scheduler: it clusters the FMAs of one accumulator, and on gfx950 the result of a packed-fp32 instruction cannot be read
// by the very next VALU instruction, so the hazard recognizer pads every such pair with an s_nop (118 per 250 FMAs in the
// cfg5 loop, 4 issue cycles each; 75 with the builtins, the rest sit in the FFT's cmul_w).  As builtins the scheduler
// interleaves the C accumulators itself -- at the price of longer live ranges: every other kernel spills with them
// (cfg2's pair kernel 6 registers, cfg3 36, the cfg4 teams 74), so only the software-pipelined cfg5 kernel takes them.
PFB_DEV void fma_tap_lo_b(v2f& acc, v2f x, v2f h) { acc = __builtin_elementwise_fma(x, __builtin_shufflevector(h, h, 0, 0), acc); }
PFB_DEV void fma_tap_hi_b(v2f& acc, v2f x, v2f h) { acc = __builtin_elementwise_fma(x, __builtin_shufflevector(h, h, 1, 1), acc); }
PFB_DEV v2f add_j(v2f a, v2f b) { return fma2(swp(b), (v2f){-1.f, 1.f}, a); }  // a + j b
PFB_DEV v2f sub_j(v2f a, v2f b) { return fma2(swp(b), (v2f){1.f, -1.f}, a); }  // a - j b

// Sync between the phases of one team.  A team that is the whole workgroup uses the workgroup
// barrier (a single-wave workgroup's barrier is free); single-wave teams inside a bigger workgroup
// only need program order within the wave: the LDS executes one wave's accesses in order, so the
// fences just stop the compiler from moving LDS accesses across the phase boundary.
template <bool WAVE_LOCAL>
PFB_DEV void team_sync() {
  if constexpr (WAVE_LOCAL) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  } else {
    __syncthreads();
  }
}

// one complex64 output element; nontemporal = streaming store (output is write-once)
PFB_DEV void store_c64(float2* dst, v2f v, int nontemporal) {
  if (nontemporal) __builtin_nontemporal_store(v, reinterpret_cast<v2f*>(dst));
  else *reinterpret_cast<v2f*>(dst) = v;
}

// ---------------------------------------------------------------------------------
// In-register N-point DFT, kernel e^{+j 2 pi n k / N}, natural order in and out.

// X[K] = E + W^K O,  X[K+N/2] = E - W^K O,   W = e^{+j 2 pi / N}, N <= 16
template <int N, int K>
PFB_DEV void butterfly(v2f& lo, v2f& hi, v2f e, v2f o) {
  static_assert(16 % N == 0, "small DFT sizes only");
  constexpr int q = K * (16 / N);  // sixteenths of a turn, 0..7
  static_assert(q >= 0 && q < 8, "only the upper half plane is needed");
  if constexpr (q == 0) {
    lo = e + o; hi = e - o;
  } else if constexpr (q == 4) {  // W = +j
    lo = add_j(e, o); hi = sub_j(e, o);
  } else {
    constexpr float c = (q == 1) ? kCosPi8 : (q == 2) ? kSqrtHalf : (q == 3) ? kSinPi8
                        : (q == 5) ? -kSinPi8 : (q == 6) ? -kSqrtHalf : -kCosPi8;
    constexpr float s = (q == 1) ? kSinPi8 : (q == 2) ? kSqrtHalf : (q == 3) ? kCosPi8
                        : (q == 5) ? kCosPi8 : (q == 6) ? kSqrtHalf : kSinPi8;
    const v2f t = cmul(o, c, s);
    lo = e + t; hi = e - t;
  }
}

template <int N> struct Dft;

template <> struct Dft<2> {
  PFB_DEV void run(v2f (&x)[2]) {
    const v2f a = x[0];
    x[0] = a + x[1]; x[1] = a - x[1];
  }
};

template <> struct Dft<4> {
  PFB_DEV void run(v2f (&x)[4]) {
    const v2f t0 = x[0] + x[2], t1 = x[0] - x[2], t2 = x[1] + x[3], t3 = x[1] - x[3];
    x[0] = t0 + t2; x[2] = t0 - t2;
    x[1] = add_j(t1, t3); x[3] = sub_j(t1, t3);
  }
};

// 7-point DFT (the reference's own band count is fs*1e-6 = 56 = 8 x 7, channelizer_example.m:29):
// pair n with 7-n, X[k] = A_k + j B_k, X[7-k] = A_k - j B_k with
// A_k = x0 + sum_n (x_n + x_{7-n}) cos(2 pi k n / 7),  B_k = sum_n (x_n - x_{7-n}) sin(2 pi k n / 7)
template <> struct Dft<7> {
  PFB_DEV void run(v2f (&x)[7]) {
    constexpr float c1 = 0.62348980185873353f, c2 = -0.22252093395631440f, c3 = -0.90096886790241913f;
    constexpr float s1 = 0.78183148246802981f, s2 = 0.97492791218182361f, s3 = 0.43388373911755812f;
    const v2f p1 = x[1] + x[6], p2 = x[2] + x[5], p3 = x[3] + x[4];
    const v2f d1 = x[1] - x[6], d2 = x[2] - x[5], d3 = x[3] - x[4];
    const v2f x0 = x[0];
    const v2f a1 = fma2(p3, splat(c3), fma2(p2, splat(c2), fma2(p1, splat(c1), x0)));
    const v2f a2 = fma2(p3, splat(c1), fma2(p2, splat(c3), fma2(p1, splat(c2), x0)));
    const v2f a3 = fma2(p3, splat(c2), fma2(p2, splat(c1), fma2(p1, splat(c3), x0)));
    const v2f b1 = fma2(d3, splat(s3), fma2(d2, splat(s2), d1 * splat(s1)));
    const v2f b2 = fma2(d3, splat(-s1), fma2(d2, splat(-s3), d1 * splat(s2)));
    const v2f b3 = fma2(d3, splat(s2), fma2(d2, splat(-s1), d1 * splat(s3)));
    x[0] = x0 + p1 + p2 + p3;
    x[1] = add_j(a1, b1); x[6] = sub_j(a1, b1);
    x[2] = add_j(a2, b2); x[5] = sub_j(a2, b2);
    x[3] = add_j(a3, b3); x[4] = sub_j(a3, b3);
  }
};

// 5- and 10-point DFTs: the reference's other band count is round(fs / 0.1e6) = 560 = 10 x 8 x 7
// (generate_channelized_training_iq.m:95-96).  Same pairing as the 7-point one.
template <> struct Dft<5> {
  PFB_DEV void run(v2f (&x)[5]) {
    constexpr float c1 = 0.30901699437494742f, c2 = -0.80901699437494742f;
    constexpr float s1 = 0.95105651629515357f, s2 = 0.58778525229247313f;
    const v2f p1 = x[1] + x[4], p2 = x[2] + x[3];
    const v2f d1 = x[1] - x[4], d2 = x[2] - x[3];
    const v2f x0 = x[0];
    const v2f a1 = fma2(p2, splat(c2), fma2(p1, splat(c1), x0));
    const v2f a2 = fma2(p2, splat(c1), fma2(p1, splat(c2), x0));
    const v2f b1 = fma2(d2, splat(s2), d1 * splat(s1));
    const v2f b2 = fma2(d2, splat(-s1), d1 * splat(s2));
    x[0] = x0 + p1 + p2;
    x[1] = add_j(a1, b1); x[4] = sub_j(a1, b1);
    x[2] = add_j(a2, b2); x[3] = sub_j(a2, b2);
  }
};

template <> struct Dft<10> {
  PFB_DEV void run(v2f (&x)[10]) {
    // W_10^k = e^{+j 2 pi k / 10}
    constexpr float c1 = 0.80901699437494742f, s1 = 0.58778525229247313f;
    constexpr float c2 = 0.30901699437494742f, s2 = 0.95105651629515357f;
    v2f e[5], o[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) { e[k] = x[2 * k]; o[k] = x[2 * k + 1]; }
    Dft<5>::run(e);
    Dft<5>::run(o);
    const v2f t1 = cmul(o[1], c1, s1), t2 = cmul(o[2], c2, s2), t3 = cmul(o[3], -c2, s2), t4 = cmul(o[4], -c1, s1);
    x[0] = e[0] + o[0]; x[5] = e[0] - o[0];
    x[1] = e[1] + t1;   x[6] = e[1] - t1;
    x[2] = e[2] + t2;   x[7] = e[2] - t2;
    x[3] = e[3] + t3;   x[8] = e[3] - t3;
    x[4] = e[4] + t4;   x[9] = e[4] - t4;
  }
};

// 3-, 6- and 12-point DFTs: band counts with a factor 3 (numBands = fs * 1e-6 at 12, 24, 30, 48, 96, 120 Msps,
// channelizer_example.m:29).  W_3 = e^{+j 2 pi / 3} = -1/2 + j sqrt(3)/2.
template <> struct Dft<3> {
  PFB_DEV void run(v2f (&x)[3]) {
    constexpr float s = 0.86602540378443865f;
    const v2f p = x[1] + x[2], d = x[1] - x[2];
    const v2f a = fma2(p, splat(-0.5f), x[0]), b = d * splat(s);
    x[0] = x[0] + p;
    x[1] = add_j(a, b);
    x[2] = sub_j(a, b);
  }
};

template <> struct Dft<6> {
  PFB_DEV void run(v2f (&x)[6]) {
    constexpr float s = 0.86602540378443865f;
    v2f e[3] = {x[0], x[2], x[4]}, o[3] = {x[1], x[3], x[5]};
    Dft<3>::run(e);
    Dft<3>::run(o);
    const v2f t1 = cmul(o[1], 0.5f, s), t2 = cmul(o[2], -0.5f, s);  // W_6^1, W_6^2
    x[0] = e[0] + o[0]; x[3] = e[0] - o[0];
    x[1] = e[1] + t1;   x[4] = e[1] - t1;
    x[2] = e[2] + t2;   x[5] = e[2] - t2;
  }
};

template <> struct Dft<12> {
  PFB_DEV void run(v2f (&x)[12]) {
    constexpr float s = 0.86602540378443865f;
    v2f e[6], o[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) { e[k] = x[2 * k]; o[k] = x[2 * k + 1]; }
    Dft<6>::run(e);
    Dft<6>::run(o);
    // W_12^k = e^{+j 2 pi k / 12}: (s, 1/2), (1/2, s), j, (-1/2, s), (-s, 1/2)
    const v2f t1 = cmul(o[1], s, 0.5f), t2 = cmul(o[2], 0.5f, s), t4 = cmul(o[4], -0.5f, s), t5 = cmul(o[5], -s, 0.5f);
    x[0] = e[0] + o[0];       x[6] = e[0] - o[0];
    x[1] = e[1] + t1;         x[7] = e[1] - t1;
    x[2] = e[2] + t2;         x[8] = e[2] - t2;
    x[3] = add_j(e[3], o[3]); x[9] = sub_j(e[3], o[3]);
    x[4] = e[4] + t4;         x[10] = e[4] - t4;
    x[5] = e[5] + t5;         x[11] = e[5] - t5;
  }
};

// 14 = 2 x 7 (560 = 14 x 10 x 4 keeps every non-final pass of the team kernel at one item per lane)
template <> struct Dft<14> {
  PFB_DEV void run(v2f (&x)[14]) {
    // W_14^k = e^{+j 2 pi k / 14}, k = 1..6
    constexpr float c1 = 0.90096886790241915f, s1 = 0.43388373911755812f;
    constexpr float c2 = 0.62348980185873359f, s2 = 0.78183148246802980f;
    constexpr float c3 = 0.22252093395631445f, s3 = 0.97492791218182362f;
    v2f e[7], o[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) { e[k] = x[2 * k]; o[k] = x[2 * k + 1]; }
    Dft<7>::run(e);
    Dft<7>::run(o);
    const v2f t1 = cmul(o[1], c1, s1), t2 = cmul(o[2], c2, s2), t3 = cmul(o[3], c3, s3);
    const v2f t4 = cmul(o[4], -c3, s3), t5 = cmul(o[5], -c2, s2), t6 = cmul(o[6], -c1, s1);
    x[0] = e[0] + o[0]; x[7] = e[0] - o[0];
    x[1] = e[1] + t1;   x[8] = e[1] - t1;
    x[2] = e[2] + t2;   x[9] = e[2] - t2;
    x[3] = e[3] + t3;   x[10] = e[3] - t3;
    x[4] = e[4] + t4;   x[11] = e[4] - t4;
    x[5] = e[5] + t5;   x[12] = e[5] - t5;
    x[6] = e[6] + t6;   x[13] = e[6] - t6;
  }
};

template <int N, int K>
struct DftCombine {
  PFB_DEV void run(v2f (&x)[N], const v2f (&e)[N / 2], const v2f (&o)[N / 2]) {
    butterfly<N, K>(x[K], x[K + N / 2], e[K], o[K]);
    if constexpr (K + 1 < N / 2) DftCombine<N, K + 1>::run(x, e, o);
  }
};

template <int N> struct Dft {
  PFB_DEV void run(v2f (&x)[N]) {
    v2f e[N / 2], o[N / 2];
#pragma unroll
    for (int k = 0; k < N / 2; ++k) { e[k] = x[2 * k]; o[k] = x[2 * k + 1]; }
    Dft<N / 2>::run(e);
    Dft<N / 2>::run(o);
    DftCombine<N, 0>::run(x, e, o);
  }
};

// ---------------------------------------------------------------------------------
// Kernel configuration

template <int M_, int P_, int D_, int CPT_, int FMT_, int C_, int NP_, int R0_, int R1_, int R2_, int RS0_,
          int RS1_, int RS2_, int FS_, bool PINGPONG_, int MIN_WAVES_, bool TW_TABLE_ = false, bool WAVE_FRAMES_ = false>
struct FastCfg {
  static constexpr bool WAVE_FRAMES = WAVE_FRAMES_;  // schedule W only: every wave transforms whole frames by itself
  static constexpr int M = M_, P = P_, D = D_, CPT = CPT_, FMT = FMT_, C = C_, NP = NP_;
  static constexpr int LANES = D / CPT;                  // threads that own columns
  static constexpr int NT = (LANES + 63) / 64 * 64;      // threads per workgroup (whole waves)
  static constexpr bool POW2 = (M & (M - 1)) == 0;
  static constexpr int W = M * P / D;  // window rows = taps per column
  static constexpr int OS = M / D;     // branches per column (1, or 2 when oversampled)
  static constexpr int FS = FS_;       // frame stride in LDS (complex elements)
  static constexpr bool PINGPONG = PINGPONG_;
  static constexpr bool TW_TABLE = TW_TABLE_;  // inter-pass twiddles re-read from the L1-resident table
                                               // every chunk instead of living in registers
  static constexpr int MIN_WAVES = MIN_WAVES_;
  static constexpr int R(int i) { return i == 0 ? R0_ : i == 1 ? R1_ : R2_; }
  static constexpr int RS(int i) { return i == 0 ? RS0_ : i == 1 ? RS1_ : RS2_; }
  static constexpr int S(int i) { int s = 1; for (int j = i + 1; j < NP; ++j) s *= R(j); return s; }
  static constexpr int K(int i) { int k = 1; for (int j = 0; j < i; ++j) k *= R(j); return k; }
  static constexpr int WP = (W + 3) / 4 * 4;  // taps per column padded to whole float4s
  static constexpr int TAPS_LANE_FLOATS = D * WP;  // per-column tap table built by init_tables
  // inter-pass twiddle table: per non-final pass S rows of R entries, rows padded to an even length so that every row
  // starts on a 16-byte boundary (odd radices -- 5, 7, 3 -- in front of the last pass)
  static constexpr int TWR(int i) { return R(i) + (R(i) & 1); }
  static constexpr int TW_OFF(int i) { int o = 0; for (int j = 0; j < i; ++j) o += S(j) * TWR(j); return o; }
  static constexpr int TW_LANE_ELEMS = TW_OFF(NP - 1) > 0 ? TW_OFF(NP - 1) : 1;  // inter-pass twiddle rows
  static constexpr int BUF = C * FS;   // one chunk buffer (complex elements)
  static constexpr int LDS_ELEMS = BUF * (PINGPONG ? 2 : 1);
  static_assert(D % CPT == 0, "columns split evenly over threads");
  static_assert(M % D == 0 && (M * P) % D == 0, "D divides M");
  static_assert(NP >= 2 && NP <= 3, "2 or 3 passes");
  static_assert(R0_ * R1_ * (NP_ == 3 ? R2_ : 1) == M_, "radices multiply to M");
  static_assert(R(0) * RS(0) <= FS && R(1) * RS(1) <= FS && (NP < 3 || R(2) * RS(2) <= FS), "frame fits");
  // in-place non-final passes need every read of the pass to precede every write: one iteration per
  // thread, and (multi-wave teams) a barrier between the reads and the writes
  static_assert(PINGPONG || WAVE_FRAMES || (C * (M / R(0)) <= NT && (NP < 3 || C * (M / R(1)) <= NT)),
                "multi-iteration non-final passes need ping-pong buffers");
};

// ---------------------------------------------------------------------------------

// Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 labels the XCD), each with its own L2.  mode 1:
// consecutive runs go to ONE XCD (XCD x walks the x-th eighth of the stream): a run's halo rows are its predecessor's
// last rows, read from that L2 -- eight sweeps through the stream.  mode G > 1: each XCD takes G consecutive runs at a
// time, the eight XCDs 8 G consecutive runs: one window sweeping the stream, G - 1 of G halos still inside an XCD (the
// blocks past the last whole group of 8 G stay where they are).  Bijective for any grid size.
PFB_DEV long long xcd_remap_block(long long blk, long long nb, int mode) {
  if (mode == 1) {
    const long long q = nb >> 3, r = nb & 7, xc = blk & 7;
    return (xc < r ? xc * (q + 1) : r * (q + 1) + (xc - r) * q) + (blk >> 3);
  }
  if (mode > 1) {
    const long long G = mode, span = 8 * G, base = (blk / span) * span;
    if (base + span <= nb) {
      const long long in = blk - base;
      return base + (in & 7) * G + (in >> 3);
    }
  }
  return blk;
}

// CM = channel-major output, out[k * out_ld + out_frame0 + m] (MATLAB's column-major F x M): its own
// instantiation, so the extra address arithmetic never costs the frame-major kernels a register.
// MS = fused abs() with the magnitudes staged in LDS (its own instantiation of the sliding-run kernel, like CM)
template <class K, bool CM = false, bool MS = false>
struct FastKernel {
  using ST = SampleT<K::FMT>;
  using raw_t = typename ST::raw_t;
  static constexpr int M = K::M, P = K::P, D = K::D, CPT = K::CPT, C = K::C, W = K::W, OS = K::OS, NT = K::NT;
  static constexpr int NW = W - 1 + C;  // window rows held in registers during a chunk
  // the window as a RING of NWP rows (NW rounded up to whole chunks): after PERIOD chunks every row is back in its
  // register, so a chunk loop unrolled PERIOD times indexes the window with compile-time constants and never moves it
  static constexpr int NWP = (NW + C - 1) / C * C, PERIOD = NWP / C;
  static constexpr bool kRingOk = PERIOD >= 2 && PERIOD <= 4;

  struct alignas(sizeof(raw_t) * CPT) RawVec { raw_t v[CPT]; };

  PFB_DEV v2f cvt(raw_t r) {
    float re, im;
    ST::cvt(r, re, im);
    return (v2f){re, im};
  }

  // Row r of the stream -> CPT raw samples for this thread.  r is uniform across the workgroup.
  // INTERIOR runs (every row inside `in`, aligned) take the unchecked vector load; runs that touch
  // the history, the end of the stream or a misaligned buffer take the checked per-sample path.
  template <bool INTERIOR>
  PFB_DEV void load_row(const KernelParams& p, const raw_t* run_ptr, long long r, long long r_rel, int c0,
                        raw_t (&raw)[CPT]) {
    if (K::LANES < NT && c0 >= D) {  // lanes beyond the last column (D not a multiple of 64)
#pragma unroll
      for (int cc = 0; cc < CPT; ++cc) raw[cc] = raw_t{};
      return;
    }
    if constexpr (INTERIOR) {
      const RawVec* vp = reinterpret_cast<const RawVec*>(run_ptr + r_rel * D + c0);
      RawVec v;
      if constexpr (sizeof(RawVec) == 4) {
        if (p.experiment & 1) {  // streaming (nontemporal) row loads
          const uint32_t u = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(vp));
          __builtin_memcpy(&v, &u, 4);
        } else {
          v = *vp;
        }
      } else {
        v = *vp;
      }
#pragma unroll
      for (int cc = 0; cc < CPT; ++cc) raw[cc] = v.v[cc];
    } else {
      if (r >= p.frames) {  // padding frames of a partial last chunk
#pragma unroll
        for (int cc = 0; cc < CPT; ++cc) raw[cc] = raw_t{};
        return;
      }
      const long long s0 = r * D + p.base;
      const raw_t* in = static_cast<const raw_t*>(p.in);
      const raw_t* hist = static_cast<const raw_t*>(p.hist);
#pragma unroll
      for (int cc = 0; cc < CPT; ++cc) {
        const long long s = s0 + c0 + cc;
        raw[cc] = (s >= 0) ? in[s] : hist[p.hist_samples + s];
      }
    }
  }

  // The C rows of a chunk.  2-byte samples, one column per lane, rows that are not whole cache lines (M = 56 int8:
  // 112-byte rows, 21 % of roofline with one 2-byte load per row): pairs of rows are fetched as ONE dword load --
  // lanes [0, D/2) take row r (two columns each), lanes [D/2, D) row r + 1 -- and, when the chunk is consumed, two
  // ds_bpermutes hand every lane its own column of both rows.  load_rows only issues the loads (they stay in flight
  // under the previous chunk's arithmetic like the ordinary row loads); finish_rows does the exchange.  (With
  // 128-byte rows, M = 64 int8, pairing measured +2 % complex, -10 % with fused abs(): off.)  Needs the run's rows on a 4-byte boundary.
  static constexpr bool kPairedRows = sizeof(raw_t) == 2 && CPT == 1 && C % 2 == 0 && D % 2 == 0 && D < 64 && NT == 64;
  struct RowFetch {
    uint32_t pw[C / 2 > 0 ? C / 2 : 1];
    bool paired;
  };

  template <bool INTERIOR>
  PFB_DEV void begin_rows(const raw_t* run_ptr, RowFetch& rf) {
    rf.paired = kPairedRows && INTERIOR && (reinterpret_cast<uintptr_t>(run_ptr) & 3) == 0;
  }

  template <bool INTERIOR>
  PFB_DEV void load_rows(const KernelParams& p, const raw_t* run_ptr, long long f_first, long long rel_first, int c0,
                         raw_t (&raw)[C][CPT], RowFetch& rf) {
    if constexpr (kPairedRows && INTERIOR) {
      if (rf.paired) {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int t = 0; t < C; t += 2) {
          const uint32_t* rp = reinterpret_cast<const uint32_t*>(run_ptr + (rel_first + t) * D);
          rf.pw[t / 2] = lane < D ? rp[lane] : 0u;
        }
        return;
      }
    }
#pragma unroll
    for (int t = 0; t < C; ++t) load_row<INTERIOR>(p, run_ptr, f_first + t, rel_first + t, c0, raw[t]);
  }

  PFB_DEV void finish_rows(int c0, raw_t (&raw)[C][CPT], const RowFetch& rf) {
    if constexpr (kPairedRows) {
      if (rf.paired) {
        const int src = (c0 < D ? c0 : 0) >> 1, sh = (c0 & 1) * 16;
#pragma unroll
        for (int t = 0; t < C; t += 2) {
          const uint32_t a = (uint32_t)__builtin_amdgcn_ds_bpermute(src * 4, (int)rf.pw[t / 2]);
          const uint32_t b = (uint32_t)__builtin_amdgcn_ds_bpermute((D / 2 + src) * 4, (int)rf.pw[t / 2]);
          raw[t][0] = c0 < D ? (raw_t)((a >> sh) & 0xffffu) : raw_t{};
          raw[t + 1][0] = c0 < D ? (raw_t)((b >> sh) & 0xffffu) : raw_t{};
        }
      }
    }
  }

  // FULL: every frame of the chunk exists (an interior run): the stores are unconditional, so that the number of
  // vector-memory operations per step is the same on every path -- the compiler's s_waitcnt counts stay exact across
  // the chunk loop (a conditional store makes it assume the fewest, i.e. wait for MORE than the load it needs)
  // MAGSEL: -1 = PFB_FLAG_MAGNITUDE is tested here, 0 / 1 = the caller has (outside its chunk loop: same reason as FULL)
  template <int I, bool FULL = false, int MAGSEL = -1>
  PFB_DEV void pass(const KernelParams& p, float2* src, float2* dst, int tid, long long f0,
                    const v2f (&tw)[2][16]) {
    float2* const p_out = p.out;
    const long long p_frames = p.frames;
    constexpr int R = K::R(I), S = K::S(I), KK = K::K(I), RS = K::RS(I);
    constexpr int IPF = M / R, ITEMS = C * IPF, ITERS = (ITEMS + NT - 1) / NT;
    constexpr bool LAST = (I == K::NP - 1);
    constexpr bool TW_REGS = (ITERS == 1) && !K::TW_TABLE;
    constexpr bool READ_BARRIER = !LAST && !K::PINGPONG && NT > 64;  // in place across several waves
    constexpr bool kMagStaged = MS && LAST;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int w = tid + it * NT;
      const bool active = (ITEMS % NT == 0) || (w < ITEMS);
      const int fc = w / IPF, item = w % IPF;
      const int kk = item / S, rest = item % S;
      v2f x[R];
      if (active) {
        const v2f* s2 = reinterpret_cast<const v2f*>(src) + fc * K::FS + item;
#pragma unroll
        for (int n = 0; n < R; ++n) x[n] = s2[n * RS];
      } else {
#pragma unroll
        for (int n = 0; n < R; ++n) x[n] = (v2f){0.f, 0.f};
      }
      if constexpr (READ_BARRIER) __syncthreads();
      Dft<R>::run(x);
      if constexpr (!LAST) {
        constexpr int S1 = K::S(I + 1), RS1 = K::RS(I + 1);
        // twiddle e^{+j 2 pi rest k / (R S)}: row `rest` of this pass's table
        if constexpr (TW_REGS) {
#pragma unroll
          for (int k = 1; k < R; ++k) x[k] = cmul_w(x[k], tw[I][k]);
        } else {
          const float4* t4 = reinterpret_cast<const float4*>(p.tw_lane + K::TW_OFF(I) + rest * K::TWR(I));
#pragma unroll
          for (int k2 = 0; k2 < K::TWR(I) / 2; ++k2) {
            const float4 t = t4[k2];
            if (k2 > 0) x[2 * k2] = cmul_w(x[2 * k2], (v2f){t.x, t.y});
            if (2 * k2 + 1 < R) x[2 * k2 + 1] = cmul_w(x[2 * k2 + 1], (v2f){t.z, t.w});
          }
        }
        const int n1 = rest / S1, rest2 = rest % S1;
        if (active) {
          v2f* d2 = reinterpret_cast<v2f*>(dst) + fc * K::FS + n1 * RS1 + kk * S1 + rest2;
#pragma unroll
          for (int k = 0; k < R; ++k) d2[k * KK * S1] = x[k];
        }
      } else {
        if constexpr (kMagStaged) {
          // fused abs() of the single-wave M = 64 kernels: a lane's 8 magnitudes are 8 channels apart, so storing them
          // directly writes 32-byte pieces.  The chunk buffer is free once the wave has read it (the LDS executes a
          // wave's accesses in order), so the magnitudes go there as rows of M floats (+8 pad: the 8 frames land on
          // distinct banks) and leave as 16 bytes per lane: 4 frames x 256 contiguous bytes per instruction.
          // (the sliding-run kernel only: there it is worth 7 %, 2.07 -> 1.93 ms per 2^30 samples, and makes sliding
          // runs the fastest way to magnitudes; on the FFT wave of the pair schedules the extra LDS trip costs 2 %.
          // launch_fast picks this instantiation when the flag is set and `out` is 16-byte aligned.)
          static_assert(!CM && NT == 64 && ITERS == 1 && K::NP == 2 && !K::PINGPONG && M % 4 == 0, "single-wave two-pass plans");
          {
            constexpr int SR = M + ((8 - M % 64) + 64) % 64;  // = 8 (mod 64), a multiple of 4
            static_assert(C * SR * sizeof(float) <= K::BUF * sizeof(float2), "the staged magnitudes fit the chunk buffer");
            float* stage = reinterpret_cast<float*>(src);
            const int shift = (p.flags & PFB_FLAG_FFTSHIFT) ? (M / 2) : 0;
            team_sync<true>();
#pragma unroll
            for (int k = 0; k < R; ++k) {
              int col = kk + k * KK + shift;
              col = col >= M ? col - M : col;
              if (active) stage[fc * SR + col] = mag_out(x[k].x, x[k].y, p.flags);
            }
            team_sync<true>();
            constexpr int NV = C * M / 4;  // float4s in the chunk
#pragma unroll
            for (int j = 0; j < (NV + 63) / 64; ++j) {
              const int idx = tid + 64 * j, fr = idx / (M / 4), q = idx % (M / 4);
              if ((NV % 64 == 0 || idx < NV) && f0 + fr < p_frames) {
                const float4 v = *reinterpret_cast<const float4*>(stage + fr * SR + q * 4);
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(p_out) + (f0 + fr) * M + q * 4) = v;
              }
            }
            return;
          }
        }
        const long long f = f0 + fc;
        if (active && (FULL || f < p_frames)) {
          const bool flip_odd = (OS == 2) && (p.flags & PFB_FLAG_DEROTATE) && ((p.frame0 + f) & 1);
          // derotation of the 2x oversampled bank = a sign on the odd channels of odd frames.  As ONE multiply by a
          // per-lane +-1 (exact, -0 included); `if (flip) v = -v` per store compiled to a negate, a nop and four
          // v_cndmask in front of every store, flag set or not: 14 % of the cfg5 kernel's VALU instructions
          const v2f sg0 = splat((flip_odd && (kk & 1)) ? -1.f : 1.f), sg1 = splat((flip_odd && ((kk + KK) & 1)) ? -1.f : 1.f);
          auto derot = [&](v2f v, int k) { return OS == 2 ? v * (((k * KK) & 1) ? sg1 : sg0) : v; };
          const int shift = (p.flags & PFB_FLAG_FFTSHIFT) ? (M / 2) : 0;
          // fftshift(out,2): column (k + M/2) mod M.  For a power-of-two M that swaps the two halves of
          // the row, and since KK * R == M the butterfly outputs k < R/2 land in one half and k >= R/2 in
          // the other: two base pointers plus compile-time offsets instead of one address per store.
          auto col_of = [&](int ch) {
            const int c2 = ch + shift;
            return c2 >= M ? c2 - M : c2;
          };
          auto slot = [&](auto* rowp, int k) {
            if constexpr (K::POW2) {
              auto* lo = rowp + kk + shift;
              auto* hi = rowp + kk + (M / 2 - shift);
              return (k < R / 2) ? lo + k * KK : hi + (k - R / 2) * KK;
            } else {
              return rowp + col_of(kk + k * KK);
            }
          };
          if constexpr (CM || (!K::POW2 && K::NT == 64)) {
            // Channel-major: a frame-chunk's C frames of a channel are C consecutive elements, so a store
            // instruction still fills whole 32/64-byte runs (its lanes differ in fc).  The R addresses are
            // K columns apart (wrapping at M under fftshift); they are produced one at a time -- the opaque
            // asm keeps the compiler from materialising all R 64-bit addresses ahead of the butterfly,
            // which spilled.  The single-wave frame-major kernels with a non-power-of-two M (56) take the same
            // route with column stride 1 (+5 %); the 576-thread M=560 kernel measured 7 % slower that way.
            const bool mag = (p.flags & PFB_FLAG_MAGNITUDE) != 0;
            const long long esz = mag ? 4 : 8;
            const long long cs = CM ? p.out_ld : 1;  // elements between adjacent channels
            int col = col_of(kk);
            char* ptr = reinterpret_cast<char*>(p_out) + ((long long)col * cs + (CM ? p.out_frame0 + f : f * M)) * esz;
            const long long step = (long long)KK * cs * esz, wrap = (long long)M * cs * esz;
#pragma unroll
            for (int k = 0; k < R; ++k) {
              v2f v = x[k];
              if (mag) {
                *reinterpret_cast<float*>(ptr) = mag_out(v.x, v.y, p.flags);
              } else {
                v = derot(v, k);
                store_c64(reinterpret_cast<float2*>(ptr), v, p.nontemporal);
              }
              asm volatile("" : "+v"(ptr) : : "memory");
              col += KK;
              ptr += step;
              if (col >= M) { col -= M; ptr -= wrap; }
            }
          } else if (MAGSEL == 1 || (MAGSEL < 0 && (p.flags & PFB_FLAG_MAGNITUDE))) {  // fused abs(): 4 bytes per channel instead of 8
            float* rowm = reinterpret_cast<float*>(p_out) + f0 * M + fc * M;
#pragma unroll
            for (int k = 0; k < R; ++k) *slot(rowm, k) = mag_out(x[k].x, x[k].y, p.flags);
          } else {
            float2* row = p_out + f0 * M + fc * M;
            // (probed: R/2 16-byte stores per lane instead of R 8-byte ones -- same bytes, half the store instructions,
            // written in a wrong layout just for the timing -- gain 0.4 % cfg2, 1.5 % cfg5, 2 % cfg4, 2.7 % cfg3 BEFORE the
            // lane exchange a correct layout needs (4 DPP moves per pair): not store-issue-bound, left alone)
#pragma unroll
            for (int k = 0; k < R; ++k) {
              const v2f v = derot(x[k], k);
              store_c64(slot(row, k), v, p.nontemporal);
            }
          }
        }
      }
    }
  }

  // The last pass of a single-wave kernel with its outputs left in LDS instead of stored: the chunk's buffer is
  // overwritten in place by the chunk TRANSPOSED, slot[column * C + frame] (fftshift and the derotation sign
  // applied), for run_tile_t's channel-major flush.  All of the wave's reads are issued before its first write
  // (the LDS executes one wave's accesses in order), so no second buffer is needed.
  // Where frame fc of column col sits inside the column's C-frame group of a transposed slot.  The LDS serves a
  // ds_write_b64 sixteen lanes at a time over 32 banks, i.e. over float2 addresses mod 16; the sixteen lanes of a
  // last-pass store are min(16, M / R) adjacent columns x the rest in frames, and col * C + fc puts columns
  // 16 / C apart on the same banks (rocprofv3: SQ_LDS_BANK_CONFLICT 0.19 cycles per sample, a 4-way conflict on
  // every store).  XOR-ing the column's higher bits into the frame index gives the sixteen lanes sixteen different
  // addresses mod 16; the flush reads whole groups per column, so the permutation inside a group costs it nothing
  // (counter after: 0).
  PFB_DEV int tslot_frame(int col, int fc) {
    constexpr int IPF = M / K::R(K::NP - 1), NFC = IPF >= 16 ? 1 : 16 / IPF, Q = 16 / C;
    static_assert(16 % C == 0 && (IPF >= 16 || 16 % IPF == 0) && C % NFC == 0, "power-of-two chunk and lane groups");
    return fc ^ (NFC * ((col / Q) % (C / NFC)));
  }

  PFB_DEV void last_pass_transposed(const KernelParams& p, float2* slot, int tid, long long f0) {
    constexpr int I = K::NP - 1;
    constexpr int R = K::R(I), KK = K::K(I), RS = K::RS(I);
    constexpr int IPF = M / R, ITEMS = C * IPF, ITERS = (ITEMS + NT - 1) / NT;
    static_assert(NT == 64 && !K::PINGPONG && K::S(I) == 1 && M * C <= K::LDS_ELEMS, "wave-local, in place");
    v2f x[ITERS][R];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int w = tid + it * NT;
      const bool active = (ITEMS % NT == 0) || (w < ITEMS);
      const int fc = w / IPF, item = w % IPF;
      const v2f* s2 = reinterpret_cast<const v2f*>(slot) + fc * K::FS + item;
#pragma unroll
      for (int n = 0; n < R; ++n) x[it][n] = active ? s2[n * RS] : (v2f){0.f, 0.f};
    }
    team_sync<true>();
    const int shift = (p.flags & PFB_FLAG_FFTSHIFT) ? (M / 2) : 0;
    v2f* t2 = reinterpret_cast<v2f*>(slot);
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int w = tid + it * NT;
      const bool active = (ITEMS % NT == 0) || (w < ITEMS);
      const int fc = w / IPF, kk = w % IPF;
      Dft<R>::run(x[it]);
      if (active) {
        const bool flip_odd = (OS == 2) && (p.flags & PFB_FLAG_DEROTATE) && ((p.frame0 + f0 + fc) & 1);
        const v2f sg0 = splat((flip_odd && (kk & 1)) ? -1.f : 1.f), sg1 = splat((flip_odd && ((kk + KK) & 1)) ? -1.f : 1.f);
        int col = kk + shift;
        if (col >= M) col -= M;
#pragma unroll
        for (int k = 0; k < R; ++k) {
          v2f v = x[it][k];
          if constexpr (OS == 2) v = v * (((k * KK) & 1) ? sg1 : sg0);  // derotation sign, one multiply (see pass<>)
          t2[col * C + tslot_frame(col, fc)] = v;
          col += KK;
          if (col >= M) col -= M;
        }
      }
    }
  }

  // ---- per-thread constants shared by both schedules -------------------------------------------
  struct Consts {
    v2f hp[(W + 1) / 2][CPT];  // taps of this thread's columns, two per register pair
    v2f tw[2][16];             // inter-pass twiddles (c, s)
    int upos[OS][CPT];         // LDS position of the FIR outputs inside a frame (pass-0 layout)
    v2f conj_mul;
  };

  PFB_DEV float tap(const Consts& k, int j, int cc) { return (j & 1) ? k.hp[j >> 1][cc].y : k.hp[j >> 1][cc].x; }

  // Per-thread constants come from two small L2-resident tables laid out for 16-byte loads (built once
  // per handle by init_tables): taps_lane[c][0..WP) = h[(D-1-c) + D*j] and, per non-final pass,
  // tw_lane[rest][k] = e^{+j 2 pi rest k / (R S)}.  A wave needs WP/4 + R/2 wide loads instead of
  // W + R-1 narrow ones, which is what makes short-lived workgroups affordable.
  PFB_DEV void setup(const KernelParams& p, int tid, Consts& k) {
    const int c0 = tid * CPT;
#pragma unroll
    for (int cc = 0; cc < CPT; ++cc) {
      const int col = (K::LANES < NT && c0 >= D) ? 0 : c0 + cc;  // idle lanes read column 0's taps
      const float4* tl = reinterpret_cast<const float4*>(p.taps_lane + (size_t)col * K::WP);
#pragma unroll
      for (int q4 = 0; q4 < K::WP / 4; ++q4) {
        const float4 v = tl[q4];
        if (2 * q4 < (W + 1) / 2) k.hp[2 * q4][cc] = (v2f){v.x, v.y};
        if (2 * q4 + 1 < (W + 1) / 2) k.hp[2 * q4 + 1][cc] = (v2f){v.z, v.w};
      }
    }
    k.conj_mul = (v2f){1.f, (p.flags & PFB_FLAG_CONJUGATE_INPUT) ? -1.f : 1.f};
#pragma unroll
    for (int i = 0; i < K::NP - 1; ++i) {
      constexpr int dummy = 0; (void)dummy;
      const int R = K::R(i), S = K::S(i), IPF = M / R;
      if (C * IPF <= NT && !K::TW_TABLE) {
        const int rest = (tid % IPF) % S;
        const float4* t4 = reinterpret_cast<const float4*>(p.tw_lane + K::TW_OFF(i) + rest * K::TWR(i));
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) {
          if (2 * k2 < R) {
            const float4 v = t4[k2];
            k.tw[i][2 * k2] = (v2f){v.x, v.y};
            k.tw[i][2 * k2 + 1] = (v2f){v.z, v.w};  // (the pad entry of an odd row: never used)
          }
        }
      }
    }
#pragma unroll
    for (int ph = 0; ph < OS; ++ph)
#pragma unroll
      for (int cc = 0; cc < CPT; ++cc) {
        const int n = (D - 1 - (c0 + cc)) + D * ph;
        k.upos[ph][cc] = (n / K::S(0)) * K::RS(0) + (n % K::S(0));
      }
  }

  // FIR of C frames from the window x (x[i] = row f0-(W-1)+i) into LDS, then the FFT passes and the
  // stores.  u_{p_lo + D ph}[t] = sum_q h[ph + OS q] * x[row t - ph - OS q]: one v_pk_fma_f32 per tap.
  template <bool WAVE_LOCAL = false, bool TRANSPOSED = false, bool FULL = false, int MAGSEL = -1>
  PFB_DEV void fir_fft_store(const KernelParams& p, const Consts& k, const v2f (&x)[NW][CPT], float2* lds, int tid,
                             long long f0) {
    float2* buf0 = lds;
    float2* buf1 = K::PINGPONG ? lds + K::BUF : lds;
#pragma unroll
    for (int ph = 0; ph < OS; ++ph)
#pragma unroll
      for (int cc = 0; cc < CPT; ++cc) {
        v2f acc[C];  // C independent chains: tap-major order keeps dependent pk_fma's C issues apart
        int tok = 0;  // FMA ordering token (fma_tap_lo)
#pragma unroll
        for (int q = 0; q < P; ++q) {
          const int j = ph + OS * q;
#pragma unroll
          for (int t = 0; t < C; ++t) {
            if (q == 0 && (j & 1)) fma_tap0_hi(acc[t], x[W - 1 + t - j][cc], k.hp[j >> 1][cc]);
            else if (q == 0) fma_tap0_lo(acc[t], x[W - 1 + t - j][cc], k.hp[j >> 1][cc]);
            else if (j & 1) fma_tap_hi(acc[t], x[W - 1 + t - j][cc], k.hp[j >> 1][cc], tok);
            else fma_tap_lo(acc[t], x[W - 1 + t - j][cc], k.hp[j >> 1][cc], tok);
          }
        }
        if (!(K::LANES < NT) || tid < K::LANES) {
#pragma unroll
          for (int t = 0; t < C; ++t)
            reinterpret_cast<v2f*>(buf0)[t * K::FS + k.upos[ph][cc]] = acc[t] * k.conj_mul;
        }
      }
    team_sync<WAVE_LOCAL>();
    pass<0>(p, buf0, buf1, tid, f0, k.tw);
    team_sync<WAVE_LOCAL>();
    if constexpr (TRANSPOSED) {
      static_assert(K::NP == 2 && WAVE_LOCAL, "single-wave two-pass plans");
      last_pass_transposed(p, buf1, tid, f0);
      return;
    } else if constexpr (K::NP == 2) {
      pass<1, FULL, MAGSEL>(p, buf1, nullptr, tid, f0, k.tw);
    } else {
      pass<1>(p, buf1, buf0, tid, f0, k.tw);
      team_sync<WAVE_LOCAL>();
      pass<2, FULL, MAGSEL>(p, buf0, nullptr, tid, f0, k.tw);
    }
    team_sync<WAVE_LOCAL>();  // the next chunk's FIR overwrites buf0
  }

  // the two halves of fir_fft_store as separate steps (schedule F gives them to different waves)
  // A thread's two adjacent columns c0, c0 + 1 are the adjacent branch outputs n0 = D-1-c0 (odd) and n0 - 1 (even) of
  // a frame, and where pass 0's rows are unpadded (RS_0 = S_0) they are adjacent in LDS: ONE 16-byte write per frame
  // instead of two 8-byte ones whose lanes sit 16 bytes apart (a 2-way bank conflict: 15 % of the M = 1024 team kernel's
  // LDS cycles, 19 % at M = 560).
  static constexpr bool kPairWrite = CPT == 2 && K::RS(0) == K::S(0) && K::S(0) % 2 == 0 && D % 2 == 0 && K::FS % 2 == 0 &&
                                     K::BUF % 2 == 0;
  PFB_DEV void fir_to_lds(const Consts& k, const v2f (&x)[NW][CPT], float2* buf, int tid) {
    if constexpr (kPairWrite) {
#pragma unroll
      for (int ph = 0; ph < OS; ++ph) {
        v2f acc[2][C];
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
          int tok = 0;  // FMA ordering token (fma_tap_lo)
#pragma unroll
          for (int q = 0; q < P; ++q) {
            const int j = ph + OS * q;
#pragma unroll
            for (int t = 0; t < C; ++t) {
              if (q == 0 && (j & 1)) fma_tap0_hi(acc[cc][t], x[W - 1 + t - j][cc], k.hp[j >> 1][cc]);
              else if (q == 0) fma_tap0_lo(acc[cc][t], x[W - 1 + t - j][cc], k.hp[j >> 1][cc]);
              else if (j & 1) fma_tap_hi(acc[cc][t], x[W - 1 + t - j][cc], k.hp[j >> 1][cc], tok);
              else fma_tap_lo(acc[cc][t], x[W - 1 + t - j][cc], k.hp[j >> 1][cc], tok);
            }
          }
        }
        if (!(K::LANES < NT) || tid < K::LANES) {
#pragma unroll
          for (int t = 0; t < C; ++t) {
            const v2f lo = acc[1][t] * k.conj_mul, hi = acc[0][t] * k.conj_mul;  // positions n0 - 1, n0
            *reinterpret_cast<float4*>(buf + t * K::FS + k.upos[ph][1]) = make_float4(lo.x, lo.y, hi.x, hi.y);
          }
        }
      }
      return;
    }
#pragma unroll
    for (int ph = 0; ph < OS; ++ph)
#pragma unroll
      for (int cc = 0; cc < CPT; ++cc) {
        v2f acc[C];
        int tok = 0;  // FMA ordering token (fma_tap_lo)
#pragma unroll
        for (int q = 0; q < P; ++q) {
          const int j = ph + OS * q;
#pragma unroll
          for (int t = 0; t < C; ++t) {
            if (q == 0 && (j & 1)) fma_tap0_hi(acc[t], x[W - 1 + t - j][cc], k.hp[j >> 1][cc]);
            else if (q == 0) fma_tap0_lo(acc[t], x[W - 1 + t - j][cc], k.hp[j >> 1][cc]);
            else if (j & 1) fma_tap_hi(acc[t], x[W - 1 + t - j][cc], k.hp[j >> 1][cc], tok);
            else fma_tap_lo(acc[t], x[W - 1 + t - j][cc], k.hp[j >> 1][cc], tok);
          }
        }
        if (!(K::LANES < NT) || tid < K::LANES) {
#pragma unroll
          for (int t = 0; t < C; ++t)
            reinterpret_cast<v2f*>(buf)[t * K::FS + k.upos[ph][cc]] = acc[t] * k.conj_mul;
        }
      }
  }

  PFB_DEV void fft_from_lds(const KernelParams& p, const Consts& k, float2* buf, int tid, long long f0) {
    static_assert(K::NP == 2 && !K::PINGPONG, "two in-place passes");
    pass<0>(p, buf, buf, tid, f0, k.tw);
    team_sync<true>();
    pass<1>(p, buf, nullptr, tid, f0, k.tw);
  }

  // ---- schedule A: sliding window over a long contiguous run per workgroup ---------------------
  // (MAGSEL / interior runs: the loop issues the same vector-memory operations on every path and is rotated -- the next
  // chunk's rows, requested before this chunk's FIR, are taken at the END of the iteration -- so that the compiler's
  // s_waitcnt for them counts the chunk's stores exactly instead of waiting for them too: see pass<FULL>)
  template <bool INTERIOR, int MAGSEL = -1>
  PFB_DEV void run_impl(const KernelParams& p, const Consts& k, float2* lds, long long f_begin, long long f_end) {
    const int tid = threadIdx.x;
    const int c0 = tid * CPT;
    // uniform pointer to (row f_begin-(W-1), column 0); only dereferenced on the INTERIOR path
    const raw_t* run_ptr = static_cast<const raw_t*>(p.in) + ((f_begin - (W - 1)) * D + p.base);
    v2f x[NW][CPT];
    raw_t raw[C][CPT];
#pragma unroll
    for (int i = 0; i < W - 1; ++i) {
      raw_t t[CPT];
      load_row<INTERIOR>(p, run_ptr, f_begin - (W - 1) + i, i, c0, t);
#pragma unroll
      for (int cc = 0; cc < CPT; ++cc) x[i][cc] = cvt(t[cc]);
    }
    RowFetch rf;
    begin_rows<INTERIOR>(run_ptr, rf);
    load_rows<INTERIOR>(p, run_ptr, f_begin, W - 1, c0, raw, rf);
    auto take_rows = [&]() {
      finish_rows(c0, raw, rf);
#pragma unroll
      for (int t = 0; t < C; ++t)
#pragma unroll
        for (int cc = 0; cc < CPT; ++cc) x[W - 1 + t][cc] = cvt(raw[t][cc]);
    };
    take_rows();
    for (long long f0 = f_begin; f0 < f_end; f0 += C) {
      if constexpr (INTERIOR) {  // the next chunk's rows under this chunk's FFT; past the run's end: its last chunk again
        const long long nxt = f0 + C < f_end ? f0 + C : f0;
        load_rows<true>(p, run_ptr, nxt, (nxt - f_begin) + (W - 1), c0, raw, rf);
      } else if (f0 + C < f_end) {
        const long long rel = (f0 - f_begin) + C + (W - 1);
        load_rows<INTERIOR>(p, run_ptr, f0 + C, rel, c0, raw, rf);
      }
      fir_fft_store<false, false, INTERIOR, MAGSEL>(p, k, x, lds, tid, f0);
      // slide the window by C rows
#pragma unroll
      for (int i = 0; i < W - 1; ++i)
#pragma unroll
        for (int cc = 0; cc < CPT; ++cc) x[i][cc] = x[i + C][cc];
      if (INTERIOR || f0 + C < f_end) take_rows();
    }
  }

  // (The same loop with the window as a ring -- PERIOD chunks per iteration, no slide, see run_overlap_ring -- measured on
  // cfg3: 0.676-0.686 either way at 24- and 36-frame runs; not kept here.)
  template <int MAGSEL = -1>
  PFB_DEV void run(const KernelParams& p, float2* lds) {
    // Consecutive runs go to one XCD (blocks are dealt round-robin over the 8 XCDs, so bid%8 labels
    // the XCD).  Bijective for any grid size.
    long long run = blockIdx.x;
    run = xcd_remap_block(run, gridDim.x, p.xcd_remap);
    const long long f_begin = run * p.frames_per_block;
    if (f_begin >= p.frames) return;
    const long long f_last = f_begin + p.frames_per_block;
    const long long f_end = f_last < p.frames ? f_last : p.frames;
    Consts k;
    setup(p, threadIdx.x, k);
    // every row of the run (halo included) lies inside `in`, whole chunks only, aligned vectors
    const bool interior = p.vec_ok && ((f_begin - (W - 1)) * D + p.base >= 0) && (f_last <= p.frames);
    if (interior) run_impl<true, MAGSEL>(p, k, lds, f_begin, f_end);
    else run_impl<false, MAGSEL>(p, k, lds, f_begin, f_end);
  }

  // ---- schedule P: schedule A software-pipelined inside the wave ---------------------------------------------------
  // PMC of the sliding kernels (profiles/r02_rocprofv3_pmc_summary_all_shapes.txt): a wave spends 26-29 % of its cycles in
  // VALU instructions and ~40 % parked on s_waitcnt -- the chunk is a dependent chain FIR -> LDS -> pass 0 -> LDS -> pass 1 ->
  // stores, and at 2 waves per SIMD (the window lives in ~200 VGPRs) nothing else is there to issue meanwhile.  Here the
  // FIR of chunk i + 1 (pure VALU on the register window) sits in the SAME basic block as pass 0 of chunk i (LDS reads,
  // butterflies, LDS writes on the other of two chunk buffers), so the compiler's scheduler can fill the LDS round trips
  // with the next chunk's FMAs; the branch outputs wait in registers and go to LDS after pass 1.  Rows are fetched two
  // chunks ahead instead of one.  Same arithmetic per output as schedule A: bit-identical.
  static constexpr bool kBuiltinFir = OS == 2 && CPT == 1;  // cfg5's shape: no spills with the scheduler-visible FMAs (see fma_tap_lo_b)
  // (PH / NX: the window as a ring of NX rows -- logical row i is x[(i + C PH) % NX] -- for run_overlap_ring; the sliding
  // callers pass the NW logical rows themselves, PH = 0)
  template <int PH = 0, int NX = NW>
  PFB_DEV void fir_compute(const Consts& k, const v2f (&xw)[NX][CPT], v2f (&acc)[OS][CPT][C]) {
    auto x = [&](int i) -> const v2f (&)[CPT] { return xw[(i + C * PH) % NX]; };
#pragma unroll
    for (int ph = 0; ph < OS; ++ph)
#pragma unroll
      for (int cc = 0; cc < CPT; ++cc) {
        if constexpr (kBuiltinFir) {
#pragma unroll
          for (int t = 0; t < C; ++t) acc[ph][cc][t] = (v2f){0.f, 0.f};
        }
        int tok = 0;  // FMA ordering token (fma_tap_lo)
#pragma unroll
        for (int q = 0; q < P; ++q) {
          const int j = ph + OS * q;
#pragma unroll
          for (int t = 0; t < C; ++t) {
            if constexpr (kBuiltinFir) {
              if (j & 1) fma_tap_hi_b(acc[ph][cc][t], x(W - 1 + t - j)[cc], k.hp[j >> 1][cc]);
              else fma_tap_lo_b(acc[ph][cc][t], x(W - 1 + t - j)[cc], k.hp[j >> 1][cc]);
            } else {
              if (q == 0 && (j & 1)) fma_tap0_hi(acc[ph][cc][t], x(W - 1 + t - j)[cc], k.hp[j >> 1][cc]);
              else if (q == 0) fma_tap0_lo(acc[ph][cc][t], x(W - 1 + t - j)[cc], k.hp[j >> 1][cc]);
              else if (j & 1) fma_tap_hi(acc[ph][cc][t], x(W - 1 + t - j)[cc], k.hp[j >> 1][cc], tok);
              else fma_tap_lo(acc[ph][cc][t], x(W - 1 + t - j)[cc], k.hp[j >> 1][cc], tok);
            }
          }
        }
      }
  }

  PFB_DEV void fir_write(const Consts& k, const v2f (&acc)[OS][CPT][C], float2* buf, int tid) {
    if (!(K::LANES < NT) || tid < K::LANES) {
#pragma unroll
      for (int ph = 0; ph < OS; ++ph)
#pragma unroll
        for (int cc = 0; cc < CPT; ++cc)
#pragma unroll
          for (int t = 0; t < C; ++t) reinterpret_cast<v2f*>(buf)[t * K::FS + k.upos[ph][cc]] = acc[ph][cc][t] * k.conj_mul;
    }
  }

  // MAGSEL: PFB_FLAG_MAGNITUDE as a template parameter of the kernel, and (interior runs) unconditional stores: the
  // number of vector-memory operations per chunk is then the same on every path, and the compiler's s_waitcnt for the
  // rows requested two chunks ahead stops waiting for half of the previous chunk's stores as well (pass<FULL>)
  // (tried for cfg3, whose 4 columns per lane spill 16-26 registers inside this loop: pass 0's twiddles from an LDS copy
  // instead of 30 registers -- the spills stayed, the rate fell from 0.64 to 0.50; cfg3 went back to schedule 0)
  template <bool INTERIOR, int MAGSEL = -1>
  PFB_DEV void run_overlap_impl(const KernelParams& p, const Consts& k, float2* lds, long long f_begin, long long f_end) {
    static_assert(NT == 64 && K::NP == 2 && !K::PINGPONG, "single-wave two-pass plans");
    const int tid = threadIdx.x;
    const int c0 = tid * CPT;
    const raw_t* run_ptr = static_cast<const raw_t*>(p.in) + ((f_begin - (W - 1)) * D + p.base);
    const long long nchunks = (f_end - f_begin + C - 1) / C;
    v2f x[NW][CPT];
    raw_t raw[C][CPT];
    v2f acc[OS][CPT][C];
#pragma unroll
    for (int i = 0; i < W - 1; ++i) {
      raw_t t[CPT];
      load_row<INTERIOR>(p, run_ptr, f_begin - (W - 1) + i, i, c0, t);
#pragma unroll
      for (int cc = 0; cc < CPT; ++cc) x[i][cc] = cvt(t[cc]);
    }
    auto load_chunk_rows = [&](long long ci) {  // chunk ci of this run (clamped: the last chunk is fetched again rather than branching)
      const long long cl = ci < nchunks ? ci : nchunks - 1;
#pragma unroll
      for (int t = 0; t < C; ++t) load_row<INTERIOR>(p, run_ptr, f_begin + cl * C + t, W - 1 + cl * C + t, c0, raw[t]);
    };
    auto take_rows = [&]() {
#pragma unroll
      for (int t = 0; t < C; ++t)
#pragma unroll
        for (int cc = 0; cc < CPT; ++cc) x[W - 1 + t][cc] = cvt(raw[t][cc]);
    };
    auto slide = [&]() {
#pragma unroll
      for (int i = 0; i < W - 1; ++i)
#pragma unroll
        for (int cc = 0; cc < CPT; ++cc) x[i][cc] = x[i + C][cc];
    };
    // chunk 0 by itself
    load_chunk_rows(0);
    take_rows();
    load_chunk_rows(1);
    fir_compute(k, x, acc);
    slide();
    fir_write(k, acc, lds, tid);
    team_sync<true>();
    float2* cur = lds;
    float2* nxt = lds + K::BUF;
    // (the loop is rotated -- the rows requested at the top of an iteration are taken at its END -- so that a load and
    // its wait sit in the same iteration: across the back edge the compiler merges the loop-entry state, which has no
    // stores in flight, into its s_waitcnt count and the wait for the rows would also wait for the chunk's stores)
    take_rows();                   // rows of chunk 1
    for (long long ci = 0; ci + 1 < nchunks; ++ci) {
      load_chunk_rows(ci + 2);     // two chunks ahead
      // one basic block: the next chunk's FIR next to this chunk's first pass
      fir_compute(k, x, acc);
      pass<0>(p, cur, cur, tid, f_begin + ci * C, k.tw);
      slide();
      team_sync<true>();
      pass<1, INTERIOR, MAGSEL>(p, cur, nullptr, tid, f_begin + ci * C, k.tw);
      fir_write(k, acc, nxt, tid);
      team_sync<true>();
      float2* t = cur; cur = nxt; nxt = t;
      take_rows();                 // rows of chunk ci + 2 (waits for them; the chunk's stores stay in flight)
    }
    pass<0>(p, cur, cur, tid, f_begin + (nchunks - 1) * C, k.tw);
    team_sync<true>();
    pass<1, INTERIOR, MAGSEL>(p, cur, nullptr, tid, f_begin + (nchunks - 1) * C, k.tw);
  }

  // The same pipeline over a run of exactly PERIOD = NWP / C chunks with the window as a RING of NWP = NW rounded up to
  // whole chunks: the chunk loop is gone (PERIOD straight-line steps, every window index a compile-time constant), and so
  // are the W-1 register moves per column that slide the window after every chunk (62 of the 499 VALU instructions of
  // the cfg5 chunk loop).  Same taps, same order: bit-identical.  Interior runs only; any other run takes the loop above.
  template <int MAGSEL>
  PFB_DEV void run_overlap_ring(const KernelParams& p, const Consts& k, float2* lds, long long f_begin) {
    static_assert(NT == 64 && K::NP == 2 && !K::PINGPONG, "single-wave two-pass plans");
    const int tid = threadIdx.x;
    const int c0 = tid * CPT;
    const raw_t* run_ptr = static_cast<const raw_t*>(p.in) + ((f_begin - (W - 1)) * D + p.base);
    v2f x[NWP][CPT];
    raw_t raw[C][CPT];
    v2f acc[OS][CPT][C];
#pragma unroll
    for (int i = 0; i < W - 1; ++i) {
      raw_t t[CPT];
      load_row<true>(p, run_ptr, f_begin - (W - 1) + i, i, c0, t);
#pragma unroll
      for (int cc = 0; cc < CPT; ++cc) x[i][cc] = cvt(t[cc]);
    }
    auto load_chunk_rows = [&](int ci) {
#pragma unroll
      for (int t = 0; t < C; ++t) load_row<true>(p, run_ptr, f_begin + ci * C + t, W - 1 + ci * C + t, c0, raw[t]);
    };
    auto take_rows = [&]<int PH>() {  // the rows of chunk PH: logical rows W-1 ... W-2+C of phase PH
#pragma unroll
      for (int t = 0; t < C; ++t)
#pragma unroll
        for (int cc = 0; cc < CPT; ++cc) x[(W - 1 + t + C * PH) % NWP][cc] = cvt(raw[t][cc]);
    };
    load_chunk_rows(0);
    take_rows.template operator()<0>();
    load_chunk_rows(1);
    fir_compute<0, NWP>(k, x, acc);
    fir_write(k, acc, lds, tid);
    team_sync<true>();
    float2* cur = lds;
    float2* nxt = lds + K::BUF;
    take_rows.template operator()<1>();
    auto step = [&]<int CI>() {  // chunk CI's FFT next to chunk CI + 1's FIR
      if constexpr (CI + 2 < PERIOD) load_chunk_rows(CI + 2);
      fir_compute<CI + 1, NWP>(k, x, acc);
      pass<0>(p, cur, cur, tid, f_begin + CI * C, k.tw);
      team_sync<true>();
      pass<1, true, MAGSEL>(p, cur, nullptr, tid, f_begin + CI * C, k.tw);
      fir_write(k, acc, nxt, tid);
      team_sync<true>();
      float2* t = cur; cur = nxt; nxt = t;
      if constexpr (CI + 2 < PERIOD) take_rows.template operator()<CI + 2>();
    };
    [&]<int... CI>(std::integer_sequence<int, CI...>) { (step.template operator()<CI>(), ...); }(std::make_integer_sequence<int, PERIOD - 1>{});
    pass<0>(p, cur, cur, tid, f_begin + (PERIOD - 1) * C, k.tw);
    team_sync<true>();
    pass<1, true, MAGSEL>(p, cur, nullptr, tid, f_begin + (PERIOD - 1) * C, k.tw);
  }

  template <int MAGSEL = -1>
  PFB_DEV void run_overlap(const KernelParams& p, float2* lds) {
    long long run = blockIdx.x;
    run = xcd_remap_block(run, gridDim.x, p.xcd_remap);
    const long long f_begin = run * p.frames_per_block;
    if (f_begin >= p.frames) return;
    const long long f_last = f_begin + p.frames_per_block;
    const long long f_end = f_last < p.frames ? f_last : p.frames;
    Consts k;
    setup(p, threadIdx.x, k);
    const bool interior = p.vec_ok && ((f_begin - (W - 1)) * D + p.base >= 0) && (f_last <= p.frames);
    if constexpr (kRingOk) {
      if (interior && p.frames_per_block == C * PERIOD) {
        run_overlap_ring<MAGSEL>(p, k, lds, f_begin);
        return;
      }
    }
    if (interior) run_overlap_impl<true, MAGSEL>(p, k, lds, f_begin, f_end);
    else run_overlap_impl<false, MAGSEL>(p, k, lds, f_begin, f_end);
  }

  // ---- schedule T: FIR team + FFT team (large M) --------------------------------------------------------
  // At M = 1024 one frame needs all 1024 columns, so the FIR is a team effort (NT threads x CPT columns), and
  // in the plain sliding kernel the same 16 waves then all turn to the FFT: every phase leaves either the VALU
  // or the LDS idle, and the passes cost several workgroup barriers per chunk (44 % VALU-busy, waves waiting
  // 63 % of their cycles: profiles/).  Here the FIR team only filters -- a sliding register window per thread,
  // chunk after chunk into one of three LDS buffers -- and C more waves transform: FFT wave w takes frame w of
  // the previous chunk and runs the first two passes of its M-point FFT alone (M / 64 points per lane), so
  // those passes need no barrier at all, only the wave's own program order.  One workgroup barrier per chunk
  // rotates the buffers.
  // One non-final pass of ONE frame by one wave, in place: every read of the pass (all iterations) happens
  // before its first write, and the wave's own program order is the only synchronisation.
  template <int I>
  PFB_DEV void pass_frame(const KernelParams& p, float2* fbuf, int lane, const v2f (&tw)[2][16]) {
    constexpr int R = K::R(I), S = K::S(I), KK = K::K(I), RS = K::RS(I);
    constexpr int IPF = M / R, ITERS = (IPF + 63) / 64;
    constexpr int S1 = K::S(I + 1), RS1 = K::RS(I + 1);
    static_assert(I < K::NP - 1, "the last pass (with the stores) belongs to the FIR team");
    constexpr bool TW_REGS = (ITERS == 1) && !K::TW_TABLE;
    v2f x[ITERS][R];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int item = lane + it * 64;
      const bool active = (IPF % 64 == 0) || (item < IPF);
      const v2f* s2 = reinterpret_cast<const v2f*>(fbuf) + (active ? item : 0);
#pragma unroll
      for (int n = 0; n < R; ++n) x[it][n] = s2[n * RS];
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int item = lane + it * 64;
      const bool active = (IPF % 64 == 0) || (item < IPF);
      const int kk = item / S, rest = item % S;
      Dft<R>::run(x[it]);
      if constexpr (TW_REGS) {
#pragma unroll
        for (int k = 1; k < R; ++k) x[it][k] = cmul_w(x[it][k], tw[I][k]);
      } else {
        const float4* t4 = reinterpret_cast<const float4*>(p.tw_lane + K::TW_OFF(I) + (active ? rest : 0) * K::TWR(I));
#pragma unroll
        for (int k2 = 0; k2 < K::TWR(I) / 2; ++k2) {
          const float4 t = t4[k2];
          if (k2 > 0) x[it][2 * k2] = cmul_w(x[it][2 * k2], (v2f){t.x, t.y});
          if (2 * k2 + 1 < R) x[it][2 * k2 + 1] = cmul_w(x[it][2 * k2 + 1], (v2f){t.z, t.w});
        }
      }
      if (active) {
        const int n1 = rest / S1, rest2 = rest % S1;
        v2f* d2 = reinterpret_cast<v2f*>(fbuf) + n1 * RS1 + kk * S1 + rest2;
#pragma unroll
        for (int k = 0; k < R; ++k) d2[k * KK * S1] = x[it][k];
      }
    }
  }

  // FIR team: chunk ci into buffer ci % 3, then the LAST pass (and the stores) of chunk ci - 2, whose first two
  // passes the FFT team finished in the step before.  The stores are most of the FFT's memory work and the FIR
  // team has issue slots to spare, while four FFT waves doing everything were the bottleneck (2.4 of 2.9 ms).
  // (A variant of this kernel for channel-major handles -- the last pass writing frame-major scratch tiles that the FFT
  // team moved into place transposed -- measured slower than frame-major slabs plus a transpose kernel, 7.4 against 6.4 ms
  // per 2^30 samples at M = 1024, and was removed: DESIGN.md section 5.4.)
  // (The window as a ring of registers -- blocks of 10 steps at M = 1024, no slide: 50 fewer VALU instructions per two steps --
  // measured 0.611-0.612 against 0.608-0.609, nothing on M = 560 / 400 / 320: the FIR team's issue slots are not what the
  // kernel waits for; not kept.  The pipelined single-wave kernel keeps its ring, run_overlap_ring.)
  template <bool INTERIOR, int MAGSEL = -1>
  PFB_DEV void fir_team(const KernelParams& p, const Consts& k, float2* bufs, long long f_begin, int nch) {
    const int tid = threadIdx.x;
    const int c0 = tid * CPT;
    auto last_pass = [&](float2* buf, int c) {   // chunk c of this run
      // the thread index is laundered so that everything the pass derives from it (LDS and store addresses) is
      // recomputed here -- a few VALU instructions -- instead of being hoisted out of the chunk loop: hoisted, ONE of
      // them was spilled, and its reload (a vector-memory load, which returns in order) made every step wait for the
      // row prefetch issued just before it: s_waitcnt vmcnt(0) four times per iteration of the steady-state loop
      int t2 = tid;
      asm volatile("" : "+v"(t2));
      pass<K::NP - 1, INTERIOR, MAGSEL>(p, buf, nullptr, t2, f_begin + (long long)c * C, k.tw);
    };
    const raw_t* run_ptr = static_cast<const raw_t*>(p.in) + ((f_begin - (W - 1)) * D + p.base);
    v2f x[NW][CPT];
    raw_t raw[2][C][CPT];  // two chunks of rows in flight: one chunk is only ~1.5 us of work, less than a loaded HBM round trip
#pragma unroll
    for (int i = 0; i < W - 1; ++i) {
      raw_t t[CPT];
      load_row<INTERIOR>(p, run_ptr, f_begin - (W - 1) + i, i, c0, t);
#pragma unroll
      for (int cc = 0; cc < CPT; ++cc) x[i][cc] = cvt(t[cc]);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int t = 0; t < C; ++t) load_row<INTERIOR>(p, run_ptr, f_begin + u * C + t, W - 1 + u * C + t, c0, raw[u][t]);
    int b_fir = 0, b_last = 1;  // buffer of chunk ci, buffer of chunk ci - 2 (= (ci + 1) % 3)
    // One chunk step.  U: which of the two row sets holds chunk ci; LASTP: chunk ci - 2 exists (every step but a run's
    // first two, which are peeled off so that the steady-state loop issues the same stores on every path: see pass<FULL>).
    // (An unconditional, clamped prefetch would make the loads path-independent too, but its 64-bit row addresses cost
    // this team the registers it does not have: 8-12 spilled, reloaded inside the loop.)
    auto step = [&]<int U, bool LASTP>(int ci) {
      const long long f0 = f_begin + (long long)ci * C;
#pragma unroll
      for (int t = 0; t < C; ++t)
#pragma unroll
        for (int cc = 0; cc < CPT; ++cc) x[W - 1 + t][cc] = cvt(raw[U][t][cc]);
      if (ci + 2 < nch) {
        const long long rel = (long long)(ci + 2) * C + (W - 1);
#pragma unroll
        for (int t = 0; t < C; ++t) load_row<INTERIOR>(p, run_ptr, f0 + 2 * C + t, rel + t, c0, raw[U][t]);
      }
      fir_to_lds(k, x, bufs + b_fir * K::BUF, tid);
#pragma unroll
      for (int i = 0; i < W - 1; ++i)
#pragma unroll
        for (int cc = 0; cc < CPT; ++cc) x[i][cc] = x[i + C][cc];
      if constexpr (LASTP) last_pass(bufs + b_last * K::BUF, ci - 2);
      __syncthreads();  // (barrier ci) chunk ci handed to the FFT team, buffer of chunk ci - 2 free again
      b_fir = (b_fir == 2) ? 0 : b_fir + 1;
      b_last = (b_last == 2) ? 0 : b_last + 1;
    };
    step.template operator()<0, false>(0);
    step.template operator()<1, false>(1);
    for (int ci2 = 2; ci2 < nch; ci2 += 2) {
      step.template operator()<0, true>(ci2);
      step.template operator()<1, true>(ci2 + 1);
    }
    // drain: the FFT team finishes chunk nch - 1 while chunk nch - 2 gets its last pass, then chunk nch - 1
    if (nch >= 2) last_pass(bufs + b_last * K::BUF, nch - 2);
    __syncthreads();  // (barrier nch)
    b_last = (b_last == 2) ? 0 : b_last + 1;
    last_pass(bufs + b_last * K::BUF, nch - 1);
  }

  template <int MAGSEL = -1>
  PFB_DEV void run_teams(const KernelParams& p, float2* bufs) {
    static_assert(K::NP == 3 && !K::PINGPONG && NT % 64 == 0, "three in-place passes");
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nch = p.frames_per_block / C;  // even (host rounds); the last workgroup filters zero padding past the end
    Consts k;
    setup(p, wave < NT / 64 ? threadIdx.x : lane, k);  // FFT team: only the twiddles are used, rows `lane % S` of the passes' tables
    long long run = blockIdx.x;  // one run per workgroup, in dispatch order
    run = xcd_remap_block(run, gridDim.x, p.xcd_remap);
    const long long f_begin = run * p.frames_per_block;
    if (f_begin >= p.frames) return;
    if (wave < NT / 64) {
      const bool interior = p.vec_ok && ((f_begin - (W - 1)) * D + p.base >= 0) && (f_begin + p.frames_per_block <= p.frames);
      if (interior) fir_team<true, MAGSEL>(p, k, bufs, f_begin, nch);
      else fir_team<false, MAGSEL>(p, k, bufs, f_begin, nch);
    } else {
      // (s_setprio for either team, measured: cfg4 -3 % / 0, M=560 +1.6 % / +1 %: noise)
      const int fr = wave - NT / 64;  // my frame inside every chunk
      int b = 0;                      // buffer of chunk s - 1
#pragma unroll 1
      for (int s = 0; s <= nch; ++s) {
        if (s >= 1) {
          float2* fbuf = bufs + b * K::BUF + fr * K::FS;
          pass_frame<0>(p, fbuf, lane, k.tw);
          team_sync<true>();
          pass_frame<1>(p, fbuf, lane, k.tw);
          b = (b == 2) ? 0 : b + 1;
        }
        __syncthreads();  // (barrier s) behind it the FIR team's last pass of chunk s - 2 is done
      }
    }
  }

  // ---- schedule W: several independent workgroups per CU (large M) --------------------------------------------
  // The team kernel (schedule T) holds its sliding window as converted float pairs: 168 registers per FIR thread,
  // three 34 KB chunk buffers, so ONE 12-wave workgroup per CU whose teams meet at a barrier every 4 frames; nothing
  // hides a run's start-up (W-1 halo rows, two steps of fill, two of drain), so runs must be long (512 frames = 2 MB
  // of input per CU), and 256 CUs each streaming their own megabytes is the access shape HBM serves worst (DESIGN.md
  // section 6).  Here a workgroup is NT/64 waves, every wave does both jobs, and the state between chunks is small
  // enough for TWO (M = 1024) or more workgroups per CU, which are not synchronised with each other: while one
  // filters (VALU) the other transforms and stores (LDS, memory), and one's start-up is covered by the other's
  // steady state, so runs can be short.
  //  * the window is kept as the RAW samples (one register per int16 / int8 pair instead of two) and converted when
  //    used: W-1+C conversions per column and chunk instead of C, i.e. +12 % VALU work in the FIR at M = 1024 for
  //    30 instead of 60 persistent registers;
  //  * a chunk is C = NT/64 frames in ONE buffer: FIR by everybody -> barrier -> wave w runs ALL passes of frame w
  //    by itself (M/64 points per lane: wave-local, no barrier inside the FFT) and stores it -> barrier;
  //  * a thread's two adjacent columns are two adjacent branch outputs: one ds_write_b128 (the team kernel's
  //    ds_write_b64 pairs were a 2-way bank conflict, 15 % of its LDS cycles).
  // Same taps, same accumulation order (taps ascending), same passes: bit-identical to the other plans of the shape.
  template <bool MAG>
  PFB_DEV void last_pass_frame(const KernelParams& p, const float2* fbuf, int lane, long long f) {
    constexpr int I = K::NP - 1, R = K::R(I), KK = K::K(I), RS = K::RS(I);
    constexpr int IPF = M / R, ITERS = (IPF + 63) / 64;
    static_assert(K::S(I) == 1 && !CM && OS == 1, "frame-major, critically sampled");
    v2f x[ITERS][R];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int item = lane + it * 64;
      const v2f* s2 = reinterpret_cast<const v2f*>(fbuf) + ((IPF % 64 == 0) || item < IPF ? item : 0);
#pragma unroll
      for (int n = 0; n < R; ++n) x[it][n] = s2[n * RS];
    }
    const int shift = (p.flags & PFB_FLAG_FFTSHIFT) ? (M / 2) : 0;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int kk = lane + it * 64;
      Dft<R>::run(x[it]);
      if ((IPF % 64 == 0) || kk < IPF) {
        auto col_of = [&](int ch) {
          const int c2 = ch + shift;
          return c2 >= M ? c2 - M : c2;
        };
        // (a wave-uniform row pointer + an UNSIGNED 32-bit lane offset: the address form that needs no 64-bit vector
        // arithmetic and no register pair per pointer)
        auto slot = [&](auto* rowp, int k) {
          if constexpr (K::POW2) {  // fftshift swaps the row's halves: two base pointers, compile-time offsets (see pass<>)
            auto* lo = rowp + (unsigned)(kk + shift);
            auto* hi = rowp + (unsigned)(kk + (M / 2 - shift));
            return (k < R / 2) ? lo + k * KK : hi + (k - R / 2) * KK;
          } else {
            return rowp + (unsigned)col_of(kk + k * KK);
          }
        };
        if constexpr (MAG) {
          float* rowm = reinterpret_cast<float*>(p.out) + f * M;
#pragma unroll
          for (int k = 0; k < R; ++k) *slot(rowm, k) = mag_out(x[it][k].x, x[it][k].y, p.flags);
        } else {
          float2* row = p.out + f * M;
#pragma unroll
          for (int k = 0; k < R; ++k) *reinterpret_cast<v2f*>(slot(row, k)) = x[it][k];
        }
      }
    }
  }

  // taps of column pair `pr` of this thread (columns c0 + 2 pr, c0 + 2 pr + 1), two taps per register pair: the table of setup()
  PFB_DEV void load_taps_pair(const KernelParams& p, int tid, int pr, v2f (&hp)[(W + 1) / 2][2]) {
    const int c0 = tid * CPT + 2 * pr;
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      const int col = (K::LANES < NT && c0 >= D) ? 0 : c0 + cc;  // idle lanes read column 0's taps
      const float4* tl = reinterpret_cast<const float4*>(p.taps_lane + (size_t)col * K::WP);
#pragma unroll
      for (int q4 = 0; q4 < K::WP / 4; ++q4) {
        const float4 v = tl[q4];
        if (2 * q4 < (W + 1) / 2) hp[2 * q4][cc] = (v2f){v.x, v.y};
        if (2 * q4 + 1 < (W + 1) / 2) hp[2 * q4 + 1][cc] = (v2f){v.z, v.w};
      }
    }
  }

  // Twiddle rows of the non-final passes as this kernel keeps them in LDS: in registers they are 2 R per pass, and from
  // the global table their loads would queue behind the row prefetch (a wave's vector-memory operations return in order:
  // a pass that waits for a twiddle load waits for every HBM load issued before it).  Row stride: even (16-byte reads)
  // with an odd half, so that the sixteen lanes of a ds_read_b128 group read sixteen different bank quads.
  static constexpr int TWS(int i) { return (K::TWR(i) / 2) % 2 ? K::TWR(i) : K::TWR(i) + 2; }
  static constexpr int TWL_OFF(int i) { int o = 0; for (int j = 0; j < i; ++j) o += K::S(j) * TWS(j); return o; }
  static constexpr int TWL_ELEMS = TWL_OFF(K::NP - 1);

  PFB_DEV void fill_twiddles(const KernelParams& p, float2* twl) {
#pragma unroll
    for (int i = 0; i < K::NP - 1; ++i) {
      const int R = K::R(i), n = K::S(i) * R;
      for (int idx = threadIdx.x; idx < n; idx += NT)
        twl[TWL_OFF(i) + (idx / R) * TWS(i) + idx % R] = p.tw_lane[K::TW_OFF(i) + (idx / R) * K::TWR(i) + idx % R];
    }
  }

  // One non-final pass of one frame by one wave, in place, like pass_frame, twiddles from the LDS table.
  template <int I>
  PFB_DEV void pass_frame_lean(float2* fbuf, const float2* twl, int lane) {
    constexpr int R = K::R(I), S = K::S(I), KK = K::K(I), RS = K::RS(I);
    constexpr int IPF = M / R;
    constexpr int S1 = K::S(I + 1), RS1 = K::RS(I + 1);
    static_assert(I < K::NP - 1 && IPF <= 64 && R % 2 == 0, "one item per lane");
    const bool active = (IPF == 64) || (lane < IPF);
    const int item = active ? lane : 0;
    const int kk = item / S, rest = item % S;
    v2f x[R];
    const v2f* s2 = reinterpret_cast<const v2f*>(fbuf) + item;
#pragma unroll
    for (int n = 0; n < R; ++n) x[n] = s2[n * RS];
    const float4* t4 = reinterpret_cast<const float4*>(twl + TWL_OFF(I) + rest * TWS(I));
    const int n1 = rest / S1, rest2 = rest % S1;
    v2f* d2 = reinterpret_cast<v2f*>(fbuf) + n1 * RS1 + kk * S1 + rest2;
    Dft<R>::run(x);
#pragma unroll
    for (int k2 = 0; k2 < R / 2; ++k2) {
      const float4 t = t4[k2];
      if (k2 > 0) x[2 * k2] = cmul_w(x[2 * k2], (v2f){t.x, t.y});
      x[2 * k2 + 1] = cmul_w(x[2 * k2 + 1], (v2f){t.z, t.w});
      if (active) {
        d2[(2 * k2) * KK * S1] = x[2 * k2];
        d2[(2 * k2 + 1) * KK * S1] = x[2 * k2 + 1];
      }
    }
  }

  // CPT raw samples of row `rel` of an interior run: uniform 64-bit base in SGPRs + a 32-bit lane offset (the address
  // form that costs one register per lane; as pointer arithmetic the compiler kept a 64-bit address pair per row)
  PFB_DEV void load_row_sbase(const raw_t* run_ptr, long long rel, int c0, raw_t (&raw)[CPT]) {
    static_assert(sizeof(RawVec) % 4 == 0, "whole dwords per lane");
    typedef unsigned dwords_t __attribute__((ext_vector_type(sizeof(RawVec) / 4)));
    const int cs = (K::LANES < NT && c0 >= D) ? 0 : c0;  // lanes beyond the last column read column 0 (and never use it)
    const raw_t* rowp = run_ptr + rel * D;               // wave-uniform
    const dwords_t v = *reinterpret_cast<const dwords_t*>(rowp + (unsigned)cs);
    __builtin_memcpy(&raw[0], &v, sizeof(RawVec));
  }

  // The fast path: whole chunks of a run whose every row (halo included) lies inside `in`, aligned vectors.
  // The chunk loop is laid out so that the compiler's s_waitcnt counts are EXACT: a wave's vector-memory operations
  // return in order, the compiler counts them per path, and wherever the count differs between paths into a point it
  // assumes the fewest younger operations, i.e. waits for more than the load it needs -- typically for every store
  // issued since.  So (1) the loop is rotated: an iteration is [transform + store chunk i, its first step requesting
  // the rows of chunk i + 1] then [FIR of chunk i + 1], which keeps a load and its wait in the SAME iteration;
  // (2) nothing in the loop is conditionally issued: the prefetch past the run's end re-reads the last row, partial
  // chunks are left to the careful path, PFB_FLAG_MAGNITUDE is a template parameter of the kernel.
  static constexpr int NWV = NT / 64;        // waves per workgroup
  static constexpr int FPW = C / (NT / 64);  // frames each wave transforms per chunk
  static constexpr bool kTapsResident = K::MIN_WAVES <= 2;  // 256 registers: the taps stay; otherwise they are re-read per chunk
  static constexpr bool kLean = K::MIN_WAVES >= 4;          // 128 registers
  // r_first, G, r_hi: this workgroup's runs r_first, r_first + G, ... below r_hi (all of them whole, with their halo
  // inside `in`), chained without a bubble: the step that would request the next chunk's rows requests the next run's
  // W-1 halo rows as well (into the window registers, which are dead at that point).  G = the grid: with one workgroup
  // per run there is no second run; with a grid of resident workgroups (PFB_OPT_GRID) the taps, the twiddle table and
  // the start-up latency are paid once per workgroup, and SHORT runs become affordable -- at any moment the chip then
  // works on G consecutive short runs, a dense window sweeping through the stream (what HBM
  // delivers depends on how compact the set of concurrently touched DRAM rows is).
  template <bool MAG>
  PFB_DEV void twin_fast(const KernelParams& p, float2* lds, const float2* twl, long long r_first, long long G, long long r_hi) {
    static_assert(C % NWV == 0 && OS == 1 && K::NP == 3 && !K::PINGPONG, "whole frames per wave and chunk");
    static_assert(CPT % 2 == 0 && K::S(0) % 2 == 0 && K::RS(0) % 2 == 0 && K::FS % 2 == 0 && D % 2 == 0, "adjacent, aligned branch pairs");
    constexpr int NPR = CPT / 2;  // column pairs per thread
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // in an SGPR: frame indices and row pointers stay scalar
    const int c0 = (K::LANES < NT && tid >= K::LANES) ? 0 : tid * CPT;  // idle lanes (D not a multiple of 64 CPT) shadow thread 0
    const bool lane_on = !(K::LANES < NT) || tid < K::LANES;
    // Registers decide this kernel (workgroups per CU = 512 / registers / waves per workgroup).  Where they do not fit,
    // the taps (W per column) are NOT kept across the FFT: a column pair's come back from the L2-resident table every
    // chunk -- the first pair's loads are issued before the last pass, so that they land under its stores and the
    // barrier, the next pair's under the FIR of the pair before.
    constexpr int NHP = kTapsResident ? NPR : (NPR > 1 ? 2 : 1);
    v2f hp[NHP][(W + 1) / 2][2];
    const v2f conj_mul = (v2f){1.f, (p.flags & PFB_FLAG_CONJUGATE_INPUT) ? -1.f : 1.f};
    int upos[NPR];  // LDS position of the lower of a pair's two adjacent branch outputs
#pragma unroll
    for (int pr = 0; pr < NPR; ++pr) {
      const int n = D - 1 - (c0 + 2 * pr + 1);
      upos[pr] = (n / K::S(0)) * K::RS(0) + (n % K::S(0));
    }
    if constexpr (kTapsResident) {
#pragma unroll
      for (int pr = 0; pr < NPR; ++pr) load_taps_pair(p, tid, pr, hp[pr]);
    } else {
      load_taps_pair(p, tid, 0, hp[0]);
    }
    const int nch = p.frames_per_block / C;  // chunks per run
    long long r = r_first;
    const raw_t* run_ptr = static_cast<const raw_t*>(p.in) + ((r * p.frames_per_block - (W - 1)) * D + p.base);
    raw_t win[W - 1][CPT];  // rows f0-(W-1) ... f0-1, as loaded
    raw_t raw[C][CPT];      // rows f0 ... f0+C-1
#pragma unroll
    for (int i = 0; i < W - 1; ++i) load_row_sbase(run_ptr, i, c0, win[i]);
#pragma unroll
    for (int t = 0; t < C; ++t) load_row_sbase(run_ptr, W - 1 + t, c0, raw[t]);
    // FIR of a chunk's C frames for my CPT columns, a column pair at a time, then the window slides.  Row i of the NW
    // window rows feeds frame t with tap j = W-1+t-i; rows are walked newest first so that every accumulator takes its
    // taps in ascending order (the order of fir_to_lds: bit-identical sums).
    auto fir_chunk = [&]() {
#pragma unroll
      for (int pr = 0; pr < NPR; ++pr) {
        const int hb = kTapsResident ? pr : (pr & 1);
        if constexpr (!kTapsResident) { if (pr + 1 < NPR) load_taps_pair(p, tid, pr + 1, hp[(pr + 1) & 1]); }
        v2f acc[2][C];
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
#pragma unroll
          for (int t = 0; t < C; ++t) acc[cc][t] = (v2f){0.f, 0.f};
          int tok = 0;
#pragma unroll
          for (int i = NW - 1; i >= 0; --i) {
            const v2f xi = cvt(i >= W - 1 ? raw[i - (W - 1)][2 * pr + cc] : win[i][2 * pr + cc]);
#pragma unroll
            for (int t = 0; t < C; ++t) {
              const int j = W - 1 + t - i;
              if (j >= 0 && j < W) {
                if (j & 1) fma_tap_hi(acc[cc][t], xi, hp[hb][j >> 1][cc], tok);
                else fma_tap_lo(acc[cc][t], xi, hp[hb][j >> 1][cc], tok);
              }
            }
          }
          if constexpr (kLean) {
            // (128 registers: a column's C sums leave before the next column starts -- 8-byte writes, a 2-way bank
            // conflict on C writes per column, instead of holding both columns' sums for the 16-byte write)
            if (lane_on) {
              v2f* d2 = reinterpret_cast<v2f*>(lds) + upos[pr] + (1 - cc);
#pragma unroll
              for (int t = 0; t < C; ++t) d2[t * K::FS] = acc[cc][t] * conj_mul;
            }
            asm volatile("" ::: "memory");
          }
        }
        if constexpr (!kLean) {
          if (lane_on) {
            // column c + 1 is branch n - 1 (even), column c branch n: adjacent positions, 16-byte aligned -> one ds_write_b128
            float4* d4 = reinterpret_cast<float4*>(lds + upos[pr]);
#pragma unroll
            for (int t = 0; t < C; ++t) {
              const v2f a = acc[1][t] * conj_mul, b = acc[0][t] * conj_mul;
              d4[t * (K::FS / 2)] = make_float4(a.x, a.y, b.x, b.y);
            }
          }
        }
      }
#pragma unroll
      for (int i = 0; i < W - 1; ++i)
#pragma unroll
        for (int cc = 0; cc < CPT; ++cc) win[i][cc] = (i + C >= W - 1) ? raw[i + C - (W - 1)][cc] : win[i + C][cc];
    };
    fir_chunk();
    int ci = 0;
    for (;;) {
      const long long f0 = r * p.frames_per_block + (long long)ci * C;
      bool more = true;
      __syncthreads();  // the chunk is in LDS
      // my frames of the chunk: wave + NWV fi.  What the next chunk needs is requested between the passes (no
      // vector-memory load inside them: a pass waiting for a table entry would wait for every row requested before it):
      // its rows behind the first pass 0 -- they land under the rest of the transform and its stores --, its first
      // taps behind the last pass 1
#pragma unroll
      for (int fi = 0; fi < FPW; ++fi) {
        const int fc = wave + NWV * fi;
        float2* fbuf = lds + fc * K::FS;
        pass_frame_lean<0>(fbuf, twl, lane);
        if (fi == 0) {
          asm volatile("" ::: "memory");
          if (ci + 1 < nch) {  // the run's next chunk
            ++ci;
            const long long rel = (long long)ci * C + (W - 1);
#pragma unroll
            for (int t = 0; t < C; ++t) load_row_sbase(run_ptr, rel + t, c0, raw[t]);
          } else if (r + G < r_hi) {  // my next run: its halo (the window registers are dead here) and its first chunk
            r += G;
            ci = 0;
            run_ptr = static_cast<const raw_t*>(p.in) + ((r * p.frames_per_block - (W - 1)) * D + p.base);
#pragma unroll
            for (int i = 0; i < W - 1; ++i) load_row_sbase(run_ptr, i, c0, win[i]);
#pragma unroll
            for (int t = 0; t < C; ++t) load_row_sbase(run_ptr, W - 1 + t, c0, raw[t]);
          } else {
            more = false;
          }
        }
        team_sync<true>();
        pass_frame_lean<1>(fbuf, twl, lane);
        team_sync<true>();
        if (fi == FPW - 1 && !kTapsResident) {
          asm volatile("" ::: "memory");
          load_taps_pair(p, tid, 0, hp[0]);
        }
        last_pass_frame<MAG>(p, fbuf, lane, f0 + fc);
      }
      __syncthreads();  // everybody has read the chunk out of LDS
      if (!more) break;
      fir_chunk();
    }
  }

  // The careful path, for the few runs that touch the history in front of the call's first sample, a partial last chunk
  // or a buffer the vector loads cannot take: no window kept in registers -- per chunk and column the NW rows are
  // fetched again, sample by sample with the checks of load_row<false> (all of a column's loads in flight together: a
  // run of this kind is a straggler among thousands, but a serial one would outlast the whole kernel), columns in a
  // rolled loop.  Same taps in the same order: the same bits.
  template <bool MAG>
  PFB_DEV void twin_careful(const KernelParams& p, float2* lds, const float2* twl, long long f_begin, long long f_end) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const raw_t* in = static_cast<const raw_t*>(p.in);
    const raw_t* hist = static_cast<const raw_t*>(p.hist);
    const v2f conj_mul = (v2f){1.f, (p.flags & PFB_FLAG_CONJUGATE_INPUT) ? -1.f : 1.f};
    for (long long f0 = f_begin; f0 < f_end; f0 += C) {
      if (!(K::LANES < NT) || tid < K::LANES) {
#pragma unroll 1
        for (int cc = 0; cc < CPT; ++cc) {
          const int c = tid * CPT + cc, n = D - 1 - c;
          const int pos = (n / K::S(0)) * K::RS(0) + (n % K::S(0));
          const float* tl = p.taps_lane + (size_t)c * K::WP;
          raw_t r[NW];
#pragma unroll
          for (int i = 0; i < NW; ++i) {
            const long long fr = f0 - (W - 1) + i;  // frame whose newest row this is
            const long long s = fr * D + p.base + c;
            r[i] = (fr >= p.frames) ? raw_t{} : ((s >= 0) ? in[s] : hist[p.hist_samples + s]);
          }
          float h[W];
#pragma unroll
          for (int j = 0; j < W; ++j) h[j] = tl[j];
          v2f acc[C];
#pragma unroll
          for (int t = 0; t < C; ++t) acc[t] = (v2f){0.f, 0.f};
#pragma unroll
          for (int i = NW - 1; i >= 0; --i) {
            const v2f xi = cvt(r[i]);
#pragma unroll
            for (int t = 0; t < C; ++t) {
              const int j = W - 1 + t - i;
              if (j >= 0 && j < W) acc[t] = fma2(xi, splat(h[j]), acc[t]);
            }
          }
#pragma unroll
          for (int t = 0; t < C; ++t) reinterpret_cast<v2f*>(lds)[t * K::FS + pos] = acc[t] * conj_mul;
        }
      }
      __syncthreads();
#pragma unroll 1
      for (int fc = wave; fc < C; fc += NWV) {
        const long long f = f0 + fc;
        float2* fbuf = lds + fc * K::FS;
        if (f < f_end) {
          pass_frame_lean<0>(fbuf, twl, lane);
          team_sync<true>();
          pass_frame_lean<1>(fbuf, twl, lane);
          team_sync<true>();
          last_pass_frame<MAG>(p, fbuf, lane, f);
        }
      }
      __syncthreads();
    }
  }

  template <bool MAG>
  PFB_DEV void run_twin(const KernelParams& p, float2* lds, float2* twl) {
    const long long G = gridDim.x, fpb = p.frames_per_block;
    long long r = blockIdx.x;
    r = xcd_remap_block(r, G, p.xcd_remap);
    const long long nruns = (p.frames + fpb - 1) / fpb;
    if (r >= nruns) return;
    // runs [r_lo, r_hi) are whole and have their halo inside `in`: the fast path; the others (the call's first run, a
    // partial last one, everything if the buffer is not aligned for the vector loads) take the careful one
    const long long need = (long long)(W - 1) * D - p.base;  // samples of halo in front of frame 0
    const long long r_lo = need > 0 ? (need + fpb * D - 1) / (fpb * D) : 0;
    const long long r_hi = p.vec_ok ? p.frames / fpb : 0;
    fill_twiddles(p, twl);  // (visible behind the chunk loops' first barrier)
    auto careful = [&](long long rr) {
      const long long fb = rr * fpb, fl = fb + fpb;
      twin_careful<MAG>(p, lds, twl, fb, fl < p.frames ? fl : p.frames);
    };
    for (; r < nruns && r < r_lo; r += G) careful(r);
    if (r < r_hi) {
      twin_fast<MAG>(p, lds, twl, r, G, r_hi);
      r += ((r_hi - 1 - r) / G + 1) * G;
    }
    for (; r < nruns; r += G) careful(r);
  }

  // ---- schedule D: sliding windows with the halo shared inside the workgroup -------------------------
  // A workgroup of NWV waves covers NWV*L consecutive frames, wave w the L frames [w*L, (w+1)*L) with its
  // own register window.  Short runs keep the whole chip inside one dense, in-order sweeping window
  // (DRAM rows are finished while open: tools/membench2), but a short run's W-1 halo rows would be
  // fetched from HBM twice -- by this wave now and by its predecessor, as the tail of ITS run, a few
  // microseconds later.  So each wave PUBLISHES the raw halo rows it loads in an LDS slot, and its
  // predecessor takes the last W-1 rows of its run from that slot instead of from memory: every row
  // is fetched once, except the W-1 rows at workgroup boundaries ((W-1)/(NWV*L) extra reads).
  template <bool INTERIOR, int NWV, int L>
  PFB_DEV void shared_impl(const KernelParams& p, const Consts& k, float2* lds, raw_t* halo_mine,
                           const raw_t* halo_next, int wave, long long f_begin) {
    static_assert(L % C == 0 && L >= W - 1, "runs are whole chunks and at least one halo long");
    constexpr int TAIL0 = L - (W - 1);  // first row of the run that the successor publishes
    const int tid = threadIdx.x & 63;
    const int c0 = tid * CPT;
    const raw_t* run_ptr = static_cast<const raw_t*>(p.in) + ((f_begin - (W - 1)) * D + p.base);
    v2f x[NW][CPT];
    raw_t raw[C][CPT];
#pragma unroll
    for (int i = 0; i < W - 1; ++i) {
      raw_t t[CPT];
      load_row<INTERIOR>(p, run_ptr, f_begin - (W - 1) + i, i, c0, t);
#pragma unroll
      for (int cc = 0; cc < CPT; ++cc) {
        x[i][cc] = cvt(t[cc]);
        if constexpr (INTERIOR) {
          if (!(K::LANES < NT) || tid < K::LANES) halo_mine[i * D + c0 + cc] = t[cc];
        }
      }
    }
#pragma unroll
    for (int t = 0; t < C; ++t) load_row<INTERIOR>(p, run_ptr, f_begin + t, W - 1 + t, c0, raw[t]);
    __syncthreads();  // every wave's halo slot is published
    const bool tail_from_lds = INTERIOR && (wave < NWV - 1);
#pragma unroll
    for (int ci = 0; ci < L / C; ++ci) {
#pragma unroll
      for (int t = 0; t < C; ++t) {
        const int r = ci * C + t;
        if (r >= TAIL0 && tail_from_lds) {
#pragma unroll
          for (int cc = 0; cc < CPT; ++cc)
            x[W - 1 + t][cc] = cvt(halo_next[(r - TAIL0) * D + ((K::LANES < NT && c0 >= D) ? 0 : c0 + cc)]);
        } else {
#pragma unroll
          for (int cc = 0; cc < CPT; ++cc) x[W - 1 + t][cc] = cvt(raw[t][cc]);
        }
      }
      if (ci + 1 < L / C) {  // prefetch the next chunk's rows (those not coming from LDS)
#pragma unroll
        for (int t = 0; t < C; ++t) {
          const int r = (ci + 1) * C + t;
          if (!(r >= TAIL0 && tail_from_lds)) load_row<INTERIOR>(p, run_ptr, f_begin + r, W - 1 + r, c0, raw[t]);
        }
      }
      fir_fft_store<true>(p, k, x, lds, tid, f_begin + ci * C);
#pragma unroll
      for (int i = 0; i < W - 1; ++i)
#pragma unroll
        for (int cc = 0; cc < CPT; ++cc) x[i][cc] = x[i + C][cc];
    }
  }

  template <int NWV, int L>
  PFB_DEV void run_shared(const KernelParams& p, float2* lds_fft, raw_t* lds_halo) {
    static_assert(NT == 64, "one wave per run");
    const int wave = threadIdx.x >> 6, tid = threadIdx.x & 63;
    long long blk = blockIdx.x;
    blk = xcd_remap_block(blk, gridDim.x, p.xcd_remap);
    const long long f_blk = blk * (long long)(NWV * L);
    const long long f_begin = f_blk + (long long)wave * L;
    Consts k;
    setup(p, tid, k);
    float2* lds = lds_fft + wave * K::LDS_ELEMS;
    raw_t* halo_mine = lds_halo + wave * ((W - 1) * D);
    const raw_t* halo_next = lds_halo + (wave + 1) * ((W - 1) * D);
    // workgroup-uniform: every row of every run lies inside `in`, whole runs only, aligned vectors
    const bool interior = p.vec_ok && ((f_blk - (W - 1)) * D + p.base >= 0) && (f_blk + NWV * L <= p.frames);
    if (interior) shared_impl<true, NWV, L>(p, k, lds, halo_mine, halo_next, wave, f_begin);
    else shared_impl<false, NWV, L>(p, k, lds, halo_mine, halo_next, wave, f_begin);
  }

  // ---- schedule F: schedule D with the FIR and the FFT on different waves ---------------------------
  // With one wave doing both halves the kernel needs ~114 VGPRs (4 waves per SIMD), and at 4 waves per
  // SIMD it sits on a latency floor (fusing abs() halves the written bytes and barely changes the time).
  // Here wave w < NPAIR slides the window and writes branch outputs for run w into one of two LDS
  // buffers while wave w + NPAIR transforms and stores the chunk before it: each role needs far fewer
  // registers, so more waves fit per SIMD.  One workgroup barrier per chunk hands the buffers over.
  template <bool INTERIOR, int NPAIR, int L, int DEPTH = 1>
  PFB_DEV void paired_fir_role(const KernelParams& p, float2* bufs, raw_t* halo_mine, const raw_t* halo_next,
                               int pair, long long f_begin) {
    constexpr int TAIL0 = L - (W - 1), NCH = L / C;
    static_assert(DEPTH == 1 || DEPTH == 2, "chunks of rows in flight");
    const int tid = threadIdx.x & 63;
    const int c0 = tid * CPT;
    Consts k;
    setup(p, tid, k);
    const raw_t* run_ptr = static_cast<const raw_t*>(p.in) + ((f_begin - (W - 1)) * D + p.base);
    const bool lane_on = !(K::LANES < NT) || tid < K::LANES;
    v2f x[NW][CPT];
    raw_t raw[DEPTH][C][CPT];
#pragma unroll
    for (int i = 0; i < W - 1; ++i) {
      raw_t t[CPT];
      load_row<INTERIOR>(p, run_ptr, f_begin - (W - 1) + i, i, c0, t);
#pragma unroll
      for (int cc = 0; cc < CPT; ++cc) {
        x[i][cc] = cvt(t[cc]);
        if constexpr (INTERIOR) { if (lane_on) halo_mine[i * D + c0 + cc] = t[cc]; }
      }
    }
    const bool tail_from_lds = INTERIOR && (pair < NPAIR - 1);
    auto fetch = [&](int cj) {  // rows of chunk cj (those not coming from the neighbour's halo slot) into raw[cj % DEPTH]
#pragma unroll
      for (int t = 0; t < C; ++t) {
        const int r = cj * C + t;
        if (!(r >= TAIL0 && tail_from_lds)) load_row<INTERIOR>(p, run_ptr, f_begin + r, W - 1 + r, c0, raw[cj % DEPTH][t]);
      }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
      if (d < NCH) fetch(d);
    __syncthreads();  // A: halo slots published
#pragma unroll
    for (int ci = 0; ci < NCH; ++ci) {
#pragma unroll
      for (int t = 0; t < C; ++t) {
        const int r = ci * C + t;
        if (r >= TAIL0 && tail_from_lds) {
#pragma unroll
          for (int cc = 0; cc < CPT; ++cc)
            x[W - 1 + t][cc] = cvt(halo_next[(r - TAIL0) * D + (lane_on ? c0 + cc : 0)]);
        } else {
#pragma unroll
          for (int cc = 0; cc < CPT; ++cc) x[W - 1 + t][cc] = cvt(raw[ci % DEPTH][t][cc]);
        }
      }
      if (ci + DEPTH < NCH) fetch(ci + DEPTH);
      fir_to_lds(k, x, bufs + (ci & 1) * K::BUF, tid);
#pragma unroll
      for (int i = 0; i < W - 1; ++i)
#pragma unroll
        for (int cc = 0; cc < CPT; ++cc) x[i][cc] = x[i + C][cc];
      __syncthreads();  // chunk ci handed to the FFT wave
    }
    __syncthreads();    // the FFT wave's last step
  }

  template <int NPAIR, int L>
  PFB_DEV void run_paired(const KernelParams& p, float2* lds_fft, raw_t* lds_halo) {
    static_assert(NT == 64, "one wave per run and role");
    static_assert(L % C == 0 && L >= W - 1, "runs are whole chunks and at least one halo long");
    constexpr int NCH = L / C;
    const int wave = threadIdx.x >> 6, tid = threadIdx.x & 63;
    const bool fir_role = wave < NPAIR;
    const int pair = fir_role ? wave : wave - NPAIR;
    long long blk = blockIdx.x;
    blk = xcd_remap_block(blk, gridDim.x, p.xcd_remap);
    const long long f_blk = blk * (long long)(NPAIR * L);
    const long long f_begin = f_blk + (long long)pair * L;
    float2* bufs = lds_fft + pair * 2 * K::BUF;
    const bool interior = p.vec_ok && ((f_blk - (W - 1)) * D + p.base >= 0) && (f_blk + NPAIR * L <= p.frames);
    if (fir_role) {
      raw_t* halo_mine = lds_halo + pair * ((W - 1) * D);
      const raw_t* halo_next = lds_halo + (pair + 1) * ((W - 1) * D);
      // (DEPTH = 2, rows two chunks ahead, measured on cfg2: 2.255 vs 2.243 ms -- no gain, 126 VGPRs; one chunk ahead stays)
      if (interior) paired_fir_role<true, NPAIR, L>(p, bufs, halo_mine, halo_next, pair, f_begin);
      else paired_fir_role<false, NPAIR, L>(p, bufs, halo_mine, halo_next, pair, f_begin);
    } else {
      Consts k;
      setup(p, tid, k);
      __syncthreads();  // A
#pragma unroll
      for (int s = 0; s <= NCH; ++s) {
        if (s >= 1) fft_from_lds(p, k, bufs + ((s - 1) & 1) * K::BUF, tid, f_begin + (long long)(s - 1) * C);
        __syncthreads();
      }
    }
  }

  // ---- schedule H: wave pairs over long sliding runs ------------------------------------------------------
  // Schedule F's split of the work (a FIR wave and an FFT wave per run, an LDS double buffer between them, one
  // workgroup barrier per chunk) without its halo sharing: every pair slides over its own long run of
  // frames_per_block frames like schedule A and re-reads only its own W-1 halo rows once.  For the shapes whose
  // single-wave kernel needs close to 200 registers (cfg5: 24 taps and a 31-row window per lane plus a radix-16
  // pass) this halves the registers per wave and doubles the waves per CU; the runs are a runtime loop, so
  // they can be long.
  template <bool INTERIOR>
  PFB_DEV void pair_fir_run(const KernelParams& p, const Consts& k, float2* bufs, long long f_begin, int nch) {
    const int tid = threadIdx.x & 63;
    const int c0 = tid * CPT;
    const raw_t* run_ptr = static_cast<const raw_t*>(p.in) + ((f_begin - (W - 1)) * D + p.base);
    v2f x[NW][CPT];
    raw_t raw[C][CPT];
#pragma unroll
    for (int i = 0; i < W - 1; ++i) {
      raw_t t[CPT];
      load_row<INTERIOR>(p, run_ptr, f_begin - (W - 1) + i, i, c0, t);
#pragma unroll
      for (int cc = 0; cc < CPT; ++cc) x[i][cc] = cvt(t[cc]);
    }
    RowFetch rf;
    begin_rows<INTERIOR>(run_ptr, rf);
    load_rows<INTERIOR>(p, run_ptr, f_begin, W - 1, c0, raw, rf);
    for (int ci2 = 0; ci2 < nch; ci2 += 2) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int ci = ci2 + u;
        finish_rows(c0, raw, rf);
#pragma unroll
        for (int t = 0; t < C; ++t)
#pragma unroll
          for (int cc = 0; cc < CPT; ++cc) x[W - 1 + t][cc] = cvt(raw[t][cc]);
        if (ci + 1 < nch) {
          const long long rel = (long long)(ci + 1) * C + (W - 1);
          load_rows<INTERIOR>(p, run_ptr, f_begin + (long long)(ci + 1) * C, rel, c0, raw, rf);
        }
        fir_to_lds(k, x, bufs + u * K::BUF, tid);
#pragma unroll
        for (int i = 0; i < W - 1; ++i)
#pragma unroll
          for (int cc = 0; cc < CPT; ++cc) x[i][cc] = x[i + C][cc];
        __syncthreads();  // chunk ci handed to the FFT wave
      }
    }
    __syncthreads();      // the FFT wave's last step
  }

  template <int NPAIR>
  PFB_DEV void run_pairs_sliding(const KernelParams& p, float2* lds_fft) {
    static_assert(NT == 64 && K::NP == 2 && !K::PINGPONG, "one wave per role, two in-place passes");
    const int wave = threadIdx.x >> 6, tid = threadIdx.x & 63;
    const bool fir_role = wave < NPAIR;
    const int pair = fir_role ? wave : wave - NPAIR;
    long long blk = blockIdx.x;
    blk = xcd_remap_block(blk, gridDim.x, p.xcd_remap);
    const long long f_begin = (blk * NPAIR + pair) * (long long)p.frames_per_block;
    const int nch = p.frames_per_block / C;  // even (host rounds); pairs past the end of the stream idle through the barriers
    float2* bufs = lds_fft + pair * 2 * K::BUF;
    Consts k;
    setup(p, tid, k);
    if (fir_role) {
      const bool interior = p.vec_ok && ((f_begin - (W - 1)) * D + p.base >= 0) && (f_begin + p.frames_per_block <= p.frames);
      if (f_begin >= p.frames) {
        for (int s = 0; s <= nch; ++s) __syncthreads();
      } else if (interior) {
        pair_fir_run<true>(p, k, bufs, f_begin, nch);
      } else {
        pair_fir_run<false>(p, k, bufs, f_begin, nch);
      }
    } else {
#pragma unroll 1
      for (int s = 0; s <= nch; ++s) {
        if (s >= 1 && f_begin < p.frames) fft_from_lds(p, k, bufs + ((s - 1) & 1) * K::BUF, tid, f_begin + (long long)(s - 1) * C);
        __syncthreads();
      }
    }
  }

  // all NW rows (halo included) of one chunk: schedule C's unit of work
  PFB_DEV void load_chunk(const KernelParams& p, long long chunk, int c0, raw_t (&raw)[NW][CPT]) {
    const long long f0 = chunk * C;
    const long long s_first = (f0 - (W - 1)) * D + p.base;
    const bool interior = p.vec_ok && s_first >= 0 && (f0 + C <= p.frames);
    const raw_t* ptr = static_cast<const raw_t*>(p.in) + s_first;  // uniform; only used when interior
    if (interior) {
#pragma unroll
      for (int i = 0; i < NW; ++i) load_row<true>(p, ptr, 0, i, c0, raw[i]);
    } else {
#pragma unroll
      for (int i = 0; i < NW; ++i) load_row<false>(p, ptr, f0 - (W - 1) + i, i, c0, raw[i]);
    }
  }

  // ---- schedule C: one chunk per wave, NWV adjacent chunks per (non-persistent) workgroup ---------
  // The dispatcher hands out workgroups in order, so the chip sweeps the stream as one compact,
  // monotonically advancing window (the fastest shape in tools/membench2); the W-1 halo rows a wave
  // shares with its neighbours in the workgroup are served by that CU's L1, and the ones shared with
  // the previous workgroup by the XCD's L2 (consecutive tiles are remapped onto one XCD).
  template <int NWV>
  PFB_DEV void run_tile(const KernelParams& p, float2* lds_all) {
    static_assert(NT == 64, "one wave per chunk");
    const int wave = threadIdx.x >> 6, tid = threadIdx.x & 63;
    const int c0 = tid * CPT;
    const long long nchunks = (p.frames + C - 1) / C;
    long long tile = blockIdx.x;
    tile = xcd_remap_block(tile, gridDim.x, p.xcd_remap);
    const long long chunk = tile * NWV + wave;
    if (chunk >= nchunks) return;
    float2* lds = lds_all + wave * K::LDS_ELEMS;
    Consts k;
    setup(p, tid, k);
    raw_t raw[NW][CPT];
    load_chunk(p, chunk, c0, raw);
    v2f x[NW][CPT];
#pragma unroll
    for (int i = 0; i < NW; ++i)
#pragma unroll
      for (int cc = 0; cc < CPT; ++cc) x[i][cc] = cvt(raw[i][cc]);
    fir_fft_store<true>(p, k, x, lds, tid, chunk * C);
  }

  // ---- schedule C', channel-major: short sliding runs whose output is transposed in LDS ----------------------
  // Channel-major rows are out_ld elements apart, so the last pass's natural store (a few frames of 8 channels
  // per instruction) scatters 64-byte pieces over 8 DRAM pages -- tools/membench5: 4.6 TB/s write-only, 1.2 TB/s
  // when out_ld is a power of two.  Here wave w of the workgroup slides over CPW chunks, each chunk ends
  // transposed in its own LDS slot (last_pass_transposed: no extra buffer), and after one barrier the workgroup
  // writes the NWV * CPW * C frames of every column as one run: an instruction is 256-512 contiguous bytes of
  // one or two columns (5.5 TB/s in the same microbenchmark, whatever out_ld is).
  static constexpr int TSLOT = K::LDS_ELEMS + ((C + 32 - K::LDS_ELEMS % 32) % 32);  // = C (mod 32): slots on distinct banks

  template <bool INTERIOR, int CPW>
  PFB_DEV void tile_t_impl(const KernelParams& p, float2* slots, int tid, long long f_begin) {
    const int c0 = tid * CPT;
    Consts k;
    setup(p, tid, k);
    const raw_t* run_ptr = static_cast<const raw_t*>(p.in) + ((f_begin - (W - 1)) * D + p.base);
    v2f x[NW][CPT];
    raw_t raw[C][CPT];
#pragma unroll
    for (int i = 0; i < W - 1; ++i) {
      raw_t t[CPT];
      load_row<INTERIOR>(p, run_ptr, f_begin - (W - 1) + i, i, c0, t);
#pragma unroll
      for (int cc = 0; cc < CPT; ++cc) x[i][cc] = cvt(t[cc]);
    }
    RowFetch rf;
    begin_rows<INTERIOR>(run_ptr, rf);
    load_rows<INTERIOR>(p, run_ptr, f_begin, W - 1, c0, raw, rf);
#pragma unroll
    for (int ci = 0; ci < CPW; ++ci) {
      finish_rows(c0, raw, rf);
#pragma unroll
      for (int t = 0; t < C; ++t)
#pragma unroll
        for (int cc = 0; cc < CPT; ++cc) x[W - 1 + t][cc] = cvt(raw[t][cc]);
      if (ci + 1 < CPW) load_rows<INTERIOR>(p, run_ptr, f_begin + (ci + 1) * C, W - 1 + (ci + 1) * C, c0, raw, rf);
      fir_fft_store<true, true>(p, k, x, slots + ci * TSLOT, tid, f_begin + ci * C);
      if (ci + 1 < CPW) {
#pragma unroll
        for (int i = 0; i < W - 1; ++i)
#pragma unroll
          for (int cc = 0; cc < CPT; ++cc) x[i][cc] = x[i + C][cc];
      }
    }
  }

  template <int NWV, int CPW>
  PFB_DEV void run_tile_t(const KernelParams& p, float2* lds_all) {
    static_assert(NT == 64 && CM, "one wave per run, channel-major output");
    constexpr int RL = NWV * CPW * C, NTH = 64 * NWV, IT = (M * RL) / NTH;
    static_assert((M * RL) % NTH == 0 && (RL & (RL - 1)) == 0 && RL % 32 == 0, "whole flush iterations over 32-frame blocks");
    const int wave = threadIdx.x >> 6, tid = threadIdx.x & 63;
    long long tile = blockIdx.x;
    tile = xcd_remap_block(tile, gridDim.x, p.xcd_remap);
    const long long tf0 = tile * RL;
    if (tf0 >= p.frames) return;  // workgroup-uniform
    const long long f_begin = tf0 + (long long)wave * (CPW * C);
    if (f_begin < p.frames) {
      float2* slots = lds_all + wave * (CPW * TSLOT);
      const bool interior = p.vec_ok && ((f_begin - (W - 1)) * D + p.base >= 0) && (f_begin + CPW * C <= p.frames);
      if (interior) tile_t_impl<true, CPW>(p, slots, tid, f_begin);
      else tile_t_impl<false, CPW>(p, slots, tid, f_begin);
    }
    team_sync<NWV == 1>();
    const bool mag = (p.flags & PFB_FLAG_MAGNITUDE) != 0;
    const v2f* t2 = reinterpret_cast<const v2f*>(lds_all);
    constexpr int HALF = IT > 8 ? 2 : 1;  // at most 8 values in flight per lane
#pragma unroll
    for (int h = 0; h < HALF; ++h) {
      v2f v[IT / HALF];
#pragma unroll
      for (int i = 0; i < IT / HALF; ++i) {
        const int e = (h * (IT / HALF) + i) * NTH + (int)threadIdx.x, col = e / RL, fr = e % RL;
        v[i] = t2[(fr / C) * TSLOT + col * C + tslot_frame(col, fr % C)];
      }
#pragma unroll
      for (int i = 0; i < IT / HALF; ++i) {
        const int e = (h * (IT / HALF) + i) * NTH + (int)threadIdx.x, col = e / RL, fr = e % RL;
        const long long f = tf0 + fr;
        if (f < p.frames) {
          const long long idx = (long long)col * p.out_ld + p.out_frame0 + f;
          if (mag) reinterpret_cast<float*>(p.out)[idx] = mag_out(v[i].x, v[i].y, p.flags);
          else store_c64(p.out + idx, v[i], p.nontemporal);
        }
      }
    }
  }

};

// ---------------------------------------------------------------------------------
// Small banks (M = 8, 10, 16, 20, 40, ...: numBands = fs * 1e-6 at 8 ... 40 Msps).  With one column per lane only M of the
// wave's 64 lanes would filter.  Here the workgroup's run of frames is cut into SEG = 64 / M contiguous
// segments and lane (seg, col) slides column col's window over segment seg: all 64 lanes filter, a chunk is
// C * SEG frames, and the two FFT passes run over all of them (ping-pong LDS buffers, twiddles from the table).
// Same tables, same arithmetic per output as FastKernel; critically sampled, two-pass plans only.
template <class K>
struct SegKernel {
  using ST = SampleT<K::FMT>;
  using raw_t = typename ST::raw_t;
  static constexpr int M = K::M, P = K::P, D = K::D, C = K::C, W = K::W;
  static constexpr int SEG = 64 / M, CT = C * SEG, NW = W - 1 + C;  // M * SEG lanes work, the rest (M not dividing 64) idle
  static_assert(K::NT == 64 && K::CPT == 1 && K::OS == 1 && K::NP == 2 && K::PINGPONG && SEG >= 1, "small banks");

  PFB_DEV v2f cvt(raw_t r) {
    float re, im;
    ST::cvt(r, re, im);
    return (v2f){re, im};
  }

  // sample `s` of the stream (index relative to this call's buffer; negative = history), row of frame `f`
  template <bool INTERIOR>
  PFB_DEV raw_t load(const KernelParams& p, long long s, long long f) {
    const raw_t* in = static_cast<const raw_t*>(p.in);
    if constexpr (INTERIOR) {
      return in[s];
    } else {
      if (f >= p.frames) return raw_t{};
      return (s >= 0) ? in[s] : static_cast<const raw_t*>(p.hist)[p.hist_samples + s];
    }
  }

  // FULL / MAGSEL / CMSEL: every frame exists (an interior run) and PFB_FLAG_MAGNITUDE / the output layout are template
  // parameters of the kernel: one store per output on every path, so the compiler's s_waitcnt counts stay exact across
  // the chunk loop (FastKernel::pass<FULL>)
  template <int I, bool FULL = false, int MAGSEL = -1, int CMSEL = -1>
  PFB_DEV void pass(const KernelParams& p, const float2* src, float2* dst, int tid, long long f_begin, long long l_seg,
                    long long chunk0) {
    constexpr int R = K::R(I), S = K::S(I), KK = K::K(I), RS = K::RS(I);
    constexpr int IPF = M / R, ITEMS = CT * IPF, ITERS = (ITEMS + 63) / 64;
    constexpr bool LAST = (I == 1);
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int w = tid + it * 64;
      const bool active = (ITEMS % 64 == 0) || (w < ITEMS);
      const int fc = active ? w / IPF : 0, item = active ? w % IPF : 0;
      const int kk = item / S, rest = item % S;
      v2f x[R];
      const v2f* s2 = reinterpret_cast<const v2f*>(src) + fc * K::FS + item;
#pragma unroll
      for (int n = 0; n < R; ++n) x[n] = s2[n * RS];
      Dft<R>::run(x);
      if constexpr (!LAST) {
        constexpr int S1 = K::S(I + 1), RS1 = K::RS(I + 1);
        if constexpr (R % 2 == 0) {
          const float4* t4 = reinterpret_cast<const float4*>(p.tw_lane + K::TW_OFF(I) + rest * K::TWR(I));
#pragma unroll
          for (int k2 = 0; k2 < R / 2; ++k2) {
            const float4 t = t4[k2];
            if (k2 > 0) x[2 * k2] = cmul_w(x[2 * k2], (v2f){t.x, t.y});
            x[2 * k2 + 1] = cmul_w(x[2 * k2 + 1], (v2f){t.z, t.w});
          }
        } else {  // odd radix (rows padded to an even length): element by element
          const float2* t2 = p.tw_lane + K::TW_OFF(I) + rest * K::TWR(I);
#pragma unroll
          for (int k = 1; k < R; ++k) x[k] = cmul_w(x[k], (v2f){t2[k].x, t2[k].y});
        }
        if (active) {
          const int n1 = rest / S1, rest2 = rest % S1;
          v2f* d2 = reinterpret_cast<v2f*>(dst) + fc * K::FS + n1 * RS1 + kk * S1 + rest2;
#pragma unroll
          for (int k = 0; k < R; ++k) d2[k * KK * S1] = x[k];
        }
      } else {
        const long long f = f_begin + (fc / C) * l_seg + chunk0 + (fc % C);  // frame (segment fc / C, chunk, t)
        if (active && (FULL || f < p.frames)) {
          const int shift = (p.flags & PFB_FLAG_FFTSHIFT) ? (M / 2) : 0;
          const bool mag = MAGSEL >= 0 ? MAGSEL == 1 : (p.flags & PFB_FLAG_MAGNITUDE) != 0;
          const bool cm = CMSEL >= 0 ? CMSEL == 1 : p.layout == PFB_LAYOUT_CHANNEL_MAJOR;
#pragma unroll
          for (int k = 0; k < R; ++k) {
            int col = kk + k * KK + shift;
            col = col >= M ? col - M : col;
            const long long o = cm ? (long long)col * p.out_ld + p.out_frame0 + f : f * M + col;
            if (mag) reinterpret_cast<float*>(p.out)[o] = mag_out(x[k].x, x[k].y, p.flags);
            else if (MAGSEL >= 0) *reinterpret_cast<v2f*>(&p.out[o]) = x[k];
            else store_c64(&p.out[o], x[k], p.nontemporal);
          }
        }
      }
    }
  }

  template <bool INTERIOR, int MAGSEL = -1, int CMSEL = -1>
  PFB_DEV void run_impl(const KernelParams& p, float2* lds, long long f_begin, long long l_seg) {
    const int tid = threadIdx.x;
    const bool lane_on = (64 % M == 0) || tid < SEG * M;
    const int seg = lane_on ? tid / M : 0, col = lane_on ? tid % M : 0;  // idle lanes shadow lane 0 and never write
    const long long f_seg = f_begin + seg * l_seg;                  // my segment's first frame
    const long long s_row0 = (f_seg - (W - 1)) * D + p.base + col;  // my column in the first halo row
    // taps of my column, two per register pair (the same table FastKernel::setup reads)
    v2f hp[(W + 1) / 2];
    {
      const float4* tl = reinterpret_cast<const float4*>(p.taps_lane + (size_t)col * K::WP);
#pragma unroll
      for (int q4 = 0; q4 < K::WP / 4; ++q4) {
        const float4 v = tl[q4];
        if (2 * q4 < (W + 1) / 2) hp[2 * q4] = (v2f){v.x, v.y};
        if (2 * q4 + 1 < (W + 1) / 2) hp[2 * q4 + 1] = (v2f){v.z, v.w};
      }
    }
    const v2f conj_mul = (v2f){1.f, (p.flags & PFB_FLAG_CONJUGATE_INPUT) ? -1.f : 1.f};
    const int n = D - 1 - col;  // my branch
    const int upos = (n / K::S(0)) * K::RS(0) + (n % K::S(0));
    float2* buf0 = lds;
    float2* buf1 = lds + CT * K::FS;
    v2f x[NW];
    raw_t raw[C];
#pragma unroll
    for (int i = 0; i < W - 1; ++i) x[i] = cvt(load<INTERIOR>(p, s_row0 + (long long)i * D, f_seg - (W - 1) + i));
#pragma unroll
    for (int t = 0; t < C; ++t) raw[t] = load<INTERIOR>(p, s_row0 + (long long)(W - 1 + t) * D, f_seg + t);
    // (rotated like FastKernel::run_impl: the rows requested at the top of an iteration are taken at its end)
#pragma unroll
    for (int t = 0; t < C; ++t) x[W - 1 + t] = cvt(raw[t]);
    for (long long c0 = 0; c0 < l_seg; c0 += C) {
      if constexpr (INTERIOR) {  // unconditional: past the segment's end its last chunk again
        const long long cn = c0 + C < l_seg ? c0 + C : c0;
#pragma unroll
        for (int t = 0; t < C; ++t) raw[t] = load<true>(p, s_row0 + (cn + (W - 1) + t) * D, f_seg + cn + t);
      } else if (c0 + C < l_seg) {
#pragma unroll
        for (int t = 0; t < C; ++t)
          raw[t] = load<INTERIOR>(p, s_row0 + (c0 + C + (W - 1) + t) * D, f_seg + c0 + C + t);
      }
      v2f acc[C];
      int tok = 0;  // FMA ordering token (fma_tap_lo)
#pragma unroll
      for (int q = 0; q < P; ++q)
#pragma unroll
        for (int t = 0; t < C; ++t) {
          if (q == 0) fma_tap0_lo(acc[t], x[W - 1 + t], hp[0]);
          else if (q & 1) fma_tap_hi(acc[t], x[W - 1 + t - q], hp[q >> 1], tok);
          else fma_tap_lo(acc[t], x[W - 1 + t - q], hp[q >> 1], tok);
        }
      if (lane_on) {
#pragma unroll
        for (int t = 0; t < C; ++t) reinterpret_cast<v2f*>(buf0)[(seg * C + t) * K::FS + upos] = acc[t] * conj_mul;
      }
      team_sync<true>();
      pass<0>(p, buf0, buf1, tid, f_begin, l_seg, c0);
      team_sync<true>();
      pass<1, INTERIOR, MAGSEL, CMSEL>(p, buf1, nullptr, tid, f_begin, l_seg, c0);
      team_sync<true>();
#pragma unroll
      for (int i = 0; i < W - 1; ++i) x[i] = x[i + C];
      if (INTERIOR || c0 + C < l_seg) {
#pragma unroll
        for (int t = 0; t < C; ++t) x[W - 1 + t] = cvt(raw[t]);
      }
    }
  }

  template <int MAGSEL = -1, int CMSEL = -1>
  PFB_DEV void run(const KernelParams& p, float2* lds) {
    long long run = blockIdx.x;
    run = xcd_remap_block(run, gridDim.x, p.xcd_remap);
    const long long f_begin = run * p.frames_per_block;
    if (f_begin >= p.frames) return;
    const long long l_seg = p.frames_per_block / SEG;  // host rounds frames_per_block to a multiple of C * SEG
    const bool interior = p.vec_ok && ((f_begin - (W - 1)) * D + p.base >= 0) && (f_begin + p.frames_per_block <= p.frames);
    if (interior) run_impl<true, MAGSEL, CMSEL>(p, lds, f_begin, l_seg);
    else run_impl<false, MAGSEL, CMSEL>(p, lds, f_begin, l_seg);
  }
};

template <class K, bool MAG, bool CMAJ>
__global__ void __launch_bounds__(64, K::MIN_WAVES) pfb_seg_kernel(const KernelParams p) {
  __shared__ float2 lds[2 * SegKernel<K>::CT * K::FS];
  SegKernel<K>::template run<MAG ? 1 : 0, CMAJ ? 1 : 0>(p, lds);
}

template <class K>
hipError_t launch_seg(const KernelParams& p, hipStream_t s) {
  if (p.frames <= 0) return hipSuccess;
  const long long nb = (p.frames + p.frames_per_block - 1) / p.frames_per_block;
  const bool mag = (p.flags & PFB_FLAG_MAGNITUDE) != 0, cm = p.layout == PFB_LAYOUT_CHANNEL_MAJOR;
  if (mag && cm) hipLaunchKernelGGL((pfb_seg_kernel<K, true, true>), dim3((unsigned)nb), dim3(64), 0, s, p);
  else if (mag) hipLaunchKernelGGL((pfb_seg_kernel<K, true, false>), dim3((unsigned)nb), dim3(64), 0, s, p);
  else if (cm) hipLaunchKernelGGL((pfb_seg_kernel<K, false, true>), dim3((unsigned)nb), dim3(64), 0, s, p);
  else hipLaunchKernelGGL((pfb_seg_kernel<K, false, false>), dim3((unsigned)nb), dim3(64), 0, s, p);
  return hipGetLastError();
}

// Builds the per-column tap table and the inter-pass twiddle rows (once per handle).
template <class K>
__global__ void __launch_bounds__(256) pfb_init_tables_kernel(const float* taps, const float2* tw, float* taps_lane,
                                                             float2* tw_lane) {
  for (int idx = threadIdx.x; idx < K::TAPS_LANE_FLOATS; idx += 256) {
    const int c = idx / K::WP, j = idx % K::WP;
    taps_lane[idx] = (j < K::W) ? taps[(K::D - 1 - c) + K::D * j] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < K::NP - 1; ++i) {
    const int R = K::R(i), S = K::S(i), KK = K::K(i);
    const int RP = K::TWR(i);
    for (int idx = threadIdx.x; idx < S * RP; idx += 256) {
      const int rest = idx / RP, kk = idx % RP;
      tw_lane[K::TW_OFF(i) + idx] = kk < R ? tw[rest * kk * KK] : make_float2(0.f, 0.f);
    }
  }
}

template <class K>
hipError_t init_tables(const float* taps, const float2* tw, float* taps_lane, float2* tw_lane, hipStream_t s) {
  hipLaunchKernelGGL(pfb_init_tables_kernel<K>, dim3(1), dim3(256), 0, s, taps, tw, taps_lane, tw_lane);
  return hipGetLastError();
}

// plans without a channel-major instantiation: it does not fit the register budget (the 1024-thread cfg4 plan
// sits at its 128-VGPR ceiling already) or would be the shape's worst (chunks of 4 frames); channel-major
// handles get the shape's next registered plan instead
// fused abs() has a faster schedule than complex output on the shapes whose last pass can stage its magnitudes
// in LDS (FastKernel::pass, kMagStaged): sliding runs
template <class K>
constexpr bool kMagStagedOk = K::NT == 64 && K::NP == 2 && !K::PINGPONG && K::M == 64 && K::C == 8;
// (M = 64 only: there the direct stores are 32-byte pieces.  Measured elsewhere: cfg3, whose pieces are 64 bytes,
// -3 %; M = 32 +-0; the M = 56 sliding kernel spilled with it)
template <class K>
constexpr int kMagnitudeSchedule = kMagStagedOk<K> ? (K::FMT == PFB_FMT_CF32 ? 7 : 0) : -1;  // (cf32: pairs still win)

template <class K>
constexpr bool kChannelMajorOk = K::NT < 1024 && !(K::NP == 3 && K::C == 4);  // (C = 4 team plans: 32-byte runs)

// MAGSEL: -1 = PFB_FLAG_MAGNITUDE is tested inside (channel-major and staged-magnitude instantiations), 0 / 1 = decided
// at launch (the frame-major kernels: their store count per chunk is then path-independent, see run_impl)
template <class K, bool CM = false, bool MS = false, int MAGSEL = -1>
__global__ void __launch_bounds__(K::NT, K::MIN_WAVES) pfb_fast_kernel(const KernelParams p) {
  __shared__ float2 lds[K::LDS_ELEMS];
  FastKernel<K, CM, MS>::template run<MAGSEL>(p, lds);
}

template <class K, bool MAG>
__global__ void __launch_bounds__(K::NT, (K::MIN_WAVES > 2 ? K::MIN_WAVES - 1 : K::MIN_WAVES)) pfb_overlap_kernel(const KernelParams p) {
  __shared__ float2 lds[2 * K::BUF];
  FastKernel<K>::template run_overlap<MAG ? 1 : 0>(p, lds);
}

template <class K>
constexpr bool kOverlapOk = K::NT == 64 && K::NP == 2 && !K::PINGPONG;

template <class K, int NWV, bool CM = false>
__global__ void __launch_bounds__(64 * NWV) pfb_tile_kernel(const KernelParams p) {
  __shared__ float2 lds[NWV * K::LDS_ELEMS];
  FastKernel<K, CM>::template run_tile<NWV>(p, lds);
}

template <class K, int NWV, bool CM = false>
hipError_t launch_tile(const KernelParams& p, hipStream_t s) {
  const long long nchunks = (p.frames + K::C - 1) / K::C;
  const long long tiles = (nchunks + NWV - 1) / NWV;
  hipLaunchKernelGGL((pfb_tile_kernel<K, NWV, CM>), dim3((unsigned)tiles), dim3(64 * NWV), 0, s, p);
  return hipGetLastError();
}

template <class K, int NWV, int CPW>
__global__ void __launch_bounds__(64 * NWV) pfb_tile_t_kernel(const KernelParams p) {
  __shared__ float2 lds[NWV * CPW * FastKernel<K, true>::TSLOT];
  FastKernel<K, true>::template run_tile_t<NWV, CPW>(p, lds);
}

template <class K, int NWV, int CPW>
hipError_t launch_tile_t(const KernelParams& p, hipStream_t s) {
  constexpr int RL = NWV * CPW * K::C;
  const long long tiles = (p.frames + RL - 1) / RL;
  hipLaunchKernelGGL((pfb_tile_t_kernel<K, NWV, CPW>), dim3((unsigned)tiles), dim3(64 * NWV), 0, s, p);
  return hipGetLastError();
}

// the transposed tile: single-wave two-pass plans whose chunk buffer holds the transposed chunk, rows of whole
// 32-frame blocks
// (the slot swizzle of tslot_frame: power-of-two chunk and lane groups)
template <class K>
constexpr bool kTileSlotOk = 16 % K::C == 0 && ((K::M / K::R(K::NP - 1)) >= 16 || 16 % (K::M / K::R(K::NP - 1)) == 0) &&
                             K::C % ((K::M / K::R(K::NP - 1)) >= 16 ? 1 : 16 / (K::M / K::R(K::NP - 1))) == 0;

template <class K, int NWV, int CPW>
constexpr bool kTileTOk = K::NT == 64 && K::NP == 2 && !K::PINGPONG && kTileSlotOk<K> && K::M * K::C <= K::LDS_ELEMS &&
                          (NWV * CPW * K::C) % 32 == 0 && ((NWV * CPW * K::C) & (NWV * CPW * K::C - 1)) == 0 &&
                          (K::M * NWV * CPW * K::C) % (64 * NWV) == 0;

template <class K, int NWV, int L>
__global__ void __launch_bounds__(64 * NWV) pfb_shared_kernel(const KernelParams p) {
  using raw_t = typename SampleT<K::FMT>::raw_t;
  __shared__ float2 lds_fft[NWV * K::LDS_ELEMS];
  __shared__ raw_t lds_halo[(NWV + 1) * (K::W - 1) * K::D];
  FastKernel<K>::template run_shared<NWV, L>(p, lds_fft, lds_halo);
}

template <class K, int NPAIR, int L, int MINW>
__global__ void __launch_bounds__(128 * NPAIR, MINW) pfb_paired_kernel(const KernelParams p) {
  using raw_t = typename SampleT<K::FMT>::raw_t;
  __shared__ float2 lds_fft[NPAIR * 2 * K::BUF];
  __shared__ raw_t lds_halo[(NPAIR + 1) * (K::W - 1) * K::D];
  FastKernel<K>::template run_paired<NPAIR, L>(p, lds_fft, lds_halo);
}

template <class K, int NPAIR, int L, int MINW>
hipError_t launch_paired(const KernelParams& p, hipStream_t s) {
  const long long per = (long long)NPAIR * L;
  const long long blocks = (p.frames + per - 1) / per;
  hipLaunchKernelGGL((pfb_paired_kernel<K, NPAIR, L, MINW>), dim3((unsigned)blocks), dim3(128 * NPAIR), 0, s, p);
  return hipGetLastError();
}

template <class K, int NPAIR, int MINW>
__global__ void __launch_bounds__(128 * NPAIR, MINW) pfb_pairs_sliding_kernel(const KernelParams p) {
  __shared__ float2 lds_fft[NPAIR * 2 * K::BUF];
  FastKernel<K>::template run_pairs_sliding<NPAIR>(p, lds_fft);
}

template <class K, int NPAIR, int MINW>
hipError_t launch_pairs_sliding(const KernelParams& p, hipStream_t s) {
  const long long per = (long long)NPAIR * p.frames_per_block;
  const long long blocks = (p.frames + per - 1) / per;
  hipLaunchKernelGGL((pfb_pairs_sliding_kernel<K, NPAIR, MINW>), dim3((unsigned)blocks), dim3(128 * NPAIR), 0, s, p);
  return hipGetLastError();
}

// shapes with a FIR-team / FFT-team instantiation: three in-place passes whose last pass fits the FIR team in one
// iteration per thread or more (the generic pass), a multi-wave FIR team, chunks of C frames = C FFT waves
template <class K>
constexpr bool kTeamsOk = !K::WAVE_FRAMES && K::NP == 3 && !K::PINGPONG && K::NT > 64 && (K::NT + 64 * K::C) <= 1024 &&
                          3 * sizeof(float2) * K::BUF <= 160 * 1024 && K::C % 2 == 0;

// schedule 13 (W): independent workgroups of NT/64 waves, a frame per wave and chunk, two or more per CU (run_twin)
template <class K>
constexpr bool kTwinOk = K::WAVE_FRAMES && K::NP == 3 && !K::PINGPONG && K::C % (K::NT / 64) == 0 && K::D == K::M && K::CPT % 2 == 0;

template <class K, bool MAG>
__global__ void __launch_bounds__(K::NT, K::MIN_WAVES) pfb_twin_kernel(const KernelParams p) {
  __shared__ float2 lds[K::BUF + FastKernel<K>::TWL_ELEMS];
  FastKernel<K>::template run_twin<MAG>(p, lds, lds + K::BUF);
}

// MAG: the handle's PFB_FLAG_MAGNITUDE, decided at launch -- inside the kernel the test made the number of stores per
// step look path-dependent to the compiler, whose s_waitcnt for the row prefetch then also waited for the stores
template <class K, bool MAG>
__global__ void __launch_bounds__(K::NT + 64 * K::C, K::MIN_WAVES) pfb_teams_kernel(const KernelParams p) {
  __shared__ float2 bufs[3 * K::BUF];
  FastKernel<K>::template run_teams<MAG ? 1 : 0>(p, bufs);
}

template <class K, int NWV, int L>
hipError_t launch_shared_impl(const KernelParams& p, hipStream_t s);

template <class K, int NWV>
constexpr bool kSharedFits = sizeof(float2) * NWV * K::LDS_ELEMS +
                                 sizeof(typename SampleT<K::FMT>::raw_t) * (NWV + 1) * (K::W - 1) * K::D <= 160 * 1024;

// a tile of NWV waves whose chunk buffers + shared halo do not fit one CU's LDS for this sample format (8-byte samples
// at M = 128) takes the 4-wave tile; hipErrorNotSupported = not even that: the caller goes on to the plain sliding runs
template <class K, int NWV, int L>
hipError_t launch_shared(const KernelParams& p, hipStream_t s) {
  if constexpr (kSharedFits<K, NWV>) {
    return launch_shared_impl<K, NWV, L>(p, s);
  } else if constexpr (NWV > 4 && kSharedFits<K, 4>) {
    return launch_shared_impl<K, 4, 64>(p, s);
  } else {
    return hipErrorNotSupported;
  }
}

template <class K, int NWV, int L>
hipError_t launch_shared_impl(const KernelParams& p, hipStream_t s) {
  const long long per = (long long)NWV * L;
  const long long blocks = (p.frames + per - 1) / per;
  // experiment bits 8.. : extra dynamic LDS in KiB (occupancy throttle for access-window studies)
  const unsigned extra_lds = (unsigned)((p.experiment >> 8) & 0xff) * 1024u;
  hipLaunchKernelGGL((pfb_shared_kernel<K, NWV, L>), dim3((unsigned)blocks), dim3(64 * NWV), extra_lds, s, p);
  return hipGetLastError();
}

template <class K>
hipError_t launch_fast(const KernelParams& p, hipStream_t s) {
  if (p.frames <= 0) return hipSuccess;
  if (p.layout == PFB_LAYOUT_CHANNEL_MAJOR) {  // schedules 8, 2 and 0; the others are frame-major tuning
    if constexpr (kChannelMajorOk<K>) {
      if constexpr (K::NT == 64) {
        if constexpr (kTileTOk<K, 4, 2>) {  // short runs transposed in LDS: schedule 8, the default where the plan allows
          if (p.schedule == 8 || p.schedule < 0) {
            // tile_waves = waves per workgroup, frames_per_block / C = chunks per wave
            const int key = p.tile_waves * 100 + (p.schedule == 8 ? p.frames_per_block / K::C : 0);
            switch (key) {
              case 401: if constexpr (kTileTOk<K, 4, 1>) return launch_tile_t<K, 4, 1>(p, s); break;
              case 801: if constexpr (kTileTOk<K, 8, 1>) return launch_tile_t<K, 8, 1>(p, s); break;
              case 202: if constexpr (kTileTOk<K, 2, 2>) return launch_tile_t<K, 2, 2>(p, s); break;
              case 402: if constexpr (kTileTOk<K, 4, 2>) return launch_tile_t<K, 4, 2>(p, s); break;
              case 104: if constexpr (kTileTOk<K, 1, 4>) return launch_tile_t<K, 1, 4>(p, s); break;
              case 204: if constexpr (kTileTOk<K, 2, 4>) return launch_tile_t<K, 2, 4>(p, s); break;
              default: break;
            }
            // measured best: 4 waves x 2 chunks on M = 56, 64, 128; 2 waves x 4 chunks on M = 32
            if constexpr (K::M <= 32 && kTileTOk<K, 2, 4>) return launch_tile_t<K, 2, 4>(p, s);
            return launch_tile_t<K, 4, 2>(p, s);
          }
        }
        // one chunk per wave, NWV adjacent chunks per workgroup: the workgroup writes NWV * C consecutive frames of
        // every channel at about the same time, which L2 merges into runs a sliding wave never produces by itself
        // (measured +17 ... +70 % over sliding runs; 16 waves win up to M = 64, 8 above)
        if (p.schedule == 2 || p.schedule < 0) {
          const int nwv = p.schedule < 0 ? (K::M <= 64 ? 16 : 8) : p.tile_waves;
          if constexpr (16 * sizeof(float2) * K::LDS_ELEMS <= 160 * 1024) {
            if (nwv == 16) return launch_tile<K, 16, true>(p, s);
          }
          return launch_tile<K, 8, true>(p, s);
        }
      }
      const long long nb = (p.frames + p.frames_per_block - 1) / p.frames_per_block;
      hipLaunchKernelGGL((pfb_fast_kernel<K, true>), dim3((unsigned)nb), dim3(K::NT), 0, s, p);
      return hipGetLastError();
    } else {
      return hipErrorInvalidValue;  // find_fast_kernel never hands this plan to a channel-major handle
    }
  }
  // (measured on cfg3, cfg5 and M=56 too: slower than their sliding runs, so only the M=64 kernels carry it)
  if constexpr (K::NT == 64 && K::NP == 2 && !K::PINGPONG) {  // wave pairs over long sliding runs
    if (p.schedule == 7) {
      // 8 pairs (16 waves, 128 registers each) where the roles fit that budget -- the shapes whose single-wave
      // kernel already runs 4 waves per SIMD -- otherwise 6 pairs (168 registers); tile_waves = 4 asks for 4
      constexpr size_t kPair = sizeof(float2) * 2 * K::BUF;
      // (workgroups of ONE or TWO pairs -- a barrier that couples 2 or 4 waves, 8 or 4 workgroups per CU -- measured: cfg2
      // 0.61-0.65 against 0.72 for its 8 halo-sharing pairs, M = 56 0.58-0.61 against 0.62, cfg5 0.44-0.49 against 0.68;
      // profiles/r03_tiny_pair_workgroups_ab.txt; four pairs at two workgroups per CU: M = 56 0.575-0.605 against 0.608)
      if (p.tile_waves == 4) return launch_pairs_sliding<K, 4, 2>(p, s);
      if constexpr (K::MIN_WAVES >= 4 && 8 * kPair <= 160 * 1024) {
        if (p.tile_waves >= 8) return launch_pairs_sliding<K, 8, 4>(p, s);
      }
      return launch_pairs_sliding<K, 6, 3>(p, s);
    }
  }
  if constexpr (kOverlapOk<K>) {  // sliding runs, FIR of the next chunk scheduled into the FFT of this one
    if (p.schedule == 11) {
      const long long nb = (p.frames + p.frames_per_block - 1) / p.frames_per_block;
      if (p.flags & PFB_FLAG_MAGNITUDE) hipLaunchKernelGGL((pfb_overlap_kernel<K, true>), dim3((unsigned)nb), dim3(K::NT), 0, s, p);
      else hipLaunchKernelGGL((pfb_overlap_kernel<K, false>), dim3((unsigned)nb), dim3(K::NT), 0, s, p);
      return hipGetLastError();
    }
  }
  if constexpr (kTwinOk<K>) {  // independent workgroups, a frame per wave and chunk
    if (p.schedule == 13) {
      if (p.layout != PFB_LAYOUT_FRAME_MAJOR) return hipErrorInvalidValue;
      long long nb = (p.frames + p.frames_per_block - 1) / p.frames_per_block;
      if (p.grid_override > 0 && nb > p.grid_override) nb = p.grid_override;  // resident workgroups walking runs b, b + G, ...
      if (p.flags & PFB_FLAG_MAGNITUDE) hipLaunchKernelGGL((pfb_twin_kernel<K, true>), dim3((unsigned)nb), dim3(K::NT), 0, s, p);
      else hipLaunchKernelGGL((pfb_twin_kernel<K, false>), dim3((unsigned)nb), dim3(K::NT), 0, s, p);
      return hipGetLastError();
    }
  }
  if constexpr (kTeamsOk<K>) {  // FIR team + FFT team
    if (p.schedule == 6) {
      const long long nb = (p.frames + p.frames_per_block - 1) / p.frames_per_block;
      if (p.flags & PFB_FLAG_MAGNITUDE) hipLaunchKernelGGL((pfb_teams_kernel<K, true>), dim3((unsigned)nb), dim3(K::NT + 64 * K::C), 0, s, p);
      else hipLaunchKernelGGL((pfb_teams_kernel<K, false>), dim3((unsigned)nb), dim3(K::NT + 64 * K::C), 0, s, p);
      return hipGetLastError();
    }
  }
  if constexpr (K::NT == 64 && K::NP == 2 && !K::PINGPONG && K::M == 64 && K::C == 8) {
    if (p.schedule == 4) {  // FIR / FFT wave pairs: tile_waves = pairs per workgroup, frames_per_block = run length
      const int key = p.tile_waves * 1000 + p.frames_per_block;
      if constexpr (K::FMT == PFB_FMT_INT16_IQ && K::M == 64 && K::P == 12) {  // tuning sweep set (cfg2 only, keeps build time sane)
        switch (key) {
          case 4064: return launch_paired<K, 4, 64, 4>(p, s);
          case 5064: return launch_paired<K, 5, 64, 5>(p, s);   // 10-wave workgroups, 5 waves per SIMD
          case 8048: return launch_paired<K, 8, 48, 4>(p, s);
          case 8128: return launch_paired<K, 8, 128, 4>(p, s);
          default: break;
        }
      }
      // the tuned shape is 8 pairs x 64 frames (16 waves, 512 frames per workgroup); instantiations whose
      // LDS image does not fit 8 pairs take 4
      constexpr size_t kPair = sizeof(float2) * 2 * K::BUF, kSlot = sizeof(typename SampleT<K::FMT>::raw_t) * (K::W - 1) * K::D;
      if constexpr (8 * kPair + 9 * kSlot <= 160 * 1024) {
        if (p.tile_waves == 4) return launch_paired<K, 4, 64, 2>(p, s);
        return launch_paired<K, 8, 64, 4>(p, s);
      } else {
        return launch_paired<K, 4, 64, 2>(p, s);
      }
    }
  }
  if constexpr (K::NT == 64 && K::C == 8 && !K::PINGPONG) {
    if (p.schedule == 3) {  // shared-halo sliding windows: tile_waves runs of frames_per_block frames
      const int key = p.tile_waves * 1000 + p.frames_per_block;
      hipError_t r = hipErrorNotSupported;
      if (key == 8024) r = launch_shared<K, 8, 24>(p, s);
      else if (key == 8032) r = launch_shared<K, 8, 32>(p, s);
      else if (key == 8064) r = launch_shared<K, 8, 64>(p, s);
      else if (key == 4064) r = launch_shared<K, 4, 64>(p, s);
      else {
        bool done = false;
        if constexpr (K::FMT == PFB_FMT_INT16_IQ && K::M == 64 && K::P == 12) {  // tuning sweep set (cfg2 only, keeps build time sane)
          switch (key) {
            case 4032: r = launch_shared<K, 4, 32>(p, s); done = true; break;
            case 8048: r = launch_shared<K, 8, 48>(p, s); done = true; break;
            case 16024: r = launch_shared<K, 16, 24>(p, s); done = true; break;
            default: break;
          }
        }
        if (!done) r = launch_shared<K, 8, 24>(p, s);  // any other shape: the tuned default
      }
      if (r != hipErrorNotSupported) return r;  // (no tile fits this sample format: the plain sliding runs below)
    }
  }
  if constexpr (K::NT == 64 && K::M == 64 && K::FMT == PFB_FMT_INT16_IQ && K::C == 8 && K::P == 12) {  // access-shape study schedules (cfg2 only)
    if (p.schedule == 2) {  // one chunk per wave, tile_waves adjacent chunks per workgroup
      if (p.tile_waves == 1) return launch_tile<K, 1>(p, s);
      return launch_tile<K, 8>(p, s);
    }
  }
  const long long blocks = (p.frames + p.frames_per_block - 1) / p.frames_per_block;
  if constexpr (kMagStagedOk<K>) {
    if ((p.flags & PFB_FLAG_MAGNITUDE) && (reinterpret_cast<uintptr_t>(p.out) & 15) == 0) {
      hipLaunchKernelGGL((pfb_fast_kernel<K, false, true>), dim3((unsigned)blocks), dim3(K::NT), 0, s, p);
      return hipGetLastError();
    }
  }
  if (p.flags & PFB_FLAG_MAGNITUDE) hipLaunchKernelGGL((pfb_fast_kernel<K, false, false, 1>), dim3((unsigned)blocks), dim3(K::NT), 0, s, p);
  else hipLaunchKernelGGL((pfb_fast_kernel<K, false, false, 0>), dim3((unsigned)blocks), dim3(K::NT), 0, s, p);
  return hipGetLastError();
}

}  // namespace pfb
