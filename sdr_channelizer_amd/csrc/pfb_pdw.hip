// pfb_pdw.hip -- channelized PDW extraction on the GPU (include/pfb_channelizer.h, pfb_pdw_extract).
//
// Restates the second half of /root/reference/matlab/create_pdws_channelized.m (lines 64-143) as
// data-parallel passes over the F x M channelizer output (frame-major complex64, fftshift-ed):
//
//   noise floor  :73    exact per-channel median of |y|.  A hashed 1-in-k row sample brackets the median
//                       (radix select of two sample ranks 5 sigma either side of the middle); ONE pass
//                       over the data counts what lies below the bracket and gathers what lies inside
//                       it (about 2 %), and a per-channel radix select over those candidates picks the
//                       exact order statistics.  The count proves the bracket held the median; if it
//                       did not (or the data are too tied / too short to sample) the full MSB-first
//                       radix select runs instead: 8-bit digit histograms in LDS (lane = channel, so
//                       LDS atomics never collide) until the bucket is small, then an exact finish.
//   threshold    :74-75 NF * 10^(SNR/10)
//   edges        :85-135 the leading/trailing-edge state machine is a 2-state automaton
//                       next = active ? (mag > thr) : (mag >= thr); each tile of frames is summarised
//                       as a 2-bit transition function, the functions are scanned per channel, and the
//                       tiles are replayed with their incoming state to count and emit edge indices.
//   per pulse    :98-132 one workgroup per pulse: medians of the magnitudes and of the wrapped phase
//                       steps by the same radix select (values cached in LDS when they fit).
//
// Arithmetic is float64 like the MATLAB script (the F x M input is promoted sample by sample).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "pfb_channelizer.h"

namespace {

constexpr int kTile = 512;        // frames per tile of the edge scan
constexpr int kCand = 2048;       // candidate capacity per channel for the exact median finish
constexpr int kPulseCache = 1024; // per-pulse values cached in LDS up to this many
constexpr int kSampleRows = 65536; // rows sampled to bracket the median (below 8x this the full select runs)
constexpr int kSamplePasses = 4;   // digits resolved on the sample: bracket edges to 2^-20 relative
constexpr int kStage = 64;         // LDS staging slots per channel and workgroup in the bracket pass
constexpr int kBracketRows = 1024; // rows per workgroup of the bracket pass
constexpr double kRadToDeg = 57.295779513082320876798154814105;

// |y|^2 of a complex64 is EXACT in float64 (two 48-bit products, 49-bit sum), so selecting on it is
// selecting on the true magnitude, and sqrt() of it is the correctly rounded magnitude.
__device__ __forceinline__ double mag2_of(float2 v) { return fma((double)v.x, (double)v.x, (double)v.y * (double)v.y); }
__device__ __forceinline__ double mag_of(float2 v) { return sqrt(mag2_of(v)); }
__device__ __forceinline__ double phase_deg(float2 v) { return atan2((double)v.y, (double)v.x) * kRadToDeg; }

// order-preserving 64-bit key of a finite double
__device__ __forceinline__ unsigned long long dkey(double d) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(d);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double dkey_inv(unsigned long long k) {
  const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)b);
}

// k-th smallest of n doubles produced by get(i), by 8 passes of 8-bit digits; the whole workgroup
// cooperates (every thread must call it, with the same n and k).  The digit holding rank k is found by
// wave 0: four counters per lane, a shuffle scan, one lane owns the answer.
template <class Get>
__device__ double block_select(Get get, long long n, long long k, unsigned* hist /* [256] shared */,
                               unsigned long long* pick /* [2] shared */) {
  unsigned long long prefix = 0ull;
  for (int pass = 0; pass < 8; ++pass) {
    const int shift = 56 - 8 * pass;
    for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0u;
    __syncthreads();
    for (long long i = threadIdx.x; i < n; i += blockDim.x) {
      const unsigned long long key = dkey(get(i));
      if (pass == 0 || (key >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&hist[(unsigned)(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 64) {
      const int l = threadIdx.x;
      const unsigned long long c0 = hist[4 * l], c1 = hist[4 * l + 1], c2 = hist[4 * l + 2], c3 = hist[4 * l + 3];
      const unsigned long long sum = c0 + c1 + c2 + c3;
      unsigned long long inc = sum;
      for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long prev = __shfl_up(inc, d);
        if (l >= d) inc += prev;
      }
      unsigned long long cum = inc - sum;
      const unsigned long long kk = (unsigned long long)k;
      if (cum <= kk && kk < inc) {  // exactly one lane: 0 <= k < n = total count
        int d = 4 * l;
        if (kk >= cum + c0) { cum += c0; ++d;
          if (kk >= cum + c1) { cum += c1; ++d;
            if (kk >= cum + c2) { cum += c2; ++d; } } }
        pick[0] = (unsigned long long)d;
        pick[1] = cum;
      }
    }
    __syncthreads();
    prefix |= pick[0] << shift;
    k -= (long long)pick[1];
    __syncthreads();
  }
  return dkey_inv(prefix);
}

template <class Get>
__device__ double block_median(Get get, long long n, unsigned* hist, unsigned long long* pick) {
  const double hi = block_select(get, n, n / 2, hist, pick);
  if (n & 1) return hi;
  return 0.5 * (block_select(get, n, n / 2 - 1, hist, pick) + hi);
}

// median of the n <= kPulseCache values in v[] (LDS) by rank counting: element i has rank
// #{v_j < v_i} + #{j < i : v_j == v_i}; the two middle ranks announce themselves.  No passes, two barriers.
__device__ double cached_median(const double* v, int n, double* mid /* [2] shared */) {
  const int kh = n / 2, kl = kh - 1;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const double vi = v[i];
    int rank = 0;
    for (int j = 0; j < n; ++j) {
      const double vj = v[j];
      rank += (vj < vi) || (vj == vi && j < i);
    }
    if (rank == kh) mid[1] = vi;
    if (rank == kl) mid[0] = vi;
  }
  __syncthreads();
  const double r = (n & 1) ? mid[1] : 0.5 * (mid[0] + mid[1]);
  __syncthreads();
  return r;
}

// ---------------------------------------------------------------------------------
// noise floor: radix select of rank[col] over the column's magnitudes

// q-th sampled row: one row out of every `stride`, at a hashed offset inside its stride block (a fixed
// offset could alias with a periodic signal)
__device__ __forceinline__ long long sample_row(long long q, long long stride) {
  if (stride == 1) return q;
  unsigned long long h = (unsigned long long)q * 0x9E3779B97F4A7C15ull;
  h ^= h >> 29;
  return q * stride + (long long)(((h >> 40) * (unsigned long long)stride) >> 24);
}

// one 8-bit digit histogram pass over F rows (row q -> sample_row(q, stride)).
// grid = (column groups of 64, row blocks); block = 256 (4 waves)
__global__ void __launch_bounds__(256) pdw_hist_kernel(const float2* y, long long F, long long stride, int M, int pass,
                                                       const unsigned long long* prefix, unsigned* hist) {
  __shared__ unsigned h[256][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 256 * 64; i += 256) (&h[0][0])[i] = 0u;
  __syncthreads();
  const int col = blockIdx.x * 64 + lane;
  const bool valid = col < M;
  const int shift = 56 - 8 * pass;
  const unsigned long long pre = valid ? prefix[col] : 0ull;
  const long long rows_per_block = (F + gridDim.y - 1) / gridDim.y;
  const long long r0 = (long long)blockIdx.y * rows_per_block;
  const long long r1 = (r0 + rows_per_block < F) ? r0 + rows_per_block : F;
  if (valid) {
    for (long long r = r0 + wave; r < r1; r += 4) {
      const unsigned long long k = dkey(mag2_of(y[sample_row(r, stride) * M + col]));  // ordered like the magnitude, no sqrt
      const bool in_bucket = (pass == 0) || ((k >> (shift + 8)) == (pre >> (shift + 8)));
      if (in_bucket) atomicAdd(&h[(unsigned)(k >> shift) & 255u][lane], 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 256 * 64; i += 256) {
    const int d = i >> 6, c = i & 63;
    const unsigned v = h[d][c];
    if (v && blockIdx.x * 64 + c < M) atomicAdd(&hist[(size_t)(blockIdx.x * 64 + c) * 256 + d], v);
  }
}

// choose the digit holding rank[col]; one thread per column
__global__ void pdw_pick_kernel(int M, int pass, unsigned* hist, unsigned long long* prefix,
                                unsigned long long* rank, unsigned* bucket, unsigned long long* below) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= M) return;
  unsigned* hc = hist + (size_t)col * 256;
  unsigned long long r = rank[col], cum = 0;
  int d = 0;
  for (; d < 255; ++d) {
    if (cum + hc[d] > r) break;
    cum += hc[d];
  }
  prefix[col] |= (unsigned long long)d << (56 - 8 * pass);
  rank[col] = r - cum;        // rank inside the chosen bucket
  below[col] += cum;          // elements strictly below the bucket so far
  bucket[col] = hc[d];
  for (int i = 0; i < 256; ++i) hc[i] = 0u;
}

// gather the bucket's exact values, and the largest value below the bucket (for the lower median)
__global__ void __launch_bounds__(256) pdw_collect_kernel(const float2* y, long long F, int M, int passes_done,
                                                          const unsigned long long* prefix, double* cand,
                                                          unsigned* cand_n, unsigned long long* max_below) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  if (col >= M) return;
  const int low_bits = 64 - 8 * passes_done;  // undecided low bits
  const unsigned long long pre = prefix[col];
  const long long rows_per_block = (F + gridDim.y - 1) / gridDim.y;
  const long long r0 = (long long)blockIdx.y * rows_per_block;
  const long long r1 = (r0 + rows_per_block < F) ? r0 + rows_per_block : F;
  unsigned long long best = 0ull;
  for (long long r = r0 + wave; r < r1; r += 4) {
    const double m = mag2_of(y[r * M + col]);
    const unsigned long long k = dkey(m);
    const unsigned long long hi = (low_bits >= 64) ? 0ull : (k >> low_bits), phi = (low_bits >= 64) ? 0ull : (pre >> low_bits);
    if (hi == phi) {
      const unsigned slot = atomicAdd(&cand_n[col], 1u);
      if (slot < (unsigned)kCand) cand[(size_t)col * kCand + slot] = m;
    } else if (hi < phi) {
      best = k > best ? k : best;
    }
  }
  if (best) atomicMax(&max_below[col], best);
}

// exact finish: sort the candidates of one column (bitonic in LDS), pick the two middle values
__global__ void __launch_bounds__(256) pdw_median_finish_kernel(long long F, int passes_done, const double* cand,
                                                                const unsigned* cand_n, const unsigned long long* prefix,
                                                                const unsigned long long* rank,
                                                                const unsigned long long* max_below, double* nf) {
  __shared__ double v[kCand];
  const int col = blockIdx.x;
  const unsigned n = cand_n[col];
  const unsigned long long r = rank[col];
  double v1, v0;
  if (n > (unsigned)kCand) {
    // only reachable when all 64 key bits are decided: the whole bucket is one value
    v1 = dkey_inv(prefix[col]);
    v0 = (r > 0) ? v1 : dkey_inv(max_below[col]);
  } else {
    for (int i = threadIdx.x; i < kCand; i += 256) v[i] = (i < (int)n) ? cand[(size_t)col * kCand + i] : INFINITY;
    __syncthreads();
    for (int k = 2; k <= kCand; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = threadIdx.x; i < kCand; i += 256) {
          const int l = i ^ j;
          if (l > i) {
            const bool up = (i & k) == 0;
            const double a = v[i], b = v[l];
            if ((a > b) == up) { v[i] = b; v[l] = a; }
          }
        }
        __syncthreads();
      }
    v1 = v[r];
    v0 = (r > 0) ? v[r - 1] : dkey_inv(max_below[col]);
  }
  (void)passes_done;
  // the candidates are squared magnitudes; MATLAB median: mean of the two middle values
  if (threadIdx.x == 0) nf[col] = (F & 1) ? sqrt(v1) : 0.5 * (sqrt(v0) + sqrt(v1));
}

// ---- sampled bracket path -------------------------------------------------------------------------

// One pass over the data with the bracket [lo, hi] of every channel (key prefixes from the sample, low
// bits cleared / set): count and track the maximum of what lies below, gather what lies inside.
// Candidates are staged per workgroup in LDS (lane = channel) and flushed as contiguous runs, so the
// global append costs one atomic per channel and workgroup; a full stage spills element by element.
__global__ void __launch_bounds__(256) pdw_bracket_kernel(const float2* y, long long F, int M,
                                                          const unsigned long long* pre_lo, const unsigned long long* pre_hi,
                                                          double* cand, unsigned cap, unsigned* cand_n,
                                                          unsigned long long* below, unsigned long long* max_below,
                                                          unsigned* flags) {
  __shared__ double stage[kStage][64];
  __shared__ unsigned cnt[64], base[64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x < 64) cnt[threadIdx.x] = 0u;
  __syncthreads();
  const int col = blockIdx.x * 64 + lane;
  const long long r0 = (long long)blockIdx.y * kBracketRows;
  const long long r1 = (r0 + kBracketRows < F) ? r0 + kBracketRows : F;
  constexpr unsigned long long kLow = (1ull << (64 - 8 * kSamplePasses)) - 1ull;
  if (col < M) {
    const unsigned long long lo = pre_lo[col] & ~kLow, hi = pre_hi[col] | kLow;
    unsigned long long nb = 0ull, best = 0ull;
    auto visit = [&](float2 v) {
      const double m = mag2_of(v);
      const unsigned long long k = dkey(m);
      if (k < lo) {
        ++nb;
        best = k > best ? k : best;
      } else if (k <= hi) {
        const unsigned slot = atomicAdd(&cnt[lane], 1u);
        if (slot < (unsigned)kStage) {
          stage[slot][lane] = m;
        } else {
          const unsigned g = atomicAdd(&cand_n[col], 1u);
          if (g < cap) cand[(size_t)col * cap + g] = m;
          else atomicOr(flags, 1u);
        }
      }
    };
    long long r = r0 + wave;
    for (; r + 12 < r1; r += 16) {  // four rows in flight per lane
      const float2 a = y[r * M + col], b = y[(r + 4) * M + col], c = y[(r + 8) * M + col], d = y[(r + 12) * M + col];
      visit(a); visit(b); visit(c); visit(d);
    }
    for (; r < r1; r += 4) visit(y[r * M + col]);
    if (nb) atomicAdd(&below[col], nb);
    if (best) atomicMax(&max_below[col], best);
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    const unsigned n = cnt[threadIdx.x] < (unsigned)kStage ? cnt[threadIdx.x] : (unsigned)kStage;
    base[threadIdx.x] = (c < M && n) ? atomicAdd(&cand_n[c], n) : 0u;
  }
  __syncthreads();
  for (int c = wave; c < 64; c += 4) {  // one channel per wave at a time: lanes = slots, contiguous stores
    const int gc = blockIdx.x * 64 + c;
    if (gc >= M) break;
    const unsigned n = cnt[c] < (unsigned)kStage ? cnt[c] : (unsigned)kStage, b = base[c];
    for (unsigned sl = lane; sl < n; sl += 64) {
      if (b + sl < cap) cand[(size_t)gc * cap + b + sl] = stage[sl][c];
      else atomicOr(flags, 1u);
    }
  }
}

// exact order statistics among the gathered candidates; one workgroup per channel.  The median's rank
// must fall inside the candidate set -- that is the proof the sampled bracket held it.
__global__ void __launch_bounds__(1024) pdw_bracket_finish_kernel(long long F, const double* cand, unsigned cap,
                                                                  const unsigned* cand_n, const unsigned long long* below,
                                                                  const unsigned long long* max_below, double* nf,
                                                                  unsigned* flags) {
  __shared__ unsigned hist[256];
  __shared__ unsigned long long pick[2];
  const int col = blockIdx.x;
  const unsigned long long n = cand_n[col], b = below[col], target = (unsigned long long)(F / 2);
  if (n > cap || b > target || target - b >= n) {  // uniform over the workgroup
    if (threadIdx.x == 0) { atomicOr(flags, 2u); nf[col] = 0.0; }
    return;
  }
  const long long r = (long long)(target - b);
  const double* v = cand + (size_t)col * cap;
  auto get = [&](long long i) { return v[i]; };
  const double v1 = block_select(get, (long long)n, r, hist, pick);
  double res;
  if (F & 1) {
    res = sqrt(v1);
  } else {
    const double v0 = (r > 0) ? block_select(get, (long long)n, r - 1, hist, pick) : dkey_inv(max_below[col]);
    res = 0.5 * (sqrt(v0) + sqrt(v1));
  }
  if (threadIdx.x == 0) nf[col] = res;
}

// ---------------------------------------------------------------------------------
// edges

// automaton step of create_pdws_channelized.m:85-135: inactive -> active on mag >= thr (:87),
// active -> inactive on mag <= thr (:94)
__device__ __forceinline__ int step_state(int active, double m, double thr) {
  return active ? (m > thr) : (m >= thr);
}

// tile summaries for BOTH incoming states: fn[tile][col] = f(0) | f(1) << 1 and the edge counts of either
// trajectory, cnt[tile][col] = (starts from 0, ends from 0, starts from 1, ends from 1).  With the counts
// of both trajectories in hand no second counting pass over the data is needed once the scan has told
// which state each tile really starts in.  grid = (column groups, tile groups of 4), one tile per wave
__global__ void __launch_bounds__(256) pdw_tilefn_kernel(const float2* y, long long F, int M, const double* thr,
                                                         unsigned char* fn, ushort4* cnt, long long ntiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  const long long tile = (long long)blockIdx.y * 4 + wave;
  if (col >= M || tile >= ntiles) return;
  const double t = thr[col];
  const long long r0 = tile * kTile, r1 = (r0 + kTile < F) ? r0 + kTile : F;
  int s0 = 0, s1 = 1;
  unsigned a0 = 0, e0 = 0, a1 = 0, e1 = 0;
  for (long long r = r0; r < r1; ++r) {
    const double m = mag_of(y[r * M + col]);
    const int n0 = step_state(s0, m, t), n1 = step_state(s1, m, t);
    a0 += (unsigned)((s0 ^ 1) & n0); e0 += (unsigned)(s0 & (n0 ^ 1));
    a1 += (unsigned)((s1 ^ 1) & n1); e1 += (unsigned)(s1 & (n1 ^ 1));
    s0 = n0; s1 = n1;
  }
  fn[tile * M + col] = (unsigned char)(s0 | (s1 << 1));
  cnt[tile * M + col] = make_ushort4((unsigned short)a0, (unsigned short)e0, (unsigned short)a1, (unsigned short)e1);
}

// per column: incoming state of every tile, then the exclusive prefix of the edge counts of the
// trajectory each tile really follows, and the column totals.  One wave per column: each lane owns a
// contiguous segment of tiles; transition functions, then counts, are scanned across the wave (lane
// order = time order) and each lane replays its segment.
__global__ void __launch_bounds__(64) pdw_tilescan_kernel(int M, long long ntiles, const unsigned char* fn,
                                                          const ushort4* cnt, unsigned char* state_in,
                                                          unsigned long long* off_s, unsigned long long* off_e,
                                                          unsigned long long* tot_s, unsigned long long* tot_e) {
  const int col = blockIdx.x, lane = threadIdx.x;
  const long long per = (ntiles + 63) / 64;
  const long long t0 = lane * per, t1 = (t0 + per < ntiles) ? t0 + per : ntiles;
  int f = 0x2;  // identity: f(0)=0, f(1)=1  -> bits (f0 | f1<<1) = 0b10
  for (long long t = t0; t < t1; ++t) {
    const int g = fn[t * M + col];  // apply g after f: h(s) = g(f(s))
    f = ((g >> (f & 1)) & 1) | (((g >> ((f >> 1) & 1)) & 1) << 1);
  }
  // inclusive scan of function composition across lanes
  int inc = f;
  for (int d = 1; d < 64; d <<= 1) {
    const int prev = __shfl_up(inc, d);
    if (lane >= d) inc = ((inc >> (prev & 1)) & 1) | (((inc >> ((prev >> 1) & 1)) & 1) << 1);
  }
  int exc = __shfl_up(inc, 1);
  if (lane == 0) exc = 0x2;
  const int s_in = exc & 1;  // state entering my segment when the stream starts inactive: exc(0)
  int s = s_in;
  unsigned long long a = 0, b = 0;
  for (long long t = t0; t < t1; ++t) {
    state_in[t * M + col] = (unsigned char)s;
    const ushort4 c = cnt[t * M + col];
    a += s ? c.z : c.x;
    b += s ? c.w : c.y;
    s = (fn[t * M + col] >> s) & 1;
  }
  unsigned long long ia = a, ib = b;
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned long long pa = __shfl_up(ia, d), pb = __shfl_up(ib, d);
    if (lane >= d) { ia += pa; ib += pb; }
  }
  unsigned long long ea = ia - a, eb = ib - b;  // exclusive
  s = s_in;
  for (long long t = t0; t < t1; ++t) {
    off_s[t * M + col] = ea; off_e[t * M + col] = eb;
    const ushort4 c = cnt[t * M + col];
    ea += s ? c.z : c.x;
    eb += s ? c.w : c.y;
    s = (fn[t * M + col] >> s) & 1;
  }
  if (lane == 63) { tot_s[col] = ia; tot_e[col] = ib; }
}

// replay a tile from its incoming state and write the leading / trailing edge frame indices
__global__ void __launch_bounds__(256) pdw_edges_kernel(const float2* y, long long F, int M, const double* thr,
                                                        const unsigned char* state_in, long long ntiles,
                                                        const unsigned long long* off_s, const unsigned long long* off_e,
                                                        long long* starts, long long* ends) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  const long long tile = (long long)blockIdx.y * 4 + wave;
  if (col >= M || tile >= ntiles) return;
  const double t = thr[col];
  const long long r0 = tile * kTile, r1 = (r0 + kTile < F) ? r0 + kTile : F;
  int s = state_in[tile * M + col];
  unsigned long long os = off_s[tile * M + col], oe = off_e[tile * M + col];
  for (long long r = r0; r < r1; ++r) {
    const double m = mag_of(y[r * M + col]);
    const int n = step_state(s, m, t);
    if (n != s) {
      if (n) starts[os++] = r;
      else ends[oe++] = r;
    }
    s = n;
  }
}

// make the per-tile offsets absolute: add the column bases (columns outermost = the reference's order)
__global__ void pdw_rebase_kernel(int M, long long ntiles, unsigned long long* off_s, unsigned long long* off_e,
                                  const unsigned long long* base_s, const unsigned long long* base_e) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ntiles * M) return;
  const int col = (int)(i % M);
  off_s[i] += base_s[col];
  off_e[i] += base_e[col];
}

// ---------------------------------------------------------------------------------
// per pulse

__global__ void __launch_bounds__(256) pdw_pulse_kernel(const float2* y, int M, const long long* starts,
                                                        const long long* ends, const unsigned long long* base_s,
                                                        const unsigned long long* base_e, const double* nf, const double* bin_freqs, double fs, double fc,
                                                        double t0, unsigned flags, pfb_pdw* out, unsigned long long capacity) {
  __shared__ unsigned hist[256];
  __shared__ unsigned long long pick[2];
  __shared__ double cache[kPulseCache];
  __shared__ double mid[2];
  __shared__ int sat_flag;
  const unsigned long long pid = blockIdx.x;
  if (pid >= capacity) return;
  // channel of this pulse: base_e is the exclusive prefix of tot_e over channels
  int lo = 0, hi = M - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (base_e[mid] <= pid) lo = mid; else hi = mid - 1;
  }
  // (the largest channel whose base <= pid is the pulse's channel: every later base is > pid)
  const int b = lo;
  const unsigned long long k = pid - base_e[b];
  const long long toa = starts[base_s[b] + k], jj = ends[base_e[b] + k];
  const long long n = jj - toa + 1;
  const int pcol = (flags & PFB_PDW_MATLAB_QUIRKS) ? 0 : b;  // :114 phase(toa:jj) linear-indexes column 1
  if (threadIdx.x == 0) sat_flag = 0;
  __syncthreads();

  // :130-132 saturation: samples strictly inside the pulse (the edge samples take the other branches)
  int sat = 0;
  for (long long i = toa + 1 + threadIdx.x; i < jj; i += blockDim.x) {
    const float2 v = y[i * M + b];
    sat |= (fabs((double)v.x) >= 0.9999) || (fabs((double)v.y) >= 0.9999);
  }
  if (sat) atomicOr(&sat_flag, 1);

  // :101 amplitude = median magnitude over toa..jj
  double amp;
  if (n <= kPulseCache) {
    for (long long i = threadIdx.x; i < n; i += blockDim.x) cache[i] = mag_of(y[(toa + i) * M + b]);
    __syncthreads();
    amp = cached_median(cache, (int)n, mid);
  } else {
    amp = block_median([&](long long i) { return mag_of(y[(toa + i) * M + b]); }, n, hist, pick);
  }
  __syncthreads();

  // :114-117 median of the wrapped phase steps (degrees)
  auto dphi = [&](long long i) {
    double d = phase_deg(y[(toa + i + 1) * M + pcol]) - phase_deg(y[(toa + i) * M + pcol]);
    if (d < -180.0) d += 360.0;
    if (d > 180.0) d -= 360.0;
    return d;
  };
  double med;
  if (n - 1 <= kPulseCache) {
    for (long long i = threadIdx.x; i < n - 1; i += blockDim.x) cache[i] = dphi(i);
    __syncthreads();
    med = cached_median(cache, (int)(n - 1), mid);
  } else {
    med = block_median(dphi, n - 1, hist, pick);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    pfb_pdw o;
    o.toa = ((double)(toa + 1) / fs) + t0;            // :98 (1-based index)
    o.snr = 10.0 * log10(amp / nf[b]);                // :105
    o.pw = (double)(jj - toa) / fs;                   // :110
    // :80 indexes the UNSHIFTED centre-frequency list with the shifted column
    const double fbin = (flags & PFB_PDW_MATLAB_QUIRKS) ? bin_freqs[b] : bin_freqs[(b + (M + 1) / 2) % M];
    o.freq = (fc + fbin) + (fs / (360.0 / med));      // :122
    o.sat = sat_flag;
    o.bin = b;
    out[pid] = o;
  }
}

// ---------------------------------------------------------------------------------

thread_local std::string g_pdw_detail;
thread_local int g_pdw_path = 0;

#define PDW_TRY(expr)                                                                  \
  do {                                                                                 \
    const hipError_t e__ = (expr);                                                     \
    if (e__ != hipSuccess) {                                                           \
      g_pdw_detail = std::string(#expr) + ": " + hipGetErrorString(e__);               \
      (void)hipGetLastError();                                                         \
      rc = (e__ == hipErrorOutOfMemory) ? PFB_ERR_NO_MEMORY : PFB_ERR_HIP;             \
      goto done;                                                                       \
    }                                                                                  \
  } while (0)

// Device scratch is kept between calls (grow-only, one pair of arenas per device): a call needs some
// twenty buffers, and allocating and freeing them cost more than the kernels of a short extraction.
// Arena 0 holds everything sized by (F, M); arena 1 the edge lists and PDWs, sized by the pulse count.
constexpr int kMaxDevices = 64;
struct Arena {
  char* p = nullptr;
  size_t cap = 0, used = 0;
};
std::mutex g_ws_mutex;
Arena g_ws[kMaxDevices][2];

hipError_t arena_reserve(Arena& a, size_t bytes) {
  a.used = 0;
  if (bytes <= a.cap) return hipSuccess;
  (void)hipFree(a.p);
  a.p = nullptr;
  a.cap = 0;
  bytes += bytes / 8;
  const hipError_t e = hipMalloc((void**)&a.p, bytes);
  if (e == hipSuccess) a.cap = bytes;
  return e;
}
constexpr size_t kAlign = 256;
size_t padded(size_t bytes) { return (bytes + kAlign - 1) / kAlign * kAlign; }
template <class T>
T* take(Arena& a, size_t count) {
  T* r = reinterpret_cast<T*>(a.p + a.used);
  a.used += padded(count * sizeof(T));
  return r;
}

}  // namespace

extern "C" const char* pfb_pdw_last_error_detail(void) { return g_pdw_detail.c_str(); }

extern "C" int pfb_pdw_last_noise_floor_path(void) { return g_pdw_path; }

extern "C" int pfb_pdw_release_workspace(int32_t device_id) {
  std::lock_guard<std::mutex> lock(g_ws_mutex);
  int prev = -1;
  (void)hipGetDevice(&prev);
  for (int d = 0; d < kMaxDevices; ++d) {
    if (device_id >= 0 && d != device_id) continue;
    for (Arena& a : g_ws[d]) {
      if (!a.p) continue;
      (void)hipSetDevice(d);
      (void)hipFree(a.p);
      a = Arena{};
    }
  }
  if (prev >= 0) (void)hipSetDevice(prev);
  (void)hipGetLastError();
  return PFB_OK;
}

extern "C" int pfb_pdw_extract(const void* y_in, uint64_t frames, uint32_t M, uint32_t decimation, double fs_in,
                               double fc, double sample_start_time, double snr_threshold_db, uint32_t flags,
                               pfb_pdw* out, uint64_t capacity, uint64_t* count, double* noise_floor_out, uint32_t mem,
                               int32_t device_id, void* hip_stream) {
  if (!y_in || !count || M < 1 || decimation < 1 || frames < 1 || mem > PFB_MEM_DEVICE || (capacity && !out))
    return PFB_ERR_BAD_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    (void)hipGetLastError();
    return PFB_ERR_NO_DEVICE;
  }
  int prev_dev = -1;
  (void)hipGetDevice(&prev_dev);
  if (device_id >= 0 && device_id != prev_dev) {
    if (device_id >= ndev || hipSetDevice(device_id) != hipSuccess) return PFB_ERR_BAD_ARG;
  }
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= kMaxDevices) return PFB_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lock(g_ws_mutex);  // one extraction per process at a time shares the scratch
  Arena& ws = g_ws[dev][0];
  Arena& ws2 = g_ws[dev][1];

  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  const long long F = (long long)frames;
  const int Mi = (int)M;
  const long long ntiles = (F + kTile - 1) / kTile;
  const size_t tm = (size_t)ntiles * M;
  const int cgroups = (Mi + 63) / 64;
  const double fs = fs_in / (double)decimation;  // :62
  int rc = PFB_OK;

  // sampled bracket: worth it once the data are several times the sample
  const bool sampled = F >= 8ll * kSampleRows;
  const long long stride = sampled ? F / kSampleRows : 1;
  const long long ns = F / stride;                                  // sampled rows
  const long long delta = (long long)std::ceil(2.5 * std::sqrt((double)ns)) + 2;  // 5 sigma of the median's sample rank
  const size_t expect = (size_t)((double)(2 * delta + 1) / (double)ns * (double)F);
  const unsigned cap = sampled ? (unsigned)std::min<size_t>((size_t)F, 2 * expect + 4096) : 0u;

  const float2* d_y = nullptr;
  unsigned *d_hist, *d_bucket, *d_cand_n, *d_flags;
  unsigned long long *d_prefix, *d_prefix_hi, *d_rank, *d_below, *d_maxbelow, *d_off_s, *d_off_e, *d_tot, *d_base;
  double *d_cand, *d_nf, *d_thr, *d_binf;
  unsigned char *d_fn, *d_state;
  ushort4* d_cnt;
  long long *d_starts = nullptr, *d_ends = nullptr;
  pfb_pdw* d_out = nullptr;
  std::vector<unsigned> h_bucket(M);
  std::vector<unsigned long long> h_rank(M), h_tot(2 * (size_t)M), h_base(2 * (size_t)M);
  std::vector<double> h_nf(M), h_thr(M), h_binf(M);
  unsigned long long total_s = 0, total_e = 0;
  unsigned h_flags = 0;
  bool have_nf = false;
  int passes = 0;
  const int row_blocks = (int)std::min<long long>(1024, std::max<long long>(1, F / 256));
  const size_t cand_elems = std::max<size_t>((size_t)M * kCand, (size_t)M * cap);

  {
    size_t need = 0;
    if (mem == PFB_MEM_HOST) need += padded((size_t)F * M * sizeof(float2));
    need += padded((size_t)M * 256 * sizeof(unsigned)) + 3 * padded(M * sizeof(unsigned)) + padded(sizeof(unsigned));
    need += 5 * padded(M * sizeof(unsigned long long)) + 2 * padded(2 * (size_t)M * sizeof(unsigned long long));
    need += padded(cand_elems * sizeof(double)) + 3 * padded(M * sizeof(double));
    need += 2 * padded(tm) + padded(tm * sizeof(ushort4)) + 2 * padded(tm * sizeof(unsigned long long));
    PDW_TRY(arena_reserve(ws, need));
  }
  if (mem == PFB_MEM_HOST) {
    float2* own = take<float2>(ws, (size_t)F * M);
    PDW_TRY(hipMemcpyAsync(own, y_in, (size_t)F * M * sizeof(float2), hipMemcpyHostToDevice, st));
    d_y = own;
  } else {
    d_y = static_cast<const float2*>(y_in);
  }
  d_hist = take<unsigned>(ws, (size_t)M * 256);
  d_bucket = take<unsigned>(ws, M);
  d_cand_n = take<unsigned>(ws, M);
  d_flags = take<unsigned>(ws, 1);
  d_prefix = take<unsigned long long>(ws, M);
  d_prefix_hi = take<unsigned long long>(ws, M);
  d_rank = take<unsigned long long>(ws, M);
  d_below = take<unsigned long long>(ws, M);
  d_maxbelow = take<unsigned long long>(ws, M);
  d_tot = take<unsigned long long>(ws, 2 * (size_t)M);
  d_base = take<unsigned long long>(ws, 2 * (size_t)M);
  d_cand = take<double>(ws, cand_elems);
  d_nf = take<double>(ws, M);
  d_thr = take<double>(ws, M);
  d_binf = take<double>(ws, M);
  d_fn = take<unsigned char>(ws, tm);
  d_state = take<unsigned char>(ws, tm);
  d_cnt = take<ushort4>(ws, tm);
  d_off_s = take<unsigned long long>(ws, tm);
  d_off_e = take<unsigned long long>(ws, tm);
  (void)take<unsigned>(ws, M);  // spare

  PDW_TRY(hipMemsetAsync(d_hist, 0, (size_t)M * 256 * sizeof(unsigned), st));

  // ---- noise floor (:73), sampled bracket first
  if (sampled) {
    const int sblocks = (int)std::min<long long>(1024, std::max<long long>(1, ns / 256));
    for (int side = 0; side < 2; ++side) {  // radix select of the two bracket ranks on the sample
      const long long k = side == 0 ? std::max<long long>(0, ns / 2 - delta) : std::min<long long>(ns - 1, ns / 2 + delta);
      unsigned long long* pre = side == 0 ? d_prefix : d_prefix_hi;
      std::fill(h_rank.begin(), h_rank.end(), (unsigned long long)k);
      PDW_TRY(hipMemsetAsync(pre, 0, M * sizeof(unsigned long long), st));
      PDW_TRY(hipMemcpyAsync(d_rank, h_rank.data(), M * sizeof(unsigned long long), hipMemcpyHostToDevice, st));
      PDW_TRY(hipStreamSynchronize(st));  // h_rank is reused by the next side
      for (int ps = 0; ps < kSamplePasses; ++ps) {
        hipLaunchKernelGGL(pdw_hist_kernel, dim3(cgroups, sblocks), dim3(256), 0, st, d_y, ns, stride, Mi, ps, pre, d_hist);
        hipLaunchKernelGGL(pdw_pick_kernel, dim3((Mi + 63) / 64), dim3(64), 0, st, Mi, ps, d_hist, pre, d_rank, d_bucket,
                           d_below);
      }
    }
    PDW_TRY(hipMemsetAsync(d_below, 0, M * sizeof(unsigned long long), st));
    PDW_TRY(hipMemsetAsync(d_maxbelow, 0, M * sizeof(unsigned long long), st));
    PDW_TRY(hipMemsetAsync(d_cand_n, 0, M * sizeof(unsigned), st));
    PDW_TRY(hipMemsetAsync(d_flags, 0, sizeof(unsigned), st));
    hipLaunchKernelGGL(pdw_bracket_kernel, dim3(cgroups, (unsigned)((F + kBracketRows - 1) / kBracketRows)), dim3(256), 0, st,
                       d_y, F, Mi, d_prefix, d_prefix_hi, d_cand, cap, d_cand_n, d_below, d_maxbelow, d_flags);
    hipLaunchKernelGGL(pdw_bracket_finish_kernel, dim3(Mi), dim3(1024), 0, st, F, d_cand, cap, d_cand_n, d_below, d_maxbelow,
                       d_nf, d_flags);
    PDW_TRY(hipGetLastError());
    PDW_TRY(hipMemcpyAsync(&h_flags, d_flags, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    PDW_TRY(hipMemcpyAsync(h_nf.data(), d_nf, M * sizeof(double), hipMemcpyDeviceToHost, st));
    PDW_TRY(hipStreamSynchronize(st));
    have_nf = (h_flags == 0);
  }
  g_pdw_path = have_nf ? 1 : (sampled ? 3 : 2);
  if (!have_nf) {  // full radix select of rank F/2, then the exact finish
    std::fill(h_rank.begin(), h_rank.end(), (unsigned long long)(F / 2));
    PDW_TRY(hipMemsetAsync(d_hist, 0, (size_t)M * 256 * sizeof(unsigned), st));
    PDW_TRY(hipMemsetAsync(d_prefix, 0, M * sizeof(unsigned long long), st));
    PDW_TRY(hipMemsetAsync(d_below, 0, M * sizeof(unsigned long long), st));
    PDW_TRY(hipMemsetAsync(d_maxbelow, 0, M * sizeof(unsigned long long), st));
    PDW_TRY(hipMemsetAsync(d_cand_n, 0, M * sizeof(unsigned), st));
    PDW_TRY(hipMemcpyAsync(d_rank, h_rank.data(), M * sizeof(unsigned long long), hipMemcpyHostToDevice, st));
    for (passes = 0; passes < 8;) {
      hipLaunchKernelGGL(pdw_hist_kernel, dim3(cgroups, row_blocks), dim3(256), 0, st, d_y, F, 1ll, Mi, passes, d_prefix,
                         d_hist);
      hipLaunchKernelGGL(pdw_pick_kernel, dim3((Mi + 63) / 64), dim3(64), 0, st, Mi, passes, d_hist, d_prefix, d_rank,
                         d_bucket, d_below);
      ++passes;
      PDW_TRY(hipMemcpyAsync(h_bucket.data(), d_bucket, M * sizeof(unsigned), hipMemcpyDeviceToHost, st));
      PDW_TRY(hipStreamSynchronize(st));
      if (*std::max_element(h_bucket.begin(), h_bucket.end()) <= (unsigned)kCand) break;
    }
    hipLaunchKernelGGL(pdw_collect_kernel, dim3(cgroups, row_blocks), dim3(256), 0, st, d_y, F, Mi, passes, d_prefix, d_cand,
                       d_cand_n, d_maxbelow);
    hipLaunchKernelGGL(pdw_median_finish_kernel, dim3(Mi), dim3(256), 0, st, F, passes, d_cand, d_cand_n, d_prefix, d_rank,
                       d_maxbelow, d_nf);
    PDW_TRY(hipGetLastError());
    PDW_TRY(hipMemcpyAsync(h_nf.data(), d_nf, M * sizeof(double), hipMemcpyDeviceToHost, st));
    PDW_TRY(hipStreamSynchronize(st));
  }
  {
    const double gain = std::pow(10.0, snr_threshold_db / 10.0);  // :74-75 (dB applied to magnitude with /10)
    for (uint32_t b = 0; b < M; ++b) h_thr[b] = h_nf[b] * gain;
    pfb_center_frequencies(M, fs_in, h_binf.data());              // :42, before fs is decimated
    if (noise_floor_out) std::memcpy(noise_floor_out, h_nf.data(), M * sizeof(double));
  }
  PDW_TRY(hipMemcpyAsync(d_thr, h_thr.data(), M * sizeof(double), hipMemcpyHostToDevice, st));
  PDW_TRY(hipMemcpyAsync(d_binf, h_binf.data(), M * sizeof(double), hipMemcpyHostToDevice, st));

  // ---- edges (:85-135)
  {
    const dim3 tgrid(cgroups, (unsigned)((ntiles + 3) / 4));
    hipLaunchKernelGGL(pdw_tilefn_kernel, tgrid, dim3(256), 0, st, d_y, F, Mi, d_thr, d_fn, d_cnt, ntiles);
    hipLaunchKernelGGL(pdw_tilescan_kernel, dim3(Mi), dim3(64), 0, st, Mi, ntiles, d_fn, d_cnt, d_state, d_off_s, d_off_e, d_tot,
                       d_tot + M);
    PDW_TRY(hipGetLastError());
    PDW_TRY(hipMemcpyAsync(h_tot.data(), d_tot, 2 * (size_t)M * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    PDW_TRY(hipStreamSynchronize(st));
    for (uint32_t b = 0; b < M; ++b) {  // channels outermost, like the reference's for bin = 1:M
      h_base[b] = total_s; h_base[M + b] = total_e;
      total_s += h_tot[b]; total_e += h_tot[M + b];
    }
    *count = total_e;  // a pulse still active at the end of the data produces no PDW (:94 never fires)
    if (total_e > 0) {
      const unsigned long long n_out = std::min<unsigned long long>(total_e, capacity);
      PDW_TRY(arena_reserve(ws2, padded((size_t)total_s * sizeof(long long)) + padded((size_t)total_e * sizeof(long long)) +
                                     padded((size_t)n_out * sizeof(pfb_pdw)) + kAlign));
      d_starts = take<long long>(ws2, (size_t)total_s);
      d_ends = take<long long>(ws2, (size_t)total_e);
      d_out = take<pfb_pdw>(ws2, (size_t)n_out);
      PDW_TRY(hipMemcpyAsync(d_base, h_base.data(), 2 * (size_t)M * sizeof(unsigned long long), hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(pdw_rebase_kernel, dim3((unsigned)((tm + 255) / 256)), dim3(256), 0, st, Mi, ntiles, d_off_s, d_off_e,
                         d_base, d_base + M);
      hipLaunchKernelGGL(pdw_edges_kernel, tgrid, dim3(256), 0, st, d_y, F, Mi, d_thr, d_state, ntiles,
                         (const unsigned long long*)d_off_s, (const unsigned long long*)d_off_e, d_starts, d_ends);
      if (n_out > 0) {
        hipLaunchKernelGGL(pdw_pulse_kernel, dim3((unsigned)n_out), dim3(256), 0, st, d_y, Mi, d_starts, d_ends, d_base,
                           d_base + M, d_nf, d_binf, fs, fc, sample_start_time, flags, d_out, n_out);
        PDW_TRY(hipGetLastError());
        PDW_TRY(hipMemcpyAsync(out, d_out, (size_t)n_out * sizeof(pfb_pdw), hipMemcpyDeviceToHost, st));
      }
      PDW_TRY(hipStreamSynchronize(st));
    }
  }

done:
  (void)hipStreamSynchronize(st);
  if (device_id >= 0 && device_id != prev_dev && prev_dev >= 0) (void)hipSetDevice(prev_dev);
  return rc;
}
