// pfb_pdw.hip -- channelized PDW extraction on the GPU (include/pfb_channelizer.h, pfb_pdw_extract).
//
// Restates the second half of /root/reference/matlab/create_pdws_channelized.m (lines 64-143) as
// data-parallel passes over the F x M channelizer output (frame-major complex64, fftshift-ed):
//
//   noise floor  :73    exact per-channel median of |y|.  A hashed 1-in-k row sample, read once, brackets the
//                       median (per-channel select of two sample ranks 5 sigma either side of the middle); ONE
//                       pass over the data -- every sample screened in float32, only the bracket's zone promoted
//                       to float64 -- counts what lies below the bracket and gathers what lies inside it (about
//                       2 %), and a per-channel select over those candidates picks the exact order statistics.
//                       The count proves the bracket held the median; if it did not (or the data are too tied /
//                       too short to sample) the full MSB-first radix select runs instead: 8-bit digit histograms
//                       in LDS (lane = channel, so LDS atomics never collide) until the bucket is small, then an
//                       exact finish.  The same pass leaves the comparison masks of the edge stage.
//   threshold    :74-75 NF * 10^(SNR/10)
//   edges        :85-135 the leading/trailing-edge state machine is a 2-state automaton
//                       next = active ? (mag > thr) : (mag >= thr); each tile of frames is summarised
//                       as a 2-bit transition function, the functions are scanned per channel, and the
//                       tiles are replayed with their incoming state to count and emit edge indices.
//   per pulse    :98-132 one workgroup per pulse: medians of the magnitudes and of the wrapped phase
//                       steps by rank counting or a three-scan bucket select (values cached in LDS when they fit).
//
// The raw-stream script (matlab/create_pdws.m:30-105, pfb_pdw_extract_raw) shares the edge and pulse stages; its
// one column makes the noise floor a time-parallel radix select whose leading digits are predicted from a small
// sample and proven by the first counting pass, and its masks a comparison of integer keys.
//
// Arithmetic is float64 like the MATLAB scripts: everything that decides an outcome is computed on the exact float64
// |y|^2 (float32 only screens what cannot matter).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "pfb_common.h"  // abi_guard, launch_transpose_slab (pfb_kernels.hip)

namespace {

constexpr int kTile = 512;        // smallest tile of the edge scan, in frames (tiles grow with the stream, see tile_words_for)
constexpr int kCand = 2048;       // candidate capacity per channel for the exact median finish
constexpr int kPulseCache = 512;  // per-pulse values cached in LDS up to this many (channelized: pulses are tens of frames)
constexpr int kPulseCacheRaw = 7168; // same for the raw stream, whose pulses are thousands of samples (56 KB of LDS)
constexpr int kCountingMedian = 512; // cached pulses up to this long take the O(n^2 / threads) counting median
constexpr int kSampleRows = 65536; // rows sampled to bracket the median (below 8x this the full select runs)
constexpr int kSamplePasses = 3;   // digits resolved on the sample: bracket edges to 2^-12 relative
constexpr int kUndecided = 1 << 20; // samples too close to the threshold's bracket to classify before the median is known
constexpr int kBracketRows = 1024; // rows per workgroup of the bracket pass
constexpr int kBracketInFlight = 16; // rows each lane of the bracket pass has in flight
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

// histogram increment with wave aggregation of the most likely digit: the lanes that share the first
// participating lane's digit (all of them while the decided prefix is still common to every value, most
// of them on noise-dominated data) are counted by one atomic instead of serialising on one LDS word; the
// others add themselves.  Lanes with pred == false do not count.  Call with the whole wave converged.
__device__ __forceinline__ void hist_add(unsigned* h, unsigned digit, bool pred) {
  const unsigned long long act = __ballot(pred);
  if (!act) return;
  const int leader = __ffsll((long long)act) - 1;
  const unsigned d0 = (unsigned)__shfl((int)digit, leader);
  const unsigned long long same = __ballot(pred && digit == d0);
  if ((int)(threadIdx.x & 63) == leader) atomicAdd(&h[d0], (unsigned)__popcll(same));
  else if (pred && digit != d0) atomicAdd(&h[digit], 1u);
}

// one wave: the digit of a 256-bin histogram that holds rank k (0 <= k < total count) -> pick[0], and the count of
// everything in lower digits -> pick[1].  Four counters per lane, a shuffle scan, one lane owns the answer.
__device__ __forceinline__ void find_digit(const unsigned* hist, unsigned long long k, unsigned long long* pick) {
  const int l = threadIdx.x & 63;
  const unsigned long long c0 = hist[4 * l], c1 = hist[4 * l + 1], c2 = hist[4 * l + 2], c3 = hist[4 * l + 3];
  const unsigned long long sum = c0 + c1 + c2 + c3;
  unsigned long long inc = sum;
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned long long prev = __shfl_up(inc, d);
    if (l >= d) inc += prev;
  }
  unsigned long long cum = inc - sum;
  if (cum <= k && k < inc) {  // exactly one lane
    int d = 4 * l;
    if (k >= cum + c0) { cum += c0; ++d;
      if (k >= cum + c1) { cum += c1; ++d;
        if (k >= cum + c2) { cum += c2; ++d; } } }
    pick[0] = (unsigned long long)d;
    pick[1] = cum;
  }
}

// One digit of a radix select over n 64-bit keys produced by getkey(i); the whole workgroup cooperates (every thread
// must call it, with the same arguments).  The top `db` bits are decided (prefix holds them, lower bits zero); the digit
// is the next `width` (<= 8) bits.  Keys whose decided bits differ from prefix do not count (all_share: the caller
// knows that every key has them).  On return prefix has the digit, db has grown by width, k is the rank inside the
// digit's bucket and hist[digit] is still that bucket's size.  The digit holding rank k is found by wave 0.
template <int INFLIGHT = 4, class GetKey>
__device__ void block_digit_pass(GetKey getkey, long long n, long long& k, unsigned* hist /* [256] shared */,
                                 unsigned long long* pick /* [2] shared */, int& db, int width, unsigned long long& prefix,
                                 bool all_share) {
  const int shift = 64 - db - width;
  const unsigned dmask = (1u << width) - 1u;
  for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0u;
  __syncthreads();
  const bool nofilter = db == 0 || all_share;
  for (long long i0 = 0; i0 < n; i0 += (long long)INFLIGHT * blockDim.x) {  // uniform trip count: hist_add uses wave-wide votes
    unsigned long long key[INFLIGHT];
    bool in[INFLIGHT];
#pragma unroll
    for (int u = 0; u < INFLIGHT; ++u) {  // values in flight per thread
      const long long i = i0 + (long long)u * blockDim.x + threadIdx.x;
      in[u] = i < n;
      key[u] = in[u] ? getkey(i) : 0ull;
    }
#pragma unroll
    for (int u = 0; u < INFLIGHT; ++u)
      hist_add(hist, (unsigned)(key[u] >> shift) & dmask,
               in[u] && (nofilter || (key[u] >> (64 - db)) == (prefix >> (64 - db))));
  }
  __syncthreads();
  if (threadIdx.x < 64) find_digit(hist, (unsigned long long)k, pick);
  __syncthreads();
  prefix |= pick[0] << shift;
  k -= (long long)pick[1];
  db += width;
  __syncthreads();
}

// MATLAB median of n doubles produced by get(i) (any order; called by the whole workgroup with the same n).
// One scan finds the smallest and the largest key, whose common leading bits every key shares; the 8 bits right below
// spread the values over up to 256 buckets, so ONE digit pass usually leaves the middle value's bucket with a few dozen
// members (further passes only while it holds more than kCountingMedian); a third scan moves the bucket into
// `scratch` and remembers the largest key below it; the two middle order statistics are then found among the members
// by rank counting (every member counts the smaller ones).  Three scans instead of the nine of a full 8-digit select
// plus its counting pass.
template <class Get>
__device__ double block_median(Get get, long long n, unsigned* hist /* [256] shared */, unsigned long long* pick /* [2] shared */,
                               unsigned long long* scratch /* [kCountingMedian] shared */) {
  __shared__ unsigned long long s_min, s_max, s_ltmax, s_hi, s_lo;
  __shared__ unsigned s_n;
  if (threadIdx.x == 0) { s_min = ~0ull; s_max = 0ull; s_ltmax = 0ull; s_n = 0u; s_hi = 0ull; s_lo = 0ull; }
  __syncthreads();
  auto getkey = [&](long long i) { return dkey(get(i)); };
  {
    unsigned long long mn = ~0ull, mx = 0ull;
    for (long long i = threadIdx.x; i < n; i += blockDim.x) {
      const unsigned long long k = getkey(i);
      mn = k < mn ? k : mn;
      mx = k > mx ? k : mx;
    }
    atomicMin(&s_min, mn);
    atomicMax(&s_max, mx);
  }
  __syncthreads();
  const unsigned long long kmin = s_min, kmax = s_max;
  long long r = n / 2;  // rank of the upper middle value
  int db = kmin == kmax ? 64 : __clzll((long long)(kmin ^ kmax));  // bits every key shares
  unsigned long long pfx = db == 64 ? kmin : (db ? kmin & (~0ull << (64 - db)) : 0ull);
  unsigned long long bucket = (unsigned long long)n;
  bool first = true;
  while (db < 64 && bucket > (unsigned long long)kCountingMedian) {  // uniform
    const int width = 64 - db < 8 ? 64 - db : 8;
    block_digit_pass(getkey, n, r, hist, pick, db, width, pfx, first);
    first = false;
    bucket = hist[(unsigned)(pfx >> (64 - db)) & ((1u << width) - 1u)];
    __syncthreads();
  }
  unsigned long long khi, klo;
  if (db == 64) {  // the bucket is one value (all keys equal, or heavy ties)
    khi = pfx;
    klo = pfx;
    if (r == 0 && (n & 1) == 0) {  // the lower middle value is the largest key below
      unsigned long long mx = 0ull;
      for (long long i = threadIdx.x; i < n; i += blockDim.x) {
        const unsigned long long k = getkey(i);
        if (k < pfx) mx = k > mx ? k : mx;
      }
      if (mx) atomicMax(&s_ltmax, mx);
      __syncthreads();
      klo = s_ltmax;
    }
  } else {
    const unsigned long long dmask = db == 0 ? 0ull : ~0ull << (64 - db);
    const int lane = threadIdx.x & 63;
    unsigned long long mx = 0ull;
    for (long long i0 = 0; i0 < n; i0 += blockDim.x) {  // uniform trip count: wave-wide votes
      const long long i = i0 + threadIdx.x;
      const unsigned long long k = i < n ? getkey(i) : 0ull;
      const bool in = i < n && (k & dmask) == pfx;
      if (i < n && k < pfx) mx = k > mx ? k : mx;
      const unsigned long long vote = __ballot(in);
      if (vote) {
        const int leader = __ffsll((long long)vote) - 1;
        unsigned base = 0u;
        if (lane == leader) base = atomicAdd(&s_n, (unsigned)__popcll(vote));
        base = (unsigned)__shfl((int)base, leader);
        if (in) scratch[base + (unsigned)__popcll(vote & ((1ull << lane) - 1ull))] = k;
      }
    }
    if (mx) atomicMax(&s_ltmax, mx);
    __syncthreads();
    const int m = (int)bucket;
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
      const unsigned long long ki = scratch[i];
      long long rank = 0;
      for (int j = 0; j < m; ++j) {
        const unsigned long long kj = scratch[j];
        rank += (kj < ki) || (kj == ki && j < i);
      }
      if (rank == r) s_hi = ki;
      if (rank == r - 1) s_lo = ki;
    }
    __syncthreads();
    khi = s_hi;
    klo = r > 0 ? s_lo : s_ltmax;
  }
  const double hi = dkey_inv(khi);
  const double res = (n & 1) ? hi : 0.5 * (dkey_inv(klo) + hi);
  __syncthreads();  // the shared words are free for the next call
  return res;
}

// median of the n <= kCountingMedian values in v[] (LDS) by rank counting: element i has rank
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
// grid = (column groups of 64, row blocks, selects); block = 256 (4 waves).  blockIdx.z picks one of several
// independent selects over the same rows (prefix[z][M], hist[z][M][256]): the two bracket ranks run together.
__global__ void __launch_bounds__(256) pdw_hist_kernel(const float2* y, long long F, long long stride, int M, int pass,
                                                       const unsigned long long* prefix, unsigned* hist) {
  prefix += (size_t)blockIdx.z * M;
  hist += (size_t)blockIdx.z * M * 256;
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
    auto count = [&](float2 v) {
      const unsigned long long k = dkey(mag2_of(v));  // ordered like the magnitude, no sqrt
      const bool in_bucket = (pass == 0) || ((k >> (shift + 8)) == (pre >> (shift + 8)));
      if (in_bucket) atomicAdd(&h[(unsigned)(k >> shift) & 255u][lane], 1u);
    };
    long long r = r0 + wave;
    for (; r + 28 < r1; r += 32) {  // eight rows in flight per lane: the sampled rows are far apart, each a fresh HBM line
      float2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = y[sample_row(r + 4 * u, stride) * M + col];
#pragma unroll
      for (int u = 0; u < 8; ++u) count(v[u]);
    }
    for (; r < r1; r += 4) count(y[sample_row(r, stride) * M + col]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 256 * 64; i += 256) {
    const int d = i >> 6, c = i & 63;
    const unsigned v = h[d][c];
    if (v && (int)(blockIdx.x * 64) + c < M) atomicAdd(&hist[(size_t)(blockIdx.x * 64 + c) * 256 + d], v);
  }
}

// thr = noise floor * 10^(SNR/10) on the device, so the edge stage can be queued before the host has seen the medians
__global__ void pdw_thr_kernel(const double* nf, double gain, double* thr, int M) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < M) thr[i] = nf[i] * gain;
}

// choose the digit holding rank[col]; one wave per column (four counters per lane, a shuffle scan, one lane owns
// the answer), four columns per workgroup
__global__ void __launch_bounds__(256) pdw_pick_kernel(int M, int pass, unsigned* hist, unsigned long long* prefix,
                                                       unsigned long long* rank, unsigned* bucket, unsigned long long* below) {
  const int col = blockIdx.x * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;
  if (col >= M) return;
  unsigned* hc = hist + (size_t)col * 256;
  const uint4 c4 = *reinterpret_cast<const uint4*>(hc + 4 * l);
  const unsigned long long c0 = c4.x, c1 = c4.y, c2 = c4.z, c3 = c4.w, sum = c0 + c1 + c2 + c3;
  unsigned long long inc = sum;
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned long long prev = __shfl_up(inc, d);
    if (l >= d) inc += prev;
  }
  unsigned long long cum = inc - sum;
  const unsigned long long r = rank[col];
  // the lane whose counters hold rank r; a rank past the total (cannot happen: r < count) would fall to digit 255
  const bool last = (l == 63) && r >= inc;
  if ((cum <= r && r < inc) || last) {
    int d = 4 * l;
    unsigned cnt = (unsigned)c0;
    if (r >= cum + c0) { cum += c0; ++d; cnt = (unsigned)c1;
      if (r >= cum + c1) { cum += c1; ++d; cnt = (unsigned)c2;
        if (r >= cum + c2) { cum += c2; ++d; cnt = (unsigned)c3; } } }
    prefix[col] |= (unsigned long long)d << (56 - 8 * pass);
    rank[col] = r - cum;        // rank inside the chosen bucket
    below[col] += cum;          // elements strictly below the bucket so far
    bucket[col] = cnt;
  }
  *reinterpret_cast<uint4*>(hc + 4 * l) = make_uint4(0u, 0u, 0u, 0u);
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

// The sampled rows are read ONCE: the top 32 bits of every sampled |y|^2 key (the sample decides kSamplePasses = 3
// digits = 24 bits) go to keys[channel][sample], transposed through LDS so that the per-channel select streams them.
// grid = (column groups of 64, sample blocks of 64 rows); sixteen far-apart rows in flight per lane.
__global__ void __launch_bounds__(256) pdw_sample_gather_kernel(const float2* y, long long ns, long long stride, int M,
                                                                unsigned* keys, long long ld) {
  __shared__ unsigned tile[64][65];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  const long long q0 = (long long)blockIdx.y * 64;
  float2 v[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const long long q = q0 + wave * 16 + u;
    v[u] = (col < M && q < ns) ? y[sample_row(q, stride) * M + col] : make_float2(0.f, 0.f);
  }
#pragma unroll
  for (int u = 0; u < 16; ++u) tile[wave * 16 + u][lane] = (unsigned)(dkey(mag2_of(v[u])) >> 32);
  __syncthreads();
  for (int c = wave; c < 64; c += 4) {
    const int gc = blockIdx.x * 64 + c;
    const long long q = q0 + lane;
    if (gc < M && q < ns) keys[(size_t)gc * ld + q] = tile[lane][c];
  }
}

// both bracket ranks of one channel's sample, kSamplePasses digits each; one workgroup per channel.  The channel's
// keys (ns <= 1024 * kSampleKeysPerThread, guaranteed by F >= 8 * kSampleRows) are read once into registers; every
// pass counts both selects (two histograms), wave 0 and wave 1 find their digits side by side.
constexpr int kSampleKeysPerThread = 72;
static_assert(1024ll * kSampleKeysPerThread >= (long long)kSampleRows * 9 / 8, "ns < kSampleRows * (stride + 1) / stride, stride >= 8");
__global__ void __launch_bounds__(1024) pdw_sample_select_kernel(const unsigned* keys, long long ns, long long ld,
                                                                 unsigned long long rank_lo, unsigned long long rank_hi,
                                                                 unsigned long long* pre_lo, unsigned long long* pre_hi) {
  __shared__ unsigned hist[2][256];
  __shared__ unsigned long long pick[2][2];
  const uint4* k4 = reinterpret_cast<const uint4*>(keys + (size_t)blockIdx.x * ld);
  constexpr int kQuads = kSampleKeysPerThread / 4;
  uint4 kq[kQuads];
#pragma unroll
  for (int j = 0; j < kQuads; ++j) {
    const long long q = (long long)j * 1024 + threadIdx.x;
    kq[j] = (q * 4 < ld) ? k4[q] : make_uint4(0u, 0u, 0u, 0u);
  }
  unsigned pre[2] = {0u, 0u};                       // decided digits of the two 32-bit key prefixes
  unsigned long long rk[2] = {rank_lo, rank_hi};
#pragma unroll 1
  for (int pass = 0; pass < kSamplePasses; ++pass) {
    const int shift = 24 - 8 * pass;
    for (int i = threadIdx.x; i < 512; i += 1024) (&hist[0][0])[i] = 0u;
    __syncthreads();
    const unsigned hmask = pass ? ~0u << (shift + 8) : 0u;  // the digits already decided
    const bool split = pre[0] != pre[1];  // the two ranks sit in one bucket until their digits part: one histogram serves both
#pragma unroll
    for (int j = 0; j < kQuads; ++j) {
      const long long base = ((long long)j * 1024 + threadIdx.x) * 4;
      const unsigned kk[4] = {kq[j].x, kq[j].y, kq[j].z, kq[j].w};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool valid = base + u < ns;
        const unsigned digit = (kk[u] >> shift) & 255u;
        hist_add(hist[0], digit, valid && (kk[u] & hmask) == pre[0]);
        if (split) hist_add(hist[1], digit, valid && (kk[u] & hmask) == pre[1]);  // uniform over the workgroup
      }
    }
    __syncthreads();
    if (threadIdx.x < 128) find_digit(hist[split ? threadIdx.x >> 6 : 0], rk[threadIdx.x >> 6], pick[threadIdx.x >> 6]);
    __syncthreads();
#pragma unroll
    for (int z = 0; z < 2; ++z) {
      pre[z] |= (unsigned)pick[z][0] << shift;
      rk[z] -= pick[z][1];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    pre_lo[blockIdx.x] = (unsigned long long)pre[0] << 32;
    pre_hi[blockIdx.x] = (unsigned long long)pre[1] << 32;
  }
}

// One pass over the data with the bracket [lo, hi] of every channel (key prefixes from the sample, low
// bits cleared / set): count what lies below, gather what lies inside.
//
// The same pass writes the edge machine's comparison masks.  The threshold is gain * median, and the
// median lies in [sqrt(lo), sqrt(hi)], so |y|^2 below lo * gain^2 is certainly under the threshold and
// above hi * gain^2 certainly over it (both with a 1e-9 guard band); the few samples in between are
// listed and classified exactly once the median is known (pdw_patch_kernel).  One word (64 frames)
// per wave at a time, lane = channel.
//
// Every sample is SCREENED in float32: m32 = fl(x^2 + y^2) is within 2^-23 of the exact |y|^2, and four
// per-channel float32 limits set 2^-19 outside lo / hi / t2lo / t2hi tell "surely below the bracket", "surely
// above it" and "surely over / under the threshold" in a dozen instructions.  A sample inside the bracket's
// zone (the 2 % candidates plus a 4e-6 wide rim) is parked as it is, 8 bytes, in a staging column that belongs
// to its (wave, lane, channel) -- a register counts the slots, no atomics -- and the float64 classification
// (below / inside / above, exactly as the unscreened pass did) happens once per workgroup when the columns are
// flushed: one channel per wave at a time, lanes = (source wave, slot), candidates appended as one contiguous
// run per channel.  A full column (16 slots; ~5 expected) classifies on the spot.  Samples inside the
// threshold's zone are a handful: reloaded and classified exactly.  max_below covers the zone only: it is the
// true maximum below lo whenever it is non-zero, and the finish kernel asks for a redo in the (never seen)
// case that needs it and finds it zero.
__device__ __forceinline__ void bracket_screen(double lo, double hi, float& a, float& b) {
  if (lo > 1e-30 && hi < 1e30) {  // float32 keeps its relative accuracy here
    a = (float)(lo * (1.0 - 0x1p-19));
    b = (float)(hi * (1.0 + 0x1p-19));
  } else {  // everything is "inside the zone": the exact route decides
    a = 0.0f;
    b = INFINITY;
  }
}

constexpr int kBracketSlots = 16;  // staging slots per (wave, channel): 4 waves x 16 slots = the 64 lanes of the flush

// grid = (column groups of 64, a few workgroups per CU); a workgroup walks row groups of kBracketRows frames
// (long-lived workgroups read faster than thousands of short ones), flushing its staging columns after each.
// LPR = lanes per row.  64: lane = channel, column groups of 64 (blockIdx.x).  8 / 16 / 32 for M <= LPR (the small
// banks: numBands = fs * 1e-6 at 8 ... 32 Msps): a wave-load covers 64 / LPR consecutive rows, lane (sub, channel) owns
// rows sub, sub + RPW, ... of a 64-row word -- every lane loads, where lane = channel would leave 7 of 8 idle at M = 8 --
// and the word of a channel is the OR of its RPW lanes' bits.
template <int LPR>
__global__ void __launch_bounds__(256) pdw_bracket_kernel(const float2* y, long long F, int M,
                                                          const unsigned long long* pre_lo, const unsigned long long* pre_hi,
                                                          double gain2, double* cand, unsigned cap, unsigned* cand_n,
                                                          unsigned long long* below, unsigned long long* max_below,
                                                          unsigned long long* f0, unsigned long long* f1, long long words,
                                                          unsigned long long* undecided, unsigned* und_n, unsigned* flags,
                                                          int row_groups) {
  constexpr int RPW = 64 / LPR;                                            // rows per wave-load
  constexpr int kBatch = kBracketInFlight < LPR ? kBracketInFlight : LPR;  // a lane owns LPR rows of a word
  // (rows of 65: the flush reads one column c with lanes = (wave, slot) -- 64 rows -- and with rows of 64 float2 every
  // one of those reads hit the same bank pair, a 32-way conflict: 38 % of the LDS's active cycles in round 2's counters)
  __shared__ float2 stage[4][kBracketSlots][65];
  __shared__ unsigned char cnt[4][64];
  __shared__ unsigned cand_cnt[64], cand_base[64];
  __shared__ unsigned long long below_acc[64];  // per channel of this workgroup: "below" counts, sent out once at the end
  if (threadIdx.x < 64) below_acc[threadIdx.x] = 0ull;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / LPR;                                              // 0 when LPR == 64
  const int col = LPR == 64 ? blockIdx.x * 64 + lane : lane % LPR;
  const int lane_off = sub * M + col;                                      // element offset of this lane inside a wave-load
  auto chan_of = [&](int c) { return LPR == 64 ? (int)blockIdx.x * 64 + c : c % LPR; };  // channel of staging column c
  const bool valid = col < M;
  constexpr unsigned long long kLow = (1ull << (64 - 8 * kSamplePasses)) - 1ull;
  constexpr int kWordsPerBlock = kBracketRows / 64;
  const unsigned long long lo = valid ? pre_lo[col] & ~kLow : 0ull, hi = valid ? pre_hi[col] | kLow : 0ull;
  const double t2lo = dkey_inv(lo) * gain2 * (1.0 - 1e-9), t2hi = dkey_inv(hi) * gain2 * (1.0 + 1e-9);
  float sA, sB, sC, sD;
  bracket_screen(dkey_inv(lo), dkey_inv(hi), sA, sB);
  bracket_screen(t2lo, t2hi, sC, sD);
  unsigned long long nb = 0ull, best = 0ull;
  const int ws = lane / kBracketSlots, sl = lane % kBracketSlots;  // the flush's view of a lane

  for (int rgi = blockIdx.y; rgi < row_groups; rgi += gridDim.y) {
    // last rows first: when the matrix has just been written (the channelizer ran right before), its tail is still in
    // the 256 MB Infinity Cache (tools/mall_probe.py: a 256 MB buffer reads back 1.4x faster than a large one)
    const int rg = row_groups - 1 - rgi;
    unsigned n = 0u, nb32 = 0u;
    if (valid) {
      for (int wi = wave; wi < kWordsPerBlock; wi += 4) {
        const long long w = (long long)rg * kWordsPerBlock + wi;
        if (w >= words) break;
        const long long r0 = w * 64;
        // the same number for the scalar unit (w depends on the wave only)
        const long long r0s = (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(r0 >> 32)) << 32) |
                                          (unsigned)__builtin_amdgcn_readfirstlane((int)r0));
        unsigned long long over = 0ull;
        // the float64 route, on the spot: a full staging column, the threshold's zone, the ragged last word
        auto exact = [&](float2 v, int i, bool for_median, bool for_mask) {
          const double m = mag2_of(v);
          if (for_median) {
            const unsigned long long k = dkey(m);
            if (k < lo) {
              ++nb;
              best = k > best ? k : best;
            } else if (k <= hi) {
              const unsigned g = atomicAdd(&cand_n[col], 1u);
              if (g < cap) cand[(size_t)col * cap + g] = m;
              else atomicOr(flags, 1u);
            }
          }
          if (for_mask) {
            if (m > t2hi) {
              over |= 1ull << i;
            } else if (m >= t2lo) {
              const unsigned u = atomicAdd(und_n, 1u);
              if (u < (unsigned)kUndecided) undecided[u] = (unsigned long long)(r0 + i) * (unsigned long long)M + (unsigned)col;
              else atomicOr(flags, 4u);
            }
          }
        };
        unsigned long long pad = 0ull;  // frames past F: identity (f0 = 0, f1 = 1)
        if (r0 + 64 <= F) {
          // The row address is wave-uniform arithmetic on the scalar unit (a 64-bit multiply by M per load on the vector
          // unit otherwise), and the two threshold screens of a sample are one running maximum per batch: only a lane
          // whose batch reaches the threshold's lower limit looks at its samples again.  A third fewer vector instructions
          // -- and no faster (905 against 910 us in the same process): the pass is bound by its access shape.
          const float2* rows = y + r0s * M;
          for (int i = 0; i < LPR; i += kBatch) {  // the lane's rows r0 + (i + u) RPW + sub, kBatch of them in flight
            float2 v[kBatch];
#pragma unroll
            for (int u = 0; u < kBatch; ++u) v[u] = (rows + (long long)((i + u) * RPW) * M)[lane_off];
            unsigned ov = 0u, ub = 0u, sb = 0u;
            float mx = 0.0f;
#pragma unroll
            for (int u = 0; u < kBatch; ++u) {
              const float m32 = __fmaf_rn(v[u].x, v[u].x, __fmul_rn(v[u].y, v[u].y));
              const bool is_below = m32 < sA;
              nb32 += (unsigned)is_below;
              if (!is_below && !(m32 > sB)) {
                if (n < (unsigned)kBracketSlots) stage[wave][n][lane] = v[u];
                else sb |= 1u << u;
                ++n;
              }
              mx = fmaxf(mx, m32);  // (a NaN is skipped: it would pass neither threshold comparison anyway)
            }
            if (!(mx < sC)) {
#pragma unroll
              for (int u = 0; u < kBatch; ++u) {
                const float m32 = __fmaf_rn(v[u].x, v[u].x, __fmul_rn(v[u].y, v[u].y));
                ov |= (unsigned)(m32 > sD) << u;
                ub |= (unsigned)(!(m32 < sC) && !(m32 > sD)) << u;
              }
            }
            if constexpr (RPW == 1) {
              over |= (unsigned long long)ov << i;
            } else {
              while (ov) {  // bit u of the batch is row (i + u) RPW + sub of the word
                const int u = __ffs((int)ov) - 1;
                ov &= ov - 1u;
                over |= 1ull << ((i + u) * RPW + sub);
              }
            }
            unsigned bits = ub | sb;
            while (bits) {  // the threshold's zone, a full column: reload (the line is in cache) and classify exactly
              const int u = __ffs((int)bits) - 1;
              bits &= bits - 1u;
              const int row = (i + u) * RPW + sub;
              exact(y[(r0 + row) * M + col], row, (sb >> u) & 1u, (ub >> u) & 1u);
            }
          }
        } else {
          for (int t = 0; t < LPR; ++t) {
            const int row = t * RPW + sub;
            if (r0 + row < F) exact(y[(r0 + row) * M + col], row, true, true);
            else pad |= 1ull << row;
          }
        }
        if constexpr (RPW > 1) {  // a channel's word = its RPW lanes' rows
#pragma unroll
          for (int d = LPR; d < 64; d <<= 1) {
            over |= __shfl_xor(over, d);
            pad |= __shfl_xor(pad, d);
          }
        }
        if (sub == 0) {
          f0[w * M + col] = over;
          f1[w * M + col] = over | pad;
        }
      }
    }
    nb += nb32;
    cnt[wave][lane] = (unsigned char)(n < (unsigned)kBracketSlots ? n : (unsigned)kBracketSlots);
    __syncthreads();
    // flush: one channel per wave at a time, lane = (source wave, slot); exact classification of the parked samples.
    // Counting first, then ONE round of appends to the global candidate counters for all the channels at once (a
    // returning atomic per channel inside the loop would serialise sixteen memory round trips per wave), then the stores.
    auto classify = [&](int c, int gc, double& m, unsigned long long& k, bool& is_below, bool& is_cand) {
      const bool has = ws < 4 && sl < (int)cnt[ws < 4 ? ws : 0][c];
      const unsigned long long klo = pre_lo[gc] & ~kLow, khi = pre_hi[gc] | kLow;
      const float2 v = has ? stage[ws < 4 ? ws : 0][sl][c] : make_float2(0.f, 0.f);
      m = mag2_of(v);
      k = dkey(m);
      is_below = has && k < klo;
      is_cand = has && k >= klo && k <= khi;
    };
    for (int c = wave; c < 64; c += 4) {
      const int gc = chan_of(c);
      if (gc >= M) {  // uniform over the wave
        if (lane == 0) cand_cnt[c] = 0u;
        continue;
      }
      double m;
      unsigned long long k;
      bool is_below, is_cand;
      classify(c, gc, m, k, is_below, is_cand);
      const unsigned long long vb = __ballot(is_below), vc = __ballot(is_cand);
      if (lane == 0) cand_cnt[c] = (unsigned)__popcll(vc);
      if (vb) {
        if (lane == __ffsll((long long)vb) - 1) atomicAdd(&below_acc[c % LPR], (unsigned long long)__popcll(vb));
        if (is_below) atomicMax(&max_below[gc], k);
      }
    }
    __syncthreads();
    // one append per CHANNEL and workgroup (64 / LPR staging columns share a channel when rows are packed: with a
    // returning atomic per column the eight channels of an M = 8 bank took 2 M of them each -- 5.9 ms for 2 GB)
    if (threadIdx.x < LPR) {
      const int gc = chan_of((int)threadIdx.x);
      unsigned tot = 0u;
#pragma unroll
      for (int j = 0; j < RPW; ++j) tot += cand_cnt[threadIdx.x + j * LPR];
      unsigned b0 = (gc < M && tot) ? atomicAdd(&cand_n[gc], tot) : 0u;
#pragma unroll
      for (int j = 0; j < RPW; ++j) {
        cand_base[threadIdx.x + j * LPR] = b0;
        b0 += cand_cnt[threadIdx.x + j * LPR];
      }
    }
    __syncthreads();
    for (int c = wave; c < 64; c += 4) {
      const int gc = chan_of(c);
      if (gc >= M || cand_cnt[c] == 0u) continue;  // uniform over the wave
      double m;
      unsigned long long k;
      bool is_below, is_cand;
      classify(c, gc, m, k, is_below, is_cand);
      const unsigned long long vc = __ballot(is_cand);
      if (is_cand) {
        const unsigned pos = cand_base[c] + (unsigned)__popcll(vc & ((1ull << lane) - 1ull));
        if (pos < cap) cand[(size_t)gc * cap + pos] = m;
        else atomicOr(flags, 1u);
      }
    }
    __syncthreads();  // the staging columns are free again
  }
  if (valid) {
    if (nb) atomicAdd(&below_acc[lane % LPR], nb);
    if (best) atomicMax(&max_below[col], best);
  }
  __syncthreads();
  if (threadIdx.x < LPR) {
    const int gc = chan_of((int)threadIdx.x);
    if (gc < M && below_acc[threadIdx.x]) atomicAdd(&below[gc], below_acc[threadIdx.x]);
  }
}

// exact order statistics among the gathered candidates; one workgroup per channel.  The median's rank
// must fall inside the candidate set -- that is the proof the sampled bracket held it.  The leading bits
// lo and hi share are known, so the select starts right below them: ONE histogram pass over the candidates on the
// next 8 bits, a second pass that moves that digit's bucket (1/100 of them or so) into LDS, and the
// remaining bits are decided there.  The lower middle value of an even count is the largest candidate below
// the upper one unless that one repeats.  Also checks that the threshold really lies inside the band the
// provisional masks assumed (flag 8 if not).
constexpr int kFinishLds = 4096;  // bucket members held in LDS; a larger bucket (heavily tied data) keeps selecting in memory

// The select is split over gridDim.y workgroups per channel (one workgroup scanning a channel's 84 000 candidates twice
// was 0.12 ms on 128 of the 256 CUs, and 1 ms for the 670 000 candidates of an M = 8 matrix on 8 of them):
// pdw_finish_hist_kernel -- every part histograms its share of the candidates on the first undecided digit into
// fin.hist; pdw_bracket_finish_kernel -- every part finds the median's digit in that histogram, moves its share of
// that digit's bucket into fin.bucket, and the LAST part to arrive (a ticket) holds the bucket in LDS and finishes.
constexpr int kFinishBits = 11, kFinishBins = 1 << kFinishBits;  // the first digit: wide enough to leave <= kFinishLds members of 4 M candidates
struct FinishShared {
  unsigned* hist;               // [M][kFinishBins] first-digit histogram of the candidates
  unsigned long long* bucket;   // [M][kFinishLds] keys of the median's bucket
  unsigned* bucket_n;           // [M]
  unsigned long long* lt_max;   // [M] largest candidate key below the bucket
  unsigned* ticket;             // [M]
};

// what every part derives from the channel's counters; false: the bracket did not hold the median (or overflowed)
struct FinishSetup {
  unsigned long long n, lo, hi;
  long long r0;
  int shared_bits;
};
__device__ __forceinline__ bool finish_setup(int col, long long F, unsigned cap, const unsigned* cand_n,
                                             const unsigned long long* below, const unsigned long long* pre_lo,
                                             const unsigned long long* pre_hi, FinishSetup& q) {
  constexpr unsigned long long kLow = (1ull << (64 - 8 * kSamplePasses)) - 1ull;
  const unsigned long long b = below[col], target = (unsigned long long)(F / 2);
  q.n = cand_n[col];
  if (q.n > cap || b > target || target - b >= q.n) return false;
  q.lo = pre_lo[col] & ~kLow;
  q.hi = pre_hi[col] | kLow;
  q.shared_bits = q.lo == q.hi ? 64 : __clzll((long long)(q.lo ^ q.hi));  // leading bits every candidate has
  q.r0 = (long long)(target - b);  // the upper middle value's rank among the candidates
  return true;
}

__global__ void __launch_bounds__(1024) pdw_finish_hist_kernel(long long F, const double* cand, unsigned cap,
                                                               const unsigned* cand_n, const unsigned long long* below,
                                                               const unsigned long long* pre_lo,
                                                               const unsigned long long* pre_hi, FinishShared fin) {
  __shared__ unsigned hist[kFinishBins];
  const int col = blockIdx.x;
  FinishSetup q;
  if (!finish_setup(col, F, cap, cand_n, below, pre_lo, pre_hi, q) || q.shared_bits == 64) return;  // uniform
  const int width = 64 - q.shared_bits < kFinishBits ? 64 - q.shared_bits : kFinishBits, shift = 64 - q.shared_bits - width;
  const unsigned dmask = (1u << width) - 1u;
  const double* v = cand + (size_t)col * cap;
  const long long i_begin = (long long)(q.n * blockIdx.y / gridDim.y), i_end = (long long)(q.n * (blockIdx.y + 1) / gridDim.y);
  for (int i = threadIdx.x; i < kFinishBins; i += blockDim.x) hist[i] = 0u;
  __syncthreads();
  for (long long i0 = i_begin; i0 < i_end; i0 += 8ll * blockDim.x) {  // uniform trip count: hist_add uses wave-wide votes
    unsigned long long kk[8];
    bool in[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long long i = i0 + (long long)u * blockDim.x + threadIdx.x;
      in[u] = i < i_end;
      kk[u] = in[u] ? dkey(v[i]) : 0ull;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) hist_add(hist, (unsigned)(kk[u] >> shift) & dmask, in[u]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kFinishBins; i += blockDim.x)
    if (hist[i]) atomicAdd(&fin.hist[(size_t)col * kFinishBins + i], hist[i]);
}

__global__ void __launch_bounds__(1024) pdw_bracket_finish_kernel(long long F, const double* cand, unsigned cap,
                                                                  const unsigned* cand_n, const unsigned long long* below,
                                                                  const unsigned long long* max_below,
                                                                  const unsigned long long* pre_lo,
                                                                  const unsigned long long* pre_hi, double gain, double* nf,
                                                                  unsigned* flags, FinishShared fin) {
  __shared__ unsigned hist[256];
  __shared__ unsigned long long pick[2];
  __shared__ unsigned long long lt_count, lt_max;
  __shared__ unsigned long long members[kFinishLds];
  __shared__ unsigned members_n, my_ticket;
  const int col = blockIdx.x, part = blockIdx.y, parts = gridDim.y;
  FinishSetup q;
  if (!finish_setup(col, F, cap, cand_n, below, pre_lo, pre_hi, q)) {  // uniform over the workgroup
    if (part == 0 && threadIdx.x == 0) { atomicOr(flags, 2u); nf[col] = 0.0; }
    return;
  }
  const unsigned long long n = q.n, lo = q.lo, hi = q.hi;
  const int shared_bits = q.shared_bits;
  const long long r0 = q.r0;
  const double* v = cand + (size_t)col * cap;
  auto getkey = [&](long long i) { return dkey(v[i]); };
  const bool even = (F & 1) == 0;
  if (threadIdx.x == 0) { lt_count = 0ull; lt_max = 0ull; members_n = 0u; }
  unsigned long long k1;
  bool lower_known = false;  // lt_count / lt_max already hold the candidates below k1
  if (shared_bits == 64) {
    if (part != 0) return;
    k1 = lo;
  } else {
    long long r = r0;
    int db = shared_bits;  // bits decided so far
    unsigned long long pfx = db ? lo & (~0ull << (64 - db)) : 0ull;
    unsigned bucket;
    {  // the first digit (kFinishBits wide): every part reads the histogram all the parts built (pdw_finish_hist_kernel)
      const int width = 64 - db < kFinishBits ? 64 - db : kFinishBits, shift = 64 - db - width;
      if (threadIdx.x < 64) {  // wave 0: kFinishBins / 64 bins per lane, a shuffle scan, the owning lane walks its bins
        constexpr int PER = kFinishBins / 64;
        const unsigned* hc = fin.hist + (size_t)col * kFinishBins + threadIdx.x * PER;
        unsigned long long sum = 0ull;
        for (int j = 0; j < PER; ++j) sum += hc[j];
        unsigned long long inc = sum;
        for (int d = 1; d < 64; d <<= 1) {
          const unsigned long long prev = __shfl_up(inc, d);
          if ((int)threadIdx.x >= d) inc += prev;
        }
        unsigned long long cum = inc - sum;
        const unsigned long long kk = (unsigned long long)r;
        if (cum <= kk && kk < inc) {  // exactly one lane
          int j = 0;
          for (; j < PER - 1; ++j) {
            if (kk < cum + hc[j]) break;
            cum += hc[j];
          }
          pick[0] = (unsigned long long)(threadIdx.x * PER + j);
          pick[1] = cum;
          lt_count = hc[j];  // (borrowed until the setup below: the bucket's size)
        }
      }
      __syncthreads();
      pfx |= pick[0] << shift;
      r -= (long long)pick[1];
      db += width;
      bucket = (unsigned)lt_count;
      __syncthreads();
      if (threadIdx.x == 0) lt_count = 0ull;
    }
    if (bucket > (unsigned)kFinishLds && part != 0) return;  // heavily tied data: part 0 keeps selecting in memory, alone
    // further passes over all the candidates until the bucket fits LDS (the 8 bits right below the shared ones spread
    // the bracket's population over up to 256 buckets, so this loop does not run as a rule)
    const bool alone = bucket > (unsigned)kFinishLds || parts == 1;
    while (db < 64 && bucket > (unsigned)kFinishLds) {  // uniform
      const int width = 64 - db < 8 ? 64 - db : 8;
      block_digit_pass<16>(getkey, (long long)n, r, hist, pick, db, width, pfx, false);
      bucket = hist[(unsigned)(pfx >> (64 - db)) & ((1u << width) - 1u)];
      __syncthreads();
    }
    if (db == 64) {
      k1 = pfx;
    } else {
      // move the bucket out of the candidates: this part's share (everything when it works alone), slots claimed per
      // wave; remember the largest candidate below the bucket
      const unsigned long long dmask = db == 0 ? 0ull : ~0ull << (64 - db);
      const int lane = threadIdx.x & 63;
      const long long i_begin = alone ? 0 : (long long)(n * part / parts), i_end = alone ? (long long)n : (long long)(n * (part + 1) / parts);
      unsigned long long* dst = members;  // LDS first (a returning global atomic per vote would chain memory round trips)
      unsigned* dst_n = &members_n;
      unsigned long long mx = 0ull;
      for (long long i0 = i_begin; i0 < i_end; i0 += 8ll * blockDim.x) {  // uniform trip count: wave-wide votes
        unsigned long long kk[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {  // eight candidates in flight per thread
          const long long i = i0 + (long long)u * blockDim.x + threadIdx.x;
          kk[u] = i < i_end ? getkey(i) : ~0ull;  // ~0 is neither a member nor below
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const unsigned long long k = kk[u];
          const bool in = k != ~0ull && (k & dmask) == pfx;
          if (k != ~0ull && k < pfx) mx = k > mx ? k : mx;
          const unsigned long long vote = __ballot(in);
          if (vote) {
            const int leader = __ffsll((long long)vote) - 1;
            unsigned base = 0u;
            if (lane == leader) base = atomicAdd(dst_n, (unsigned)__popcll(vote));
            base = (unsigned)__shfl((int)base, leader);
            if (in) dst[base + (unsigned)__popcll(vote & ((1ull << lane) - 1ull))] = k;
          }
        }
      }
      if (mx) atomicMax(&lt_max, mx);
      __syncthreads();
      if (!alone) {
        // hand over: this part's members go to the channel's bucket in memory (one slot claim per part), and the last
        // part to arrive finds every part's members and maxima there
        if (threadIdx.x == 0) my_ticket = members_n ? atomicAdd(&fin.bucket_n[col], members_n) : 0u;  // (borrowed: the base)
        __syncthreads();
        {
          unsigned long long* gb = fin.bucket + (size_t)col * kFinishLds + my_ticket;
          for (unsigned i = threadIdx.x; i < members_n; i += blockDim.x) gb[i] = members[i];
        }
        __syncthreads();  // (the workgroup's stores have left for L2)
        if (threadIdx.x == 0) {
          if (lt_max) atomicMax(&fin.lt_max[col], lt_max);
          __threadfence();  // one release per workgroup: L2 written back before the ticket is drawn (a fence per
                            // thread made this kernel 0.43 ms)
          my_ticket = atomicAdd(&fin.ticket[col], 1u);
          if (my_ticket == (unsigned)parts - 1u) __threadfence();  // the last part: acquire before it reads the others' members
        }
        __syncthreads();
        if (my_ticket != (unsigned)parts - 1u) return;  // uniform
        const volatile unsigned long long* src = fin.bucket + (size_t)col * kFinishLds;
        for (unsigned i = threadIdx.x; i < bucket; i += blockDim.x) members[i] = src[i];
        if (threadIdx.x == 0) lt_max = *reinterpret_cast<const volatile unsigned long long*>(&fin.lt_max[col]);
        __syncthreads();
      }
      const long long r_in = r;  // rank inside the bucket
      bool first = true;
      while (db < 64) {  // the remaining bits, decided among the members
        const int width = 64 - db < 8 ? 64 - db : 8;
        block_digit_pass([&](long long i) { return members[i]; }, (long long)bucket, r, hist, pick, db, width, pfx, first);
        first = false;
      }
      k1 = pfx;
      if (even && r0 > 0) {
        unsigned long long c = 0ull, m2 = 0ull;
        for (unsigned i = threadIdx.x; i < bucket; i += blockDim.x) {
          const unsigned long long k = members[i];
          if (k < k1) { ++c; m2 = k > m2 ? k : m2; }
        }
        if (c) { atomicAdd(&lt_count, c); atomicMax(&lt_max, m2); }  // members outrank everything below the bucket
        __syncthreads();
        if (threadIdx.x == 0) lt_count += (unsigned long long)(r0 - r_in);  // candidates in the lower buckets
        __syncthreads();
        lower_known = true;
      }
    }
  }
  const double v1 = dkey_inv(k1);
  double res;
  if (!even) {
    res = sqrt(v1);
  } else {
    double v0;
    if (r0 == 0) {
      const unsigned long long mb = max_below[col];
      if (mb == 0ull && threadIdx.x == 0) atomicOr(flags, 2u);  // nothing near the bracket's lower edge was seen exactly: redo
      v0 = dkey_inv(mb);
    } else {
      if (!lower_known) {
        __syncthreads();
        unsigned long long c = 0ull, mx = 0ull;
        for (long long i = threadIdx.x; i < (long long)n; i += blockDim.x) {
          const unsigned long long k = getkey(i);
          if (k < k1) { ++c; mx = k > mx ? k : mx; }
        }
        if (c) { atomicAdd(&lt_count, c); atomicMax(&lt_max, mx); }
        __syncthreads();
      }
      v0 = (lt_count == (unsigned long long)r0) ? dkey_inv(lt_max) : v1;
    }
    res = 0.5 * (sqrt(v0) + sqrt(v1));
  }
  if (threadIdx.x == 0) {
    nf[col] = res;
    const double t2 = (res * gain) * (res * gain);
    const double g2 = gain * gain;
    if (!(t2 >= dkey_inv(lo) * g2 * (1.0 - 1e-10) && t2 <= dkey_inv(hi) * g2 * (1.0 + 1e-10))) atomicOr(flags, 8u);
  }
}

// classify the listed samples now that the thresholds are known: set their bits in the masks
__global__ void __launch_bounds__(256) pdw_patch_kernel(const float2* y, int M, const double* thr,
                                                        const unsigned long long* undecided, const unsigned* und_n,
                                                        unsigned long long* f0, unsigned long long* f1) {
  const unsigned n = *und_n < (unsigned)kUndecided ? *und_n : (unsigned)kUndecided;
  for (unsigned u = blockIdx.x * 256 + threadIdx.x; u < n; u += gridDim.x * 256) {
    const unsigned long long idx = undecided[u];
    const long long row = (long long)(idx / (unsigned long long)M);
    const int col = (int)(idx % (unsigned long long)M);
    const double m = mag_of(y[idx]), t = thr[col];
    const unsigned long long bit = 1ull << (row & 63);
    if (m >= t) atomicOr(&f0[(row >> 6) * M + col], bit);
    if (m > t) atomicOr(&f1[(row >> 6) * M + col], bit);
  }
}

// ---------------------------------------------------------------------------------
// edges
//
// The leading/trailing-edge state machine (create_pdws_channelized.m:85-135, create_pdws.m:54-105) is a
// 2-state automaton driven by two comparisons per sample: inactive -> active on mag >= lead (:87 / :57),
// active stays active while mag > trail (:94 / :63; the channelized script has lead == trail).  One pass
// over the data records the two comparison bits per sample (64 samples per word); everything after that
// -- tile summaries, the scan, the edge lists -- works on the bit masks, 1/64 of the data.

constexpr int kTileWords = kTile / 64;  // smallest tile, in 64-sample words

// Per-sample transition functions of one word: sample i maps state s to (s ? f1 : f0) bit i.  Returns the
// prefix compositions: bit i of p0 / p1 = state after sample i when the word is entered inactive / active
// (Kogge-Stone over function composition; bit 0 is the earliest sample).
__device__ __forceinline__ void word_scan(unsigned long long f0, unsigned long long f1, unsigned long long& p0,
                                          unsigned long long& p1) {
  p0 = f0;
  p1 = f1;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned long long e0 = p0 << d, e1 = (p1 << d) | ((1ull << d) - 1ull);  // earlier span; identity shifted in
    const unsigned long long n0 = (e0 & p1) | (~e0 & p0), n1 = (e1 & p1) | (~e1 & p0);
    p0 = n0;
    p1 = n1;
  }
}

// comparison masks of the F x M matrix, laid out [word][channel].  grid = (column groups, word groups of
// 4): one word (64 frames) per wave, lane = channel.  Frames past F are the identity (f0 = 0, f1 = 1).
// only_if != nullptr: run only when *only_if has bit 4 or 8 set (the bracket pass's provisional masks are unusable)
__global__ void __launch_bounds__(256) pdw_mask_kernel(const float2* y, long long F, int M, const double* thr,
                                                       unsigned long long* f0, unsigned long long* f1, long long words,
                                                       const unsigned* only_if) {
  if (only_if && (*only_if & 12u) == 0u) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  if (col >= M) return;
  const double t = thr[col];
  for (long long w = (long long)blockIdx.y * 4 + wave; w < words; w += 4ll * gridDim.y) {  // grid-stride over the words
    const long long r0 = w * 64;
    unsigned long long a = 0ull, b = 0ull;
    if (r0 + 64 <= F) {
#pragma unroll 8
      for (int i = 0; i < 64; ++i) {
        const double m = mag_of(y[(r0 + i) * M + col]);
        a |= (unsigned long long)(m >= t) << i;
        b |= (unsigned long long)(m > t) << i;
      }
    } else {
      for (int i = 0; i < 64; ++i) {
        if (r0 + i < F) {
          const double m = mag_of(y[(r0 + i) * M + col]);
          a |= (unsigned long long)(m >= t) << i;
          b |= (unsigned long long)(m > t) << i;
        } else {
          b |= 1ull << i;
        }
      }
    }
    f0[w * M + col] = a;
    f1[w * M + col] = b;
  }
}

// tile summaries for BOTH incoming states: fn[tile][col] = f(0) | f(1) << 1 and the edge counts of either
// trajectory, cnt[tile][col] = (starts from 0, ends from 0, starts from 1, ends from 1): with both in
// hand nothing has to be recounted once the scan has told which state each tile really starts in.
// One thread per (tile, channel), channel fastest.
__global__ void __launch_bounds__(256) pdw_tilefn_kernel(const unsigned long long* f0, const unsigned long long* f1, int M,
                                                         long long ntiles, int tile_words, unsigned char* fn, ushort4* cnt) {
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  if (g >= ntiles * M) return;
  const int col = (int)(g % M);
  const long long tile = g / M;
  int s0 = 0, s1 = 1;
  unsigned a0 = 0, e0 = 0, a1 = 0, e1 = 0;
  for (int j = 0; j < tile_words; ++j) {
    const long long w = tile * tile_words + j;
    unsigned long long p0, p1;
    word_scan(f0[w * M + col], f1[w * M + col], p0, p1);
    const unsigned long long S0 = s0 ? p1 : p0, S1 = s1 ? p1 : p0;
    const unsigned long long P0 = (S0 << 1) | (unsigned long long)s0, P1 = (S1 << 1) | (unsigned long long)s1;
    a0 += (unsigned)__popcll(S0 & ~P0); e0 += (unsigned)__popcll(~S0 & P0);
    a1 += (unsigned)__popcll(S1 & ~P1); e1 += (unsigned)__popcll(~S1 & P1);
    s0 = (int)(S0 >> 63); s1 = (int)(S1 >> 63);
  }
  fn[g] = (unsigned char)(s0 | (s1 << 1));
  cnt[g] = make_ushort4((unsigned short)a0, (unsigned short)e0, (unsigned short)a1, (unsigned short)e1);
}

// per column: incoming state of every tile, then the exclusive prefix of the edge counts of the
// trajectory each tile really follows, and the column totals.  One workgroup of BT threads per column
// (one wave when there are many columns, 16 waves for the one-column raw stream): each thread owns a
// contiguous segment of tiles; transition functions, then counts, are scanned across the workgroup
// (thread order = time order: shuffles inside a wave, wave totals through LDS) and each thread replays
// its segment.
__device__ __forceinline__ int compose_fn(int first, int then) {  // h(s) = then(first(s)), 2-bit encodings
  return ((then >> (first & 1)) & 1) | (((then >> ((first >> 1) & 1)) & 1) << 1);
}

template <int BT>
__global__ void __launch_bounds__(BT) pdw_tilescan_kernel(int M, long long ntiles, const unsigned char* fn,
                                                          const ushort4* cnt, unsigned char* state_in,
                                                          unsigned long long* off_s, unsigned long long* off_e,
                                                          unsigned long long* tot_s, unsigned long long* tot_e) {
  constexpr int NW = BT / 64;
  __shared__ int wave_fn[NW];
  __shared__ unsigned long long wave_a[NW], wave_b[NW];
  const int col = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long per = (ntiles + BT - 1) / BT;
  const long long t0 = (long long)tid * per < ntiles ? (long long)tid * per : ntiles;
  const long long t1 = (t0 + per < ntiles) ? t0 + per : ntiles;
  int f = 0x2;  // identity: f(0)=0, f(1)=1  -> bits (f0 | f1<<1) = 0b10
  {
    long long t = t0;
    for (; t + 8 <= t1; t += 8) {  // eight loads in flight: the chain through f is cheap, the strided bytes are not
      int g[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) g[u] = fn[(t + u) * M + col];
#pragma unroll
      for (int u = 0; u < 8; ++u) f = compose_fn(f, g[u]);
    }
    for (; t < t1; ++t) f = compose_fn(f, fn[t * M + col]);
  }
  // inclusive scan of function composition across the wave, then across waves
  int inc = f;
  for (int d = 1; d < 64; d <<= 1) {
    const int prev = __shfl_up(inc, d);
    if (lane >= d) inc = compose_fn(prev, inc);
  }
  int exc = __shfl_up(inc, 1);
  if (lane == 0) exc = 0x2;
  if (NW > 1) {
    if (lane == 63) wave_fn[wave] = inc;
    __syncthreads();
    int before = 0x2;
    for (int w = 0; w < wave; ++w) before = compose_fn(before, wave_fn[w]);
    exc = compose_fn(before, exc);
  }
  const int s_in = exc & 1;  // state entering my segment when the stream starts inactive: exc(0)
  int s = s_in;
  unsigned long long a = 0, b = 0;
  {
    auto step = [&](long long t, ushort4 c, int g) {
      state_in[t * M + col] = (unsigned char)s;
      a += s ? c.z : c.x;
      b += s ? c.w : c.y;
      s = (g >> s) & 1;
    };
    long long t = t0;
    for (; t + 8 <= t1; t += 8) {
      ushort4 c[8];
      int g[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { c[u] = cnt[(t + u) * M + col]; g[u] = fn[(t + u) * M + col]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) step(t + u, c[u], g[u]);
    }
    for (; t < t1; ++t) step(t, cnt[t * M + col], fn[t * M + col]);
  }
  unsigned long long ia = a, ib = b;
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned long long pa = __shfl_up(ia, d), pb = __shfl_up(ib, d);
    if (lane >= d) { ia += pa; ib += pb; }
  }
  if (NW > 1) {
    if (lane == 63) { wave_a[wave] = ia; wave_b[wave] = ib; }
    __syncthreads();
    unsigned long long ba = 0, bb = 0;
    for (int w = 0; w < wave; ++w) { ba += wave_a[w]; bb += wave_b[w]; }
    ia += ba; ib += bb;
  }
  unsigned long long ea = ia - a, eb = ib - b;  // exclusive
  s = s_in;
  {
    auto step = [&](long long t, ushort4 c, int g) {
      off_s[t * M + col] = ea; off_e[t * M + col] = eb;
      ea += s ? c.z : c.x;
      eb += s ? c.w : c.y;
      s = (g >> s) & 1;
    };
    long long t = t0;
    for (; t + 8 <= t1; t += 8) {
      ushort4 c[8];
      int g[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { c[u] = cnt[(t + u) * M + col]; g[u] = fn[(t + u) * M + col]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) step(t + u, c[u], g[u]);
    }
    for (; t < t1; ++t) step(t, cnt[t * M + col], fn[t * M + col]);
  }
  if (tid == BT - 1) { tot_s[col] = ia; tot_e[col] = ib; }
}

// replay a tile from its incoming state and write the leading / trailing edge sample indices
__global__ void __launch_bounds__(256) pdw_edges_kernel(const unsigned long long* f0, const unsigned long long* f1, int M,
                                                        long long ntiles, int tile_words, const unsigned char* state_in,
                                                        const unsigned long long* off_s, const unsigned long long* off_e,
                                                        long long* starts, long long* ends) {
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  if (g >= ntiles * M) return;
  const int col = (int)(g % M);
  const long long tile = g / M;
  int s = state_in[g];
  unsigned long long os = off_s[g], oe = off_e[g];
  for (int j = 0; j < tile_words; ++j) {
    const long long w = tile * tile_words + j;
    unsigned long long p0, p1;
    word_scan(f0[w * M + col], f1[w * M + col], p0, p1);
    const unsigned long long S = s ? p1 : p0, P = (S << 1) | (unsigned long long)s;
    unsigned long long up = S & ~P, down = ~S & P;
    while (up) { starts[os++] = w * 64 + (__ffsll((long long)up) - 1); up &= up - 1; }
    while (down) { ends[oe++] = w * 64 + (__ffsll((long long)down) - 1); down &= down - 1; }
    s = (int)(S >> 63);
  }
}

// One-column streams (the raw recorder stream) with long tiles: a thread per tile would be 16 384 threads walking 256
// words each.  A WAVE per tile instead: lane l summarises words [l wpl, (l + 1) wpl) exactly as pdw_tilefn_kernel
// summarises a tile (function + edge counts for both incoming states), the lanes' functions are scanned by composition
// (shuffles), which gives every lane the state it is entered in on either trajectory, and the counts add up.
__device__ __forceinline__ void lane_summary(const unsigned long long* f0, const unsigned long long* f1, long long w0, int wpl,
                                             int& fn, unsigned (&c)[4]) {
  int s0 = 0, s1 = 1;
  c[0] = c[1] = c[2] = c[3] = 0u;
  for (int j = 0; j < wpl; ++j) {
    unsigned long long p0, p1;
    word_scan(f0[w0 + j], f1[w0 + j], p0, p1);
    const unsigned long long S0 = s0 ? p1 : p0, S1 = s1 ? p1 : p0;
    const unsigned long long P0 = (S0 << 1) | (unsigned long long)s0, P1 = (S1 << 1) | (unsigned long long)s1;
    c[0] += (unsigned)__popcll(S0 & ~P0); c[1] += (unsigned)__popcll(~S0 & P0);
    c[2] += (unsigned)__popcll(S1 & ~P1); c[3] += (unsigned)__popcll(~S1 & P1);
    s0 = (int)(S0 >> 63); s1 = (int)(S1 >> 63);
  }
  fn = s0 | (s1 << 1);
}

// exclusive scan of the lanes' functions by composition: the function that maps the tile's incoming state to the state
// this lane is entered in; `total` = all 64 lanes composed
__device__ __forceinline__ int lane_prefix_fn(int fn, int lane, int& total) {
  int inc = fn;
  for (int d = 1; d < 64; d <<= 1) {
    const int prev = __shfl_up(inc, d);
    if (lane >= d) inc = compose_fn(prev, inc);
  }
  total = __shfl(inc, 63);
  const int exc = __shfl_up(inc, 1);
  return lane == 0 ? 0x2 : exc;  // identity for the first lane
}

__global__ void __launch_bounds__(256) pdw_tilefn_wave_kernel(const unsigned long long* f0, const unsigned long long* f1,
                                                              long long ntiles, int tile_words, unsigned char* fn,
                                                              ushort4* cnt) {
  const int lane = threadIdx.x & 63;
  const long long tile = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile >= ntiles) return;
  const int wpl = tile_words / 64;
  int g;
  unsigned c[4];
  lane_summary(f0, f1, tile * tile_words + (long long)lane * wpl, wpl, g, c);
  int total;
  const int pre = lane_prefix_fn(g, lane, total);
  const int in0 = pre & 1, in1 = (pre >> 1) & 1;  // the state this lane is entered in when the tile is entered in 0 / 1
  unsigned a0 = in0 ? c[2] : c[0], e0 = in0 ? c[3] : c[1], a1 = in1 ? c[2] : c[0], e1 = in1 ? c[3] : c[1];
  for (int d = 32; d > 0; d >>= 1) {
    a0 += __shfl_xor(a0, d); e0 += __shfl_xor(e0, d);
    a1 += __shfl_xor(a1, d); e1 += __shfl_xor(e1, d);
  }
  if (lane == 0) {
    fn[tile] = (unsigned char)total;
    cnt[tile] = make_ushort4((unsigned short)a0, (unsigned short)e0, (unsigned short)a1, (unsigned short)e1);
  }
}

__global__ void __launch_bounds__(256) pdw_edges_wave_kernel(const unsigned long long* f0, const unsigned long long* f1,
                                                             long long ntiles, int tile_words, const unsigned char* state_in,
                                                             const unsigned long long* off_s, const unsigned long long* off_e,
                                                             long long* starts, long long* ends) {
  const int lane = threadIdx.x & 63;
  const long long tile = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile >= ntiles) return;
  const int wpl = tile_words / 64;
  const long long w0 = tile * tile_words + (long long)lane * wpl;
  int g;
  unsigned c[4];
  lane_summary(f0, f1, w0, wpl, g, c);
  int total;
  const int pre = lane_prefix_fn(g, lane, total);
  int s = (pre >> (int)state_in[tile]) & 1;  // the state this lane is really entered in
  const unsigned ns = s ? c[2] : c[0], ne = s ? c[3] : c[1];
  unsigned is = ns, ie = ne;  // inclusive prefix sums over the lanes
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned ps = __shfl_up(is, d), pe = __shfl_up(ie, d);
    if (lane >= d) { is += ps; ie += pe; }
  }
  unsigned long long os = off_s[tile] + (is - ns), oe = off_e[tile] + (ie - ne);
  for (int j = 0; j < wpl; ++j) {
    const long long w = w0 + j;
    unsigned long long p0, p1;
    word_scan(f0[w], f1[w], p0, p1);
    const unsigned long long S = s ? p1 : p0, P = (S << 1) | (unsigned long long)s;
    unsigned long long up = S & ~P, down = ~S & P;
    while (up) { starts[os++] = w * 64 + (__ffsll((long long)up) - 1); up &= up - 1; }
    while (down) { ends[oe++] = w * 64 + (__ffsll((long long)down) - 1); down &= down - 1; }
    s = (int)(S >> 63);
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
// sample sources: where a (sample index, channel) pair finds its complex value

struct ChanSrc {  // F x M channelizer output, frame-major complex64
  static constexpr int kCache = kPulseCache;
  static constexpr int kThreads = 64;  // pulses are tens of frames and a pulse's workgroup is a chain of memory round trips: many small workgroups per CU
  const float2* y;
  int M;
  __device__ __forceinline__ double mag(long long i, int col) const { return mag_of(y[i * M + col]); }
  __device__ __forceinline__ double phase(long long i, int col) const { return phase_deg(y[i * M + col]); }
  __device__ __forceinline__ bool saturated(long long i, int col) const {
    const float2 v = y[i * M + col];
    return (fabs((double)v.x) >= 0.9999) || (fabs((double)v.y) >= 0.9999);
  }
};

// the raw recorder stream (create_pdws.m:30-33): x = (I + jQ) / 2^(bit_width-1), one column.
// |x|^2 orders like I^2 + Q^2, which is an exact integer for the integer formats.
template <int FMT>
struct RawSrc {
  static constexpr int kCache = kPulseCacheRaw;
  static constexpr int kThreads = 512;  // 256: 0.84 ms for 4794 pulses of 5600 samples, 512: 0.66, 1024: 1.09 (one workgroup per CU)
  const void* p;
  double inv_scale;  // 2^-(bit_width-1); 1 for cf32
  __device__ __forceinline__ void reim(long long i, double& re, double& im) const {
    if constexpr (FMT == PFB_FMT_INT8_IQ) {
      const char2 v = static_cast<const char2*>(p)[i];
      re = (double)v.x * inv_scale; im = (double)v.y * inv_scale;
    } else if constexpr (FMT == PFB_FMT_INT16_IQ) {
      const short2 v = static_cast<const short2*>(p)[i];
      re = (double)v.x * inv_scale; im = (double)v.y * inv_scale;
    } else {
      const float2 v = static_cast<const float2*>(p)[i];
      re = (double)v.x; im = (double)v.y;
    }
  }
  // order-preserving key of |x_i|^2 and the magnitude it stands for
  __device__ __forceinline__ unsigned long long key(long long i) const {
    if constexpr (FMT == PFB_FMT_INT8_IQ) {
      const char2 v = static_cast<const char2*>(p)[i];
      return (unsigned long long)((int)v.x * (int)v.x + (int)v.y * (int)v.y);
    } else if constexpr (FMT == PFB_FMT_INT16_IQ) {
      const short2 v = static_cast<const short2*>(p)[i];
      return (unsigned long long)((long long)v.x * v.x + (long long)v.y * v.y);
    } else {
      return dkey(mag2_of(static_cast<const float2*>(p)[i]));
    }
  }
  // keys of samples 4q .. 4q+3 from one 16-byte (int16), 8-byte (int8) or two 16-byte (cf32) loads; p 16-byte aligned
  __device__ __forceinline__ void key4(long long q, unsigned long long (&k)[4]) const {
    if constexpr (FMT == PFB_FMT_INT8_IQ) {
      const int2 w = static_cast<const int2*>(p)[q];
      const int v[2] = {w.x, w.y};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int half = (v[j >> 1] >> (16 * (j & 1))) & 0xffff;
        const int re = (int)(signed char)(half & 0xff), im = (int)(signed char)(half >> 8);
        k[j] = (unsigned long long)(re * re + im * im);
      }
    } else if constexpr (FMT == PFB_FMT_INT16_IQ) {
      const int4 w = static_cast<const int4*>(p)[q];
      const int v[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const long long re = (short)(v[j] & 0xffff), im = (short)(v[j] >> 16);
        k[j] = (unsigned long long)(re * re + im * im);
      }
    } else {
      const float4 a = static_cast<const float4*>(p)[2 * q], b = static_cast<const float4*>(p)[2 * q + 1];
      k[0] = dkey(mag2_of(make_float2(a.x, a.y))); k[1] = dkey(mag2_of(make_float2(a.z, a.w)));
      k[2] = dkey(mag2_of(make_float2(b.x, b.y))); k[3] = dkey(mag2_of(make_float2(b.z, b.w)));
    }
  }
  __device__ __forceinline__ double key_mag(unsigned long long k) const {
    if constexpr (FMT == PFB_FMT_CF32) return sqrt(dkey_inv(k));
    else return sqrt((double)k) * inv_scale;
  }
  __device__ __forceinline__ double mag(long long i, int) const { return key_mag(key(i)); }
  __device__ __forceinline__ double phase(long long i, int) const {
    double re, im;
    reim(i, re, im);
    return atan2(im, re) * kRadToDeg;
  }
  __device__ __forceinline__ bool saturated(long long i, int) const {
    double re, im;
    reim(i, re, im);
    return (fabs(re) >= 0.9999) || (fabs(im) >= 0.9999);
  }
};

// ---------------------------------------------------------------------------------
// per pulse

template <class Src, int CACHE, int THREADS>
__global__ void __launch_bounds__(THREADS) pdw_pulse_kernel(Src src, int M, const long long* starts, const long long* ends,
                                                        const unsigned long long* base_s, const unsigned long long* base_e,
                                                        const double* nf, const double* bin_freqs, double fs, double fc,
                                                        double t0, unsigned flags, pfb_pdw* out, unsigned long long capacity) {
  __shared__ unsigned hist[256];
  __shared__ unsigned long long pick[2];
  __shared__ double cache[CACHE];
  __shared__ double mid[2];
  __shared__ int sat_flag;
  // the bucket of block_median: its own array when cached pulses can be longer than the counting median handles,
  // otherwise the cache itself (block_median then only runs for pulses too long to be cached)
  __shared__ unsigned long long bucket_store[CACHE > kCountingMedian ? kCountingMedian : 1];
  static_assert(CACHE >= kCountingMedian, "the cache doubles as the bucket");
  unsigned long long* scratch = CACHE > kCountingMedian ? bucket_store : reinterpret_cast<unsigned long long*>(cache);
  const unsigned long long pid = blockIdx.x;
  if (pid >= capacity) return;
  // channel of this pulse: base_e is the exclusive prefix of tot_e over channels, so the pulse's channel is the largest
  // one whose base <= pid (every later base is > pid).  Wave 0 counts those bases 64 at a time -- one memory round trip
  // for M <= 64 lanes' worth, where a binary search would chain log2(M) of them.
  __shared__ int chan;
  if (threadIdx.x < 64) {
    int cnt_le = 0;
    for (int c0 = 0; c0 < M; c0 += 64) {
      const int c = c0 + (int)threadIdx.x;
      cnt_le += __popcll(__ballot(c < M && base_e[c] <= pid));
    }
    if (threadIdx.x == 0) chan = cnt_le - 1;  // base_e[0] = 0 <= pid
  }
  __syncthreads();
  const int b = chan;
  const unsigned long long k = pid - base_e[b];
  const long long toa = starts[base_s[b] + k], jj = ends[base_e[b] + k];
  const long long n = jj - toa + 1;
  const int pcol = (flags & PFB_PDW_MATLAB_QUIRKS) ? 0 : b;  // :114 phase(toa:jj) linear-indexes column 1
  if (threadIdx.x == 0) sat_flag = 0;
  __syncthreads();

  // :130-132 / create_pdws.m:100-102 saturation: samples strictly inside the pulse (the edge samples take
  // the other branches)
  int sat = 0;
  for (long long i = toa + 1 + threadIdx.x; i < jj; i += blockDim.x) sat |= src.saturated(i, b);
  if (sat) atomicOr(&sat_flag, 1);

  // :101 / :70 amplitude = median magnitude over toa..jj
  double amp;
  if (n <= CACHE) {
    for (long long i = threadIdx.x; i < n; i += blockDim.x) cache[i] = src.mag(toa + i, b);
    __syncthreads();
    amp = (n <= kCountingMedian) ? cached_median(cache, (int)n, mid)
                                 : block_median([&](long long i) { return cache[i]; }, n, hist, pick, scratch);
  } else {
    amp = block_median([&](long long i) { return src.mag(toa + i, b); }, n, hist, pick, scratch);
  }
  __syncthreads();

  // :114-117 / :83-86 median of the wrapped phase steps (degrees)
  auto dphi = [&](long long i) {
    double d = src.phase(toa + i + 1, pcol) - src.phase(toa + i, pcol);
    if (d < -180.0) d += 360.0;
    if (d > 180.0) d -= 360.0;
    return d;
  };
  double med;
  if (n <= CACHE) {  // one atan2 per sample: phases into the cache, steps into registers, steps back into the cache
    constexpr int PER = (CACHE + THREADS - 1) / THREADS;
    for (long long i = threadIdx.x; i < n; i += blockDim.x) cache[i] = src.phase(toa + i, pcol);
    __syncthreads();
    double step[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const long long i = threadIdx.x + (long long)j * THREADS;
      if (i < n - 1) {
        double d = cache[i + 1] - cache[i];
        if (d < -180.0) d += 360.0;
        if (d > 180.0) d -= 360.0;
        step[j] = d;
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const long long i = threadIdx.x + (long long)j * THREADS;
      if (i < n - 1) cache[i] = step[j];
    }
    __syncthreads();
    med = (n - 1 <= kCountingMedian) ? cached_median(cache, (int)(n - 1), mid)
                                     : block_median([&](long long i) { return cache[i]; }, n - 1, hist, pick, scratch);
  } else {
    med = block_median(dphi, n - 1, hist, pick, scratch);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    pfb_pdw o;
    o.toa = ((double)(toa + 1) / fs) + t0;            // :98 / :67 (1-based index)
    o.snr = 10.0 * log10(amp / nf[b]);                // :105 / :74
    o.pw = (double)(jj - toa) / fs;                   // :110 / :79
    // :80 binFreqs(bin), bin = column of the fftshift-ed matrix: the column's true centre frequency, or -- with
    // PFB_PDW_BINFREQ_UNSHIFTED -- the FFT-ordered list indexed by the shifted column (what the script computes if
    // MathWorks' centerFrequencies returns the unshifted list; unpinned).  bin_freqs is FFT-ordered; the raw script has no bins
    const double fbin = !bin_freqs ? 0.0
                        : (flags & PFB_PDW_BINFREQ_UNSHIFTED) ? bin_freqs[b] : bin_freqs[(b + (M + 1) / 2) % M];
    o.freq = (fc + fbin) + (fs / (360.0 / med));      // :122 / :91
    o.sat = sat_flag;
    o.bin = b;
    o.mag = amp;
    out[pid] = o;
  }
}

// ---------------------------------------------------------------------------------
// raw stream: noise floor and masks, time-parallel (one column, so lanes are consecutive samples)

constexpr int kRawBits = 11, kRawBins = 1 << kRawBits;

// one digit pass of the radix select of the stream's median |x|^2 key: digit = (key >> shift) & (bins-1)
// among keys whose bits above the digit equal `prefix`
// below_out (optional): also count the keys whose bits above the digit are SMALLER than prefix's -- what a pass that
// starts from a predicted prefix needs to turn the stream's rank into a rank inside the bucket
template <class Src, bool VEC>
__global__ void __launch_bounds__(256) pdw_raw_hist_kernel(Src src, long long n, int shift, unsigned bins_mask,
                                                           unsigned long long prefix, unsigned long long prefix_mask,
                                                           unsigned* hist, unsigned long long* below_out) {
  __shared__ unsigned h[kRawBins];
  for (int i = threadIdx.x; i < kRawBins; i += 256) h[i] = 0u;
  __syncthreads();
  unsigned long long nbelow = 0ull;
  const long long step = (long long)gridDim.x * 1024;
  for (long long i0 = (long long)blockIdx.x * 1024; i0 < n; i0 += step) {  // four samples per thread in flight
    unsigned long long k[4];
    bool in[4];
    if (VEC && i0 + 1024 <= n) {  // one wide load: samples i0 + 4 tid .. + 3
      src.key4((i0 >> 2) + threadIdx.x, k);
#pragma unroll
      for (int u = 0; u < 4; ++u) in[u] = true;
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long i = i0 + u * 256 + threadIdx.x;
        in[u] = i < n;
        k[u] = in[u] ? src.key(i) : 0ull;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      hist_add(h, (unsigned)(k[u] >> shift) & bins_mask, in[u] && ((k[u] & prefix_mask) == prefix));
      nbelow += (unsigned long long)(in[u] && (k[u] & prefix_mask) < prefix);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kRawBins; i += 256)
    if (h[i]) atomicAdd(&hist[i], h[i]);
  if (below_out) {
    for (int d = 32; d > 0; d >>= 1) nbelow += __shfl_xor(nbelow, d);
    if ((threadIdx.x & 63) == 0 && nbelow) atomicAdd(below_out, nbelow);
  }
}

// keys of ns samples spread over the stream (hashed positions, as the channelized sample): the host predicts the
// median's leading digits from them
template <class Src>
__global__ void __launch_bounds__(256) pdw_raw_sample_kernel(Src src, long long stride, int ns, unsigned long long* keys) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q < ns) keys[q] = src.key(sample_row(q, stride));
}

// number of keys below `pivot` and the largest of them (the lower middle value of an even-length median)
template <class Src, bool VEC>
__global__ void __launch_bounds__(256) pdw_raw_below_kernel(Src src, long long n, unsigned long long pivot,
                                                            unsigned long long* below, unsigned long long* max_below) {
  unsigned long long nb = 0ull, best = 0ull;
  bool any = false;
  const long long step = (long long)gridDim.x * 1024;
  for (long long i0 = (long long)blockIdx.x * 1024; i0 < n; i0 += step) {  // four samples per thread in flight
    unsigned long long k[4];
    if (VEC && i0 + 1024 <= n) {
      src.key4((i0 >> 2) + threadIdx.x, k);
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long i = i0 + u * 256 + threadIdx.x;
        k[u] = (i < n) ? src.key(i) : ~0ull;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (k[u] < pivot) { ++nb; best = (any && best > k[u]) ? best : k[u]; any = true; }
  }
  if (any) { atomicAdd(below, nb); atomicMax(max_below, best); }
}

// OR of x over the 16 lanes of a DPP row, left in every lane of the row (row_ror 1, 2, 4, 8)
__device__ __forceinline__ unsigned row_or(unsigned x) {
  x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x121, 0xf, 0xf, false);
  x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x122, 0xf, 0xf, false);
  x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x124, 0xf, 0xf, false);
  x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xf, 0xf, false);
  return x;
}

// comparison masks of the raw stream.  A wave covers 64 consecutive words: in step i every lane compares
// sample 64 * (w0 + i) + lane (one coalesced load), the wave votes, lane i keeps the word.  The magnitude is a
// monotone function of the sample's |x|^2 key (sqrt, then an exact power-of-two scale), so `mag >= lead` and
// `mag > trail` are comparisons of the KEY with the first key whose magnitude passes -- found by the host with the
// same float64 operations -- and the pass does no float64 arithmetic at all.  key_max: the largest key that is a
// number (an infinity passes every threshold, a NaN none, as with the magnitudes themselves).
template <class Src, bool VEC>
__global__ void __launch_bounds__(256) pdw_raw_mask_kernel(Src src, long long n, unsigned long long key_ge,
                                                           unsigned long long key_gt, unsigned long long key_max,
                                                           unsigned long long* f0, unsigned long long* f1, long long words) {
  const int lane = threadIdx.x & 63;
  const long long w0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
  if (w0 >= words) return;
  unsigned long long a = 0ull, b = 0ull;
  if (VEC && (w0 + 64) * 64 <= n) {
    // wide loads: in step u the wave reads 256 consecutive samples, four per lane (one 16-byte load for int16); a lane's
    // four comparison bits go to their place in the word its 16-lane row is building, the row ORs itself together
    // (DPP rotations), and lanes 4u .. 4u+3 keep the four finished words
#pragma unroll 1
    for (int u0 = 0; u0 < 16; u0 += 8) {  // eight loads in flight per lane
      unsigned long long k[8][4];
#pragma unroll
      for (int u = 0; u < 8; ++u) src.key4(((w0 * 64) >> 2) + (long long)(u0 + u) * 64 + lane, k[u]);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        unsigned na = 0u, nb = 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          na |= (unsigned)(k[u][j] >= key_ge && k[u][j] <= key_max) << j;
          nb |= (unsigned)(k[u][j] >= key_gt && k[u][j] <= key_max) << j;
        }
        // lanes 0-7 of a row fill the word's low half, lanes 8-15 the high half; OR over the row by DPP rotations
        const int sh = 4 * (lane & 7);
        const bool upper = (lane & 8) != 0;
        const unsigned a_lo = row_or(upper ? 0u : na << sh), a_hi = row_or(upper ? na << sh : 0u);
        const unsigned b_lo = row_or(upper ? 0u : nb << sh), b_hi = row_or(upper ? nb << sh : 0u);
#pragma unroll
        for (int j = 0; j < 4; ++j) {  // row j holds word 4 (u0 + u) + j
          const unsigned long long wa = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)a_hi, 16 * j) << 32) |
                                        (unsigned)__builtin_amdgcn_readlane((int)a_lo, 16 * j);
          const unsigned long long wb = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)b_hi, 16 * j) << 32) |
                                        (unsigned)__builtin_amdgcn_readlane((int)b_lo, 16 * j);
          if (lane == 4 * (u0 + u) + j) { a = wa; b = wb; }
        }
      }
    }
  } else {
    for (int i0 = 0; i0 < 64; i0 += 16) {  // sixteen loads in flight per lane
      bool ge[16], gt[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const long long sidx = (w0 + i0 + u) * 64 + lane;
        ge[u] = false; gt[u] = true;  // past the end: identity
        if (sidx < n) {
          const unsigned long long k = src.key(sidx);
          ge[u] = k >= key_ge && k <= key_max;
          gt[u] = k >= key_gt && k <= key_max;
        }
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const unsigned long long wa = __ballot(ge[u]), wb = __ballot(gt[u]);
        if (lane == i0 + u) { a = wa; b = wb; }
      }
    }
  }
  if (w0 + lane < words) { f0[w0 + lane] = a; f1[w0 + lane] = b; }
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

// small pinned host block per device: the words that cross the bus in the middle of an extraction (edge totals, flags,
// medians down; column bases up) move by DMA instead of through the runtime's pageable-copy staging
struct HostPin {
  char* p = nullptr;
  size_t cap = 0;
};
HostPin g_pin[kMaxDevices];
hipError_t pin_reserve(HostPin& h, size_t bytes) {
  if (bytes <= h.cap) return hipSuccess;
  if (h.p) (void)hipHostFree(h.p);
  h.p = nullptr;
  h.cap = 0;
  const hipError_t e = hipHostMalloc((void**)&h.p, bytes, hipHostMallocDefault);
  if (e == hipSuccess) h.cap = bytes;
  return e;
}

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
  return pfb::abi_guard([&] {
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
    if (g_pin[d].p) {
      (void)hipSetDevice(d);
      (void)hipHostFree(g_pin[d].p);
      g_pin[d] = HostPin{};
    }
  }
  if (prev >= 0) (void)hipSetDevice(prev);
  (void)hipGetLastError();
  return (int)PFB_OK;
  });
}

// host twin of dkey_inv (the raw cf32 noise floor is finished on the host)
static double dkey_inv_host(unsigned long long k) {
  const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  double d;
  std::memcpy(&d, &b, sizeof d);
  return d;
}

namespace {

// device buffers of the edge stage, all inside arena 0
struct EdgeStage {
  unsigned long long *f0, *f1, *off_s, *off_e, *tot, *base;
  unsigned char *fn, *state;
  ushort4* cnt;
  double *nf, *binf;  // binf == nullptr: no per-channel centre frequencies (raw stream)
};

size_t edge_stage_bytes(long long words, long long ntiles, uint32_t M) {
  const size_t wm = (size_t)words * M, tm = (size_t)ntiles * M;
  return 2 * padded(wm * sizeof(unsigned long long)) + 2 * padded(tm * sizeof(unsigned long long)) +
         2 * padded(2 * (size_t)M * sizeof(unsigned long long)) + 2 * padded(tm) + padded(tm * sizeof(ushort4)) +
         2 * padded(M * sizeof(double));
}

EdgeStage take_edge_stage(Arena& ws, long long words, long long ntiles, uint32_t M, bool with_binf) {
  const size_t wm = (size_t)words * M, tm = (size_t)ntiles * M;
  EdgeStage e{};
  e.f0 = take<unsigned long long>(ws, wm);
  e.f1 = take<unsigned long long>(ws, wm);
  e.off_s = take<unsigned long long>(ws, tm);
  e.off_e = take<unsigned long long>(ws, tm);
  e.tot = take<unsigned long long>(ws, 2 * (size_t)M);
  e.base = take<unsigned long long>(ws, 2 * (size_t)M);
  e.fn = take<unsigned char>(ws, tm);
  e.state = take<unsigned char>(ws, tm);
  e.cnt = take<ushort4>(ws, tm);
  e.nf = take<double>(ws, M);
  double* binf = take<double>(ws, M);
  e.binf = with_binf ? binf : nullptr;
  return e;
}

// masks (e.f0, e.f1) and noise floors (e.nf) are on the device: tile summaries, scan, edge lists, one
// workgroup per pulse, PDWs back to the host.
// d_check / h_check / h_nf (optional): flags of an optimistic noise-floor pass and its medians, fetched with the edge
// totals in the one sync; if the flags say the medians are not valid (bits 1 | 2) the function stops there and
// returns kRedo so that the caller can take the slow path and call again.
constexpr int kRedo = 1;
template <class Src>
int edges_and_pulses(Src src, int Mi, long long ntiles, int tile_words, const EdgeStage& e, Arena& ws2, double fs, double fc, double t0,
                     unsigned flags, pfb_pdw* out, uint64_t capacity, uint64_t* count, hipStream_t st,
                     const unsigned* d_check = nullptr, unsigned* h_check = nullptr, double* h_nf = nullptr) {
  int rc = PFB_OK;
  const uint32_t M = (uint32_t)Mi;
  const size_t tm = (size_t)ntiles * M;
  // pinned: [tot 2M u64 | base 2M u64 | nf M f64 | flags u32]
  int dev = 0;
  (void)hipGetDevice(&dev);
  HostPin& pin = g_pin[dev];
  unsigned long long *h_tot = nullptr, *h_base = nullptr;
  double* p_nf = nullptr;
  unsigned* p_check = nullptr;
  unsigned long long total_s = 0, total_e = 0;
  const unsigned tblocks = (unsigned)((tm + 255) / 256);
  const bool wave_tiles = Mi == 1 && tile_words >= 64 && tile_words % 64 == 0;  // one column, long tiles: a wave per tile
  PDW_TRY(pin_reserve(pin, (5 * (size_t)M + 1) * sizeof(unsigned long long)));
  h_tot = reinterpret_cast<unsigned long long*>(pin.p);
  h_base = h_tot + 2 * (size_t)M;
  p_nf = reinterpret_cast<double*>(h_base + 2 * (size_t)M);
  p_check = reinterpret_cast<unsigned*>(p_nf + M);
  if (wave_tiles) {
    hipLaunchKernelGGL(pdw_tilefn_wave_kernel, dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, st, (const unsigned long long*)e.f0,
                       (const unsigned long long*)e.f1, ntiles, tile_words, e.fn, e.cnt);
  } else {
    hipLaunchKernelGGL(pdw_tilefn_kernel, dim3(tblocks), dim3(256), 0, st, (const unsigned long long*)e.f0,
                       (const unsigned long long*)e.f1, Mi, ntiles, tile_words, e.fn, e.cnt);
  }
  if (Mi >= 32 && ntiles < 2048) {
    hipLaunchKernelGGL(pdw_tilescan_kernel<64>, dim3(Mi), dim3(64), 0, st, Mi, ntiles, (const unsigned char*)e.fn,
                       (const ushort4*)e.cnt, e.state, e.off_s, e.off_e, e.tot, e.tot + M);
  } else {  // few columns or many tiles per column: the parallelism has to come from time
    hipLaunchKernelGGL(pdw_tilescan_kernel<1024>, dim3(Mi), dim3(1024), 0, st, Mi, ntiles, (const unsigned char*)e.fn,
                       (const ushort4*)e.cnt, e.state, e.off_s, e.off_e, e.tot, e.tot + M);
  }
  PDW_TRY(hipGetLastError());
  PDW_TRY(hipMemcpyAsync(h_tot, e.tot, 2 * (size_t)M * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
  if (d_check) {
    PDW_TRY(hipMemcpyAsync(p_check, d_check, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    PDW_TRY(hipMemcpyAsync(p_nf, e.nf, (size_t)M * sizeof(double), hipMemcpyDeviceToHost, st));
  }
  PDW_TRY(hipStreamSynchronize(st));
  if (d_check) {
    *h_check = *p_check;
    std::memcpy(h_nf, p_nf, (size_t)M * sizeof(double));
    if (*h_check & 3u) return kRedo;
  }
  for (uint32_t b = 0; b < M; ++b) {  // channels outermost, like the reference's for bin = 1:M
    h_base[b] = total_s; h_base[M + b] = total_e;
    total_s += h_tot[b]; total_e += h_tot[M + b];
  }
  *count = total_e;  // a pulse still active at the end of the data produces no PDW (the trailing test never fires)
  if (total_e > 0) {
    const unsigned long long n_out = std::min<unsigned long long>(total_e, capacity);
    PDW_TRY(arena_reserve(ws2, padded((size_t)total_s * sizeof(long long)) + padded((size_t)total_e * sizeof(long long)) +
                                   padded((size_t)n_out * sizeof(pfb_pdw)) + kAlign));
    long long* d_starts = take<long long>(ws2, (size_t)total_s);
    long long* d_ends = take<long long>(ws2, (size_t)total_e);
    pfb_pdw* d_out = take<pfb_pdw>(ws2, (size_t)n_out);
    PDW_TRY(hipMemcpyAsync(e.base, h_base, 2 * (size_t)M * sizeof(unsigned long long), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(pdw_rebase_kernel, dim3(tblocks), dim3(256), 0, st, Mi, ntiles, e.off_s, e.off_e,
                       (const unsigned long long*)e.base, (const unsigned long long*)(e.base + M));
    if (wave_tiles) {
      hipLaunchKernelGGL(pdw_edges_wave_kernel, dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, st, (const unsigned long long*)e.f0,
                         (const unsigned long long*)e.f1, ntiles, tile_words, (const unsigned char*)e.state,
                         (const unsigned long long*)e.off_s, (const unsigned long long*)e.off_e, d_starts, d_ends);
    } else {
      hipLaunchKernelGGL(pdw_edges_kernel, dim3(tblocks), dim3(256), 0, st, (const unsigned long long*)e.f0,
                         (const unsigned long long*)e.f1, Mi, ntiles, tile_words, (const unsigned char*)e.state,
                         (const unsigned long long*)e.off_s, (const unsigned long long*)e.off_e, d_starts, d_ends);
    }
    if (n_out > 0) {
      hipLaunchKernelGGL((pdw_pulse_kernel<Src, Src::kCache, Src::kThreads>), dim3((unsigned)n_out), dim3(Src::kThreads), 0, st, src, Mi, (const long long*)d_starts,
                         (const long long*)d_ends, (const unsigned long long*)e.base, (const unsigned long long*)(e.base + M),
                         (const double*)e.nf, (const double*)e.binf, fs, fc, t0, flags, d_out, n_out);
      PDW_TRY(hipGetLastError());
      PDW_TRY(hipMemcpyAsync(out, d_out, (size_t)n_out * sizeof(pfb_pdw), hipMemcpyDeviceToHost, st));
    }
    PDW_TRY(hipStreamSynchronize(st));
  }
done:
  return rc;
}

// Tile length of the edge scan, in words.  The scan kernel walks a column's tiles with one workgroup (a strided, latency-
// bound walk), the tile kernels before and after it want >= 2^18 (tile, column) threads: at most 2^18 / M tiles per
// column, between 2048 and 16384 (measured at M = 128, 2^22 frames: scan + tile kernels 166 us at 8192 tiles per
// column, 100 us at 2048, 111 us at 1024).  The per-tile edge counts are 16-bit, which caps a tile at 2^16 samples.
int tile_words_for(long long samples, int M) {
  const long long w = (samples + 63) / 64;
  const long long max_tiles = std::min<long long>(16384, std::max<long long>(2048, (1ll << 18) / std::max(1, M)));
  int tw = kTileWords;
  while (tw < 1024 && w / tw > max_tiles) tw *= 2;
  return tw;
}

// RAII: select the device for the call, restore on exit
struct DeviceScope {
  int prev = -1;
  bool switched = false;
  int enter(int32_t device_id, int ndev) {
    (void)hipGetDevice(&prev);
    if (device_id >= 0 && device_id != prev) {
      if (device_id >= ndev || hipSetDevice(device_id) != hipSuccess) return PFB_ERR_BAD_ARG;
      switched = true;
    }
    return PFB_OK;
  }
  ~DeviceScope() {
    if (switched && prev >= 0) (void)hipSetDevice(prev);
  }
};

}  // namespace

static int pdw_extract_impl(const void* y_in, uint64_t frames, uint32_t M, uint32_t decimation, double fs_in,
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
  DeviceScope scope;
  if (scope.enter(device_id, ndev) != PFB_OK) return PFB_ERR_BAD_ARG;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= kMaxDevices) return PFB_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lock(g_ws_mutex);  // one extraction per process at a time shares the scratch
  Arena& ws = g_ws[dev][0];
  Arena& ws2 = g_ws[dev][1];
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  const long long F = (long long)frames;
  const int Mi = (int)M;
  const int tile_words = tile_words_for(F, Mi);
  const long long ntiles = (F + 64ll * tile_words - 1) / (64ll * tile_words);
  const long long words = ntiles * tile_words;  // whole tiles; the tail is identity-padded
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
  const long long key_ld = (ns + 63) / 64 * 64;  // sample keys per channel, padded to whole 256-byte lines

  const float2* d_y = nullptr;
  unsigned *d_hist, *d_bucket, *d_cand_n, *d_flags, *d_und_n, *d_keys;
  unsigned long long *d_prefix, *d_prefix_hi, *d_rank, *d_below, *d_maxbelow, *d_und;
  double *d_cand, *d_thr;
  EdgeStage e{};
  std::vector<unsigned> h_bucket(M);
  std::vector<unsigned long long> h_rank(M);
  std::vector<double> h_nf(M), h_binf(M);
  unsigned h_flags = 0;
  size_t zero_bytes = 0;
  FinishShared fin{};
  int passes = 0;
  const double gain = std::pow(10.0, snr_threshold_db / 10.0);  // :74-75 (dB applied to magnitude with /10)
  const int row_blocks = (int)std::min<long long>(1024, std::max<long long>(1, F / 256));
  const size_t cand_elems = std::max<size_t>((size_t)M * kCand, (size_t)M * cap);

  {
    size_t need = 0;
    if (mem == PFB_MEM_HOST) need += padded((size_t)F * M * sizeof(float2));
    if (flags & PFB_PDW_CHANNEL_MAJOR) need += padded((size_t)F * M * sizeof(float2));
    need += padded((size_t)M * 256 * sizeof(unsigned)) + 2 * padded(M * sizeof(unsigned)) + 2 * padded(sizeof(unsigned));
    need += 4 * padded(2 * M * sizeof(unsigned long long)) + padded((size_t)kUndecided * sizeof(unsigned long long));
    need += padded(cand_elems * sizeof(double)) + padded(M * sizeof(double));
    need += padded((size_t)M * kFinishBins * sizeof(unsigned)) + 3 * padded(M * sizeof(unsigned long long)) +
            padded((size_t)M * kFinishLds * sizeof(unsigned long long));  // the split candidate select
    if (sampled) need += padded((size_t)M * key_ld * sizeof(unsigned));
    need += edge_stage_bytes(words, ntiles, M);
    PDW_TRY(arena_reserve(ws, need));
  }
  if (mem == PFB_MEM_HOST) {
    float2* own = take<float2>(ws, (size_t)F * M);
    PDW_TRY(hipMemcpyAsync(own, y_in, (size_t)F * M * sizeof(float2), hipMemcpyHostToDevice, st));
    d_y = own;
  } else {
    d_y = static_cast<const float2*>(y_in);
  }
  if (flags & PFB_PDW_CHANNEL_MAJOR) {
    // MATLAB's own layout (M columns of F frames): the pipeline walks rows of M channels, so the matrix is
    // transposed once into scratch (64 x 64 tiles through LDS, 512-byte reads and writes)
    if (F >= (1ll << 31)) return PFB_ERR_UNSUPPORTED;
    float2* fm = take<float2>(ws, (size_t)F * M);
    PDW_TRY(pfb::launch_transpose_slab(d_y, (long long)M, (int)F, fm, (long long)M, 0, (int)sizeof(float2), st));
    d_y = fm;
  }
  d_hist = take<unsigned>(ws, (size_t)M * 256);  // [channel][digit] of the full select
  d_bucket = take<unsigned>(ws, (size_t)M);
  d_prefix = take<unsigned long long>(ws, 2 * (size_t)M);
  d_prefix_hi = d_prefix + M;
  d_rank = take<unsigned long long>(ws, (size_t)M);
  // zeroed together by one memset (consecutive in the arena): d_below .. d_und_n
  d_below = take<unsigned long long>(ws, (size_t)M);
  d_maxbelow = take<unsigned long long>(ws, (size_t)M);
  d_cand_n = take<unsigned>(ws, (size_t)M);
  d_flags = take<unsigned>(ws, 1);
  d_und_n = take<unsigned>(ws, 1);
  fin.hist = take<unsigned>(ws, (size_t)M * kFinishBins);
  fin.bucket_n = take<unsigned>(ws, (size_t)M);
  fin.lt_max = take<unsigned long long>(ws, (size_t)M);
  fin.ticket = take<unsigned>(ws, (size_t)M);
  zero_bytes = (size_t)(reinterpret_cast<char*>(fin.ticket + M) - reinterpret_cast<char*>(d_below));
  fin.bucket = take<unsigned long long>(ws, (size_t)M * kFinishLds);
  d_und = take<unsigned long long>(ws, (size_t)kUndecided);
  d_cand = take<double>(ws, cand_elems);
  d_thr = take<double>(ws, M);
  d_keys = sampled ? take<unsigned>(ws, (size_t)M * key_ld) : nullptr;
  e = take_edge_stage(ws, words, ntiles, M, true);

  pfb_center_frequencies(M, fs_in, h_binf.data());  // :42, before fs is decimated
  PDW_TRY(hipMemcpyAsync(e.binf, h_binf.data(), M * sizeof(double), hipMemcpyHostToDevice, st));

  // ---- noise floor (:73), sampled bracket first.  Everything up to the edge totals is queued without a host sync:
  // the sample (gathered once, selected per channel), the bracket pass (which also leaves provisional masks), the
  // candidate select, thresholds on the device, the patch of the unclassified samples (or a full mask pass if the
  // device finds the provisional masks unusable), tile summaries and scan.  The host reads flags, medians and totals
  // in one go.
  if (sampled) {
    PDW_TRY(hipMemsetAsync(d_below, 0, zero_bytes, st));
    hipLaunchKernelGGL(pdw_sample_gather_kernel, dim3(cgroups, (unsigned)((ns + 63) / 64)), dim3(256), 0, st, d_y, ns, stride, Mi,
                       d_keys, key_ld);
    hipLaunchKernelGGL(pdw_sample_select_kernel, dim3(Mi), dim3(1024), 0, st, (const unsigned*)d_keys, ns, key_ld,
                       (unsigned long long)std::max<long long>(0, ns / 2 - delta),
                       (unsigned long long)std::min<long long>(ns - 1, ns / 2 + delta), d_prefix, d_prefix_hi);
    {
      const int row_groups = (int)((words * 64 + kBracketRows - 1) / kBracketRows);
      // many short-lived workgroups (one or two row groups each) beat a few long-lived ones here: 0.83 vs 0.92 ms
      const int gy = std::max(1, std::min(row_groups, 32 * 256 / cgroups));
      // small banks: several rows per wave-load (lanes per row = the power of two that holds M)
      auto kern = Mi > 32 ? pdw_bracket_kernel<64> : Mi > 16 ? pdw_bracket_kernel<32> : Mi > 8 ? pdw_bracket_kernel<16> : pdw_bracket_kernel<8>;
      hipLaunchKernelGGL(kern, dim3(cgroups, gy), dim3(256), 0, st, d_y, F, Mi,
                         (const unsigned long long*)d_prefix, (const unsigned long long*)d_prefix_hi, gain * gain, d_cand, cap,
                         d_cand_n, d_below, d_maxbelow, e.f0, e.f1, words, d_und, d_und_n, d_flags, row_groups);
    }
    {
      // parts per channel: enough workgroups to fill the chip, no more than a part's share is worth (>= 8192 candidates)
      const int parts = (int)std::max<long long>(1, std::min<long long>(std::min<long long>(16, 512 / Mi + 1), (long long)(expect / 8192)));
      hipLaunchKernelGGL(pdw_finish_hist_kernel, dim3(Mi, parts), dim3(1024), 0, st, F, (const double*)d_cand, cap,
                         (const unsigned*)d_cand_n, (const unsigned long long*)d_below, (const unsigned long long*)d_prefix,
                         (const unsigned long long*)d_prefix_hi, fin);
      hipLaunchKernelGGL(pdw_bracket_finish_kernel, dim3(Mi, parts), dim3(1024), 0, st, F, (const double*)d_cand, cap,
                         (const unsigned*)d_cand_n, (const unsigned long long*)d_below, (const unsigned long long*)d_maxbelow,
                         (const unsigned long long*)d_prefix, (const unsigned long long*)d_prefix_hi, gain, e.nf, d_flags, fin);
    }
    hipLaunchKernelGGL(pdw_thr_kernel, dim3((Mi + 255) / 256), dim3(256), 0, st, (const double*)e.nf, gain, d_thr, Mi);
    hipLaunchKernelGGL(pdw_patch_kernel, dim3(64), dim3(256), 0, st, d_y, Mi, (const double*)d_thr,
                       (const unsigned long long*)d_und, (const unsigned*)d_und_n, e.f0, e.f1);
    // (almost always a no-op: few workgroups, each striding over the words when it does run)
    hipLaunchKernelGGL(pdw_mask_kernel, dim3(cgroups, (unsigned)std::min<long long>((words + 3) / 4, 8192 / cgroups + 1)), dim3(256), 0, st,
                       d_y, F, Mi, (const double*)d_thr, e.f0, e.f1, words, (const unsigned*)d_flags);
    PDW_TRY(hipGetLastError());
    rc = edges_and_pulses(ChanSrc{d_y, Mi}, Mi, ntiles, tile_words, e, ws2, fs, fc, sample_start_time, flags, out, capacity,
                          count, st, d_flags, &h_flags, h_nf.data());
    if (rc != kRedo) {
      g_pdw_path = h_flags == 0 ? 1 : 4;  // 4: flags 4 / 8 only spoiled the provisional masks, the device redid them
      if (rc == PFB_OK && noise_floor_out) std::memcpy(noise_floor_out, h_nf.data(), M * sizeof(double));
      goto done;
    }
    rc = PFB_OK;
  }
  g_pdw_path = sampled ? 3 : 2;
  {  // full radix select of rank F/2, then the exact finish
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
      hipLaunchKernelGGL(pdw_pick_kernel, dim3((Mi + 3) / 4), dim3(256), 0, st, Mi, passes, d_hist, d_prefix, d_rank,
                         d_bucket, d_below);
      ++passes;
      PDW_TRY(hipMemcpyAsync(h_bucket.data(), d_bucket, M * sizeof(unsigned), hipMemcpyDeviceToHost, st));
      PDW_TRY(hipStreamSynchronize(st));
      if (*std::max_element(h_bucket.begin(), h_bucket.end()) <= (unsigned)kCand) break;
    }
    hipLaunchKernelGGL(pdw_collect_kernel, dim3(cgroups, row_blocks), dim3(256), 0, st, d_y, F, Mi, passes, d_prefix, d_cand,
                       d_cand_n, d_maxbelow);
    hipLaunchKernelGGL(pdw_median_finish_kernel, dim3(Mi), dim3(256), 0, st, F, passes, d_cand, d_cand_n, d_prefix, d_rank,
                       d_maxbelow, e.nf);
    hipLaunchKernelGGL(pdw_thr_kernel, dim3((Mi + 255) / 256), dim3(256), 0, st, (const double*)e.nf, gain, d_thr, Mi);
    hipLaunchKernelGGL(pdw_mask_kernel, dim3(cgroups, (unsigned)std::min<long long>((words + 3) / 4, 65535)), dim3(256), 0, st, d_y, F, Mi,
                       (const double*)d_thr, e.f0, e.f1, words, (const unsigned*)nullptr);
    PDW_TRY(hipGetLastError());
    if (noise_floor_out) {
      PDW_TRY(hipMemcpyAsync(h_nf.data(), e.nf, M * sizeof(double), hipMemcpyDeviceToHost, st));
      PDW_TRY(hipStreamSynchronize(st));
      std::memcpy(noise_floor_out, h_nf.data(), M * sizeof(double));
    }
    rc = edges_and_pulses(ChanSrc{d_y, Mi}, Mi, ntiles, tile_words, e, ws2, fs, fc, sample_start_time, flags, out, capacity,
                          count, st);
  }

done:
  (void)hipStreamSynchronize(st);
  return rc;
}

// nothing thrown (std::vector / std::string / std::mutex inside the implementation) crosses the C ABI
extern "C" int pfb_pdw_extract(const void* y_in, uint64_t frames, uint32_t M, uint32_t decimation, double fs_in,
                               double fc, double sample_start_time, double snr_threshold_db, uint32_t flags,
                               pfb_pdw* out, uint64_t capacity, uint64_t* count, double* noise_floor_out, uint32_t mem,
                               int32_t device_id, void* hip_stream) {
  return pfb::abi_guard([&] {
    return pdw_extract_impl(y_in, frames, M, decimation, fs_in, fc, sample_start_time, snr_threshold_db, flags, out, capacity,
                            count, noise_floor_out, mem, device_id, hip_stream);
  });
}

// ---- raw stream (matlab/create_pdws.m:30-105) -------------------------------------------------------

namespace {

template <int FMT>
int extract_raw(const void* d_iq, long long n, double inv_scale, double fs, double fc, double t0, double lead_db,
                double trail_db, pfb_pdw* out, uint64_t capacity, uint64_t* count, double* noise_floor_out, Arena& ws,
                Arena& ws2, const EdgeStage& e, unsigned* d_hist, unsigned long long* d_pair, long long words,
                long long ntiles, int tile_words, hipStream_t st) {
  int rc = PFB_OK;
  const RawSrc<FMT> src{d_iq, inv_scale};
  const bool vec = (reinterpret_cast<uintptr_t>(d_iq) % 16) == 0;  // wide loads in the counting passes
  (void)ws;
  // ---- noise floor (:44): radix select of rank n/2 on the |x|^2 keys, 11-bit digits
  struct Pass { int shift, bits; };
  static const Pass kIntPasses[] = {{22, 11}, {11, 11}, {0, 11}};                            // keys < 2^33
  static const Pass kF32Passes[] = {{53, 11}, {42, 11}, {31, 11}, {20, 11}, {9, 11}, {0, 9}};  // 64-bit double keys
  const Pass* pass = (FMT == PFB_FMT_CF32) ? kF32Passes : kIntPasses;
  const int npass = (FMT == PFB_FMT_CF32) ? 6 : 3;
  const unsigned grid = (unsigned)std::min<long long>(4096, std::max<long long>(1, (n + 2047) / 2048));
  std::vector<unsigned> h_hist(kRawBins);
  unsigned long long prefix = 0ull, rank = (unsigned long long)(n / 2), h_pair[2] = {0ull, 0ull};
  double nf = 0.0, lead = 0.0, trail = 0.0;
  // The leading digits of the median are predictable: the keys of a few thousand samples spread over the stream bracket
  // it (5 sigma either side of the sample's middle), and the digits both bracket ends share are, almost surely, the
  // median's.  The select starts below them -- for noise-dominated int16 data the first two of the three passes see
  // every key in one bucket -- and the first pass it does run also counts the keys below the predicted bucket, which
  // both turns the rank into a rank inside the bucket and PROVES the prediction (the rank must fall inside); if it
  // does not, the select starts over from the top.
  int first_pass = 0;
  if (n >= (1ll << 22)) {
    constexpr int kNs = 4096;
    const long long stride = n / kNs;
    std::vector<unsigned long long> sk(kNs);
    unsigned long long* d_sk = reinterpret_cast<unsigned long long*>(d_hist + kRawBins);  // room behind the histogram (see the caller)
    hipLaunchKernelGGL(pdw_raw_sample_kernel<RawSrc<FMT>>, dim3(kNs / 256), dim3(256), 0, st, src, stride, kNs, d_sk);
    PDW_TRY(hipGetLastError());
    PDW_TRY(hipMemcpyAsync(sk.data(), d_sk, kNs * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    PDW_TRY(hipStreamSynchronize(st));
    const int delta = (int)std::ceil(2.5 * std::sqrt((double)kNs)) + 2;
    std::nth_element(sk.begin(), sk.begin() + (kNs / 2 - delta), sk.end());
    const unsigned long long k_lo = sk[kNs / 2 - delta];
    std::nth_element(sk.begin(), sk.begin() + (kNs / 2 + delta), sk.end());
    const unsigned long long k_hi = sk[kNs / 2 + delta];
    while (first_pass < npass - 1) {  // passes whose digit (and everything above) both bracket ends share
      const int sh = pass[first_pass].shift;
      if ((k_lo >> sh) != (k_hi >> sh)) break;
      ++first_pass;
    }
    if (first_pass > 0) prefix = k_lo & (~0ull << pass[first_pass - 1].shift);
  }
  for (int ps = first_pass; ps < npass; ++ps) {
    const int top = pass[ps].shift + pass[ps].bits;
    const unsigned long long pmask = top >= 64 ? 0ull : (~0ull << top);
    const bool check = first_pass > 0 && ps == first_pass;  // the first pass after a prediction
    PDW_TRY(hipMemsetAsync(d_hist, 0, kRawBins * sizeof(unsigned), st));
    if (check) PDW_TRY(hipMemsetAsync(d_pair, 0, sizeof(unsigned long long), st));
    if (vec) {
      hipLaunchKernelGGL((pdw_raw_hist_kernel<RawSrc<FMT>, true>), dim3(grid), dim3(256), 0, st, src, n, pass[ps].shift,
                         (1u << pass[ps].bits) - 1u, prefix, pmask, d_hist, check ? d_pair : nullptr);
    } else {
      hipLaunchKernelGGL((pdw_raw_hist_kernel<RawSrc<FMT>, false>), dim3(grid), dim3(256), 0, st, src, n, pass[ps].shift,
                         (1u << pass[ps].bits) - 1u, prefix, pmask, d_hist, check ? d_pair : nullptr);
    }
    PDW_TRY(hipGetLastError());
    PDW_TRY(hipMemcpyAsync(h_hist.data(), d_hist, kRawBins * sizeof(unsigned), hipMemcpyDeviceToHost, st));
    if (check) PDW_TRY(hipMemcpyAsync(h_pair, d_pair, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    PDW_TRY(hipStreamSynchronize(st));
    if (check) {
      unsigned long long in_bucket = 0;
      for (int d = 0; d < kRawBins; ++d) in_bucket += h_hist[d];
      const unsigned long long below_pred = h_pair[0];
      h_pair[0] = 0ull;
      if (below_pred > rank || rank - below_pred >= in_bucket) {  // the median is not in the predicted bucket: from the top
        first_pass = 0;
        prefix = 0ull;
        rank = (unsigned long long)(n / 2);
        ps = -1;
        continue;
      }
      rank -= below_pred;
    }
    unsigned long long cum = 0;
    int d = 0;
    const int last = (1 << pass[ps].bits) - 1;
    for (; d < last; ++d) {
      if (cum + h_hist[d] > rank) break;
      cum += h_hist[d];
    }
    prefix |= (unsigned long long)d << pass[ps].shift;
    rank -= cum;
  }
  {
    auto key_mag = [&](unsigned long long k) {
      return (FMT == PFB_FMT_CF32) ? std::sqrt(dkey_inv_host(k)) : std::sqrt((double)k) * inv_scale;
    };
    unsigned long long v0 = prefix;
    if ((n & 1) == 0 && rank == 0) {
      // even count and the pivot is the first of its value in the order: the lower middle value is the largest key below
      // it.  All of the pivot's bits are decided, so the last pass's histogram (h_hist: the lowest digit among the keys
      // that share every higher bit) usually names it -- the nearest occupied digit below the pivot's; only when that
      // bucket holds nothing smaller does the data have to be read once more.
      const int dl = (int)((prefix >> pass[npass - 1].shift) & ((1u << pass[npass - 1].bits) - 1u));
      int dn = dl - 1;
      while (dn >= 0 && h_hist[dn] == 0u) --dn;
      if (dn >= 0) {
        v0 = (prefix & ~((unsigned long long)((1u << pass[npass - 1].bits) - 1u) << pass[npass - 1].shift)) |
             ((unsigned long long)dn << pass[npass - 1].shift);
      } else {
        PDW_TRY(hipMemsetAsync(d_pair, 0, 2 * sizeof(unsigned long long), st));
        if (vec) {
          hipLaunchKernelGGL((pdw_raw_below_kernel<RawSrc<FMT>, true>), dim3(grid), dim3(256), 0, st, src, n, prefix, d_pair,
                             d_pair + 1);
        } else {
          hipLaunchKernelGGL((pdw_raw_below_kernel<RawSrc<FMT>, false>), dim3(grid), dim3(256), 0, st, src, n, prefix, d_pair,
                             d_pair + 1);
        }
        PDW_TRY(hipGetLastError());
        PDW_TRY(hipMemcpyAsync(h_pair, d_pair, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        PDW_TRY(hipStreamSynchronize(st));
        if (h_pair[0] == (unsigned long long)(n / 2)) v0 = h_pair[1];
      }
    }  // (rank > 0: the pivot's value repeats below the middle, v0 = the pivot)
    nf = (n & 1) ? key_mag(prefix) : 0.5 * (key_mag(v0) + key_mag(prefix));
    lead = nf * std::pow(10.0, lead_db / 10.0);    // :45-46
    trail = nf * std::pow(10.0, trail_db / 10.0);  // :47
    if (noise_floor_out) *noise_floor_out = nf;
  }
  PDW_TRY(hipMemcpyAsync(e.nf, &nf, sizeof(double), hipMemcpyHostToDevice, st));
  PDW_TRY(hipStreamSynchronize(st));  // nf lives on this stack frame
  // ---- edges (:54-105) and pulses.  The thresholds as keys: the first key whose magnitude is >= lead / > trail
  {
    auto key_mag = [&](unsigned long long k) {
      return (FMT == PFB_FMT_CF32) ? std::sqrt(dkey_inv_host(k)) : std::sqrt((double)k) * inv_scale;
    };
    const unsigned long long k_lo = (FMT == PFB_FMT_CF32) ? 0x8000000000000000ull : 0ull;               // |x|^2 = 0
    const unsigned long long k_hi = (FMT == PFB_FMT_CF32) ? 0xFFF0000000000000ull : (1ull << 33);       // +inf / above any sample
    auto first_key = [&](auto pred) {  // smallest key in [k_lo, k_hi] that passes, k_hi + 1 if none (pred is monotone)
      if (!pred(k_hi)) return k_hi + 1;
      unsigned long long lo = k_lo, hi = k_hi;  // invariant: pred(hi)
      while (lo < hi) {
        const unsigned long long mid = lo + (hi - lo) / 2;
        if (pred(mid)) hi = mid; else lo = mid + 1;
      }
      return lo;
    };
    const unsigned long long key_ge = first_key([&](unsigned long long k) { return key_mag(k) >= lead; });
    const unsigned long long key_gt = first_key([&](unsigned long long k) { return key_mag(k) > trail; });
    if (vec) {
      hipLaunchKernelGGL((pdw_raw_mask_kernel<RawSrc<FMT>, true>), dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st, src, n,
                         key_ge, key_gt, k_hi, e.f0, e.f1, words);
    } else {
      hipLaunchKernelGGL((pdw_raw_mask_kernel<RawSrc<FMT>, false>), dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st, src, n,
                         key_ge, key_gt, k_hi, e.f0, e.f1, words);
    }
  }
  PDW_TRY(hipGetLastError());
  rc = edges_and_pulses(src, 1, ntiles, tile_words, e, ws2, fs, fc, t0, 0u, out, capacity, count, st);
done:
  return rc;
}

}  // namespace

static int pdw_extract_raw_impl(const void* iq, uint64_t num_samples, uint32_t sample_format, uint32_t bit_width,
                                double fs, double fc, double sample_start_time, double snr_threshold_db,
                                double trailing_threshold_db, pfb_pdw* out, uint64_t capacity, uint64_t* count,
                                double* noise_floor_out, uint32_t mem, int32_t device_id, void* hip_stream) {
  if (!iq || !count || num_samples < 2 || sample_format > PFB_FMT_CF32 || mem > PFB_MEM_DEVICE || (capacity && !out))
    return PFB_ERR_BAD_ARG;
  if (sample_format != PFB_FMT_CF32 && (bit_width < 1 || bit_width > 16)) return PFB_ERR_BAD_ARG;
  if (!(trailing_threshold_db <= snr_threshold_db)) return PFB_ERR_BAD_ARG;  // the masks assume lead >= trail
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    (void)hipGetLastError();
    return PFB_ERR_NO_DEVICE;
  }
  DeviceScope scope;
  if (scope.enter(device_id, ndev) != PFB_OK) return PFB_ERR_BAD_ARG;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= kMaxDevices) return PFB_ERR_BAD_ARG;
  std::lock_guard<std::mutex> lock(g_ws_mutex);
  Arena& ws = g_ws[dev][0];
  Arena& ws2 = g_ws[dev][1];
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  const long long n = (long long)num_samples;
  const int tile_words = tile_words_for(n, 1);
  const long long ntiles = (n + 64ll * tile_words - 1) / (64ll * tile_words);
  const long long words = ntiles * tile_words;
  const size_t bps = sample_format == PFB_FMT_INT8_IQ ? 2 : sample_format == PFB_FMT_INT16_IQ ? 4 : 8;
  const double inv_scale = sample_format == PFB_FMT_CF32 ? 1.0 : std::ldexp(1.0, -((int)bit_width - 1));
  int rc = PFB_OK;
  const void* d_iq = iq;
  unsigned* d_hist;
  unsigned long long* d_pair;
  EdgeStage e{};
  constexpr size_t kHistBytes = kRawBins * sizeof(unsigned) + 4096 * sizeof(unsigned long long);  // histogram + the sample's keys
  PDW_TRY(arena_reserve(ws, (mem == PFB_MEM_HOST ? padded((size_t)n * bps) : 0) + padded(kHistBytes) +
                                padded(2 * sizeof(unsigned long long)) + edge_stage_bytes(words, ntiles, 1)));
  if (mem == PFB_MEM_HOST) {
    char* own = take<char>(ws, (size_t)n * bps);
    PDW_TRY(hipMemcpyAsync(own, iq, (size_t)n * bps, hipMemcpyHostToDevice, st));
    d_iq = own;
  }
  d_hist = take<unsigned>(ws, kHistBytes / sizeof(unsigned));
  d_pair = take<unsigned long long>(ws, 2);
  e = take_edge_stage(ws, words, ntiles, 1, false);
  switch (sample_format) {
    case PFB_FMT_INT8_IQ:
      rc = extract_raw<PFB_FMT_INT8_IQ>(d_iq, n, inv_scale, fs, fc, sample_start_time, snr_threshold_db, trailing_threshold_db,
                                        out, capacity, count, noise_floor_out, ws, ws2, e, d_hist, d_pair, words, ntiles, tile_words, st);
      break;
    case PFB_FMT_INT16_IQ:
      rc = extract_raw<PFB_FMT_INT16_IQ>(d_iq, n, inv_scale, fs, fc, sample_start_time, snr_threshold_db, trailing_threshold_db,
                                         out, capacity, count, noise_floor_out, ws, ws2, e, d_hist, d_pair, words, ntiles, tile_words, st);
      break;
    default:
      rc = extract_raw<PFB_FMT_CF32>(d_iq, n, inv_scale, fs, fc, sample_start_time, snr_threshold_db, trailing_threshold_db,
                                     out, capacity, count, noise_floor_out, ws, ws2, e, d_hist, d_pair, words, ntiles, tile_words, st);
      break;
  }
done:
  (void)hipStreamSynchronize(st);
  return rc;
}

extern "C" int pfb_pdw_extract_raw(const void* iq, uint64_t num_samples, uint32_t sample_format, uint32_t bit_width,
                                   double fs, double fc, double sample_start_time, double snr_threshold_db,
                                   double trailing_threshold_db, pfb_pdw* out, uint64_t capacity, uint64_t* count,
                                   double* noise_floor_out, uint32_t mem, int32_t device_id, void* hip_stream) {
  return pfb::abi_guard([&] {
    return pdw_extract_raw_impl(iq, num_samples, sample_format, bit_width, fs, fc, sample_start_time, snr_threshold_db,
                                trailing_threshold_db, out, capacity, count, noise_floor_out, mem, device_id, hip_stream);
  });
}
