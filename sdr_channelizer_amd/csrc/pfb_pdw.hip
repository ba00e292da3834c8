// pfb_pdw.hip -- channelized PDW extraction on the GPU (include/pfb_channelizer.h, pfb_pdw_extract).
//
// Restates the second half of /root/reference/matlab/create_pdws_channelized.m (lines 64-143) as
// data-parallel passes over the F x M channelizer output (frame-major complex64, fftshift-ed):
//
//   noise floor  :73    exact per-channel median of |y| by MSB-first radix select on the float64 key:
//                       8-bit digit histograms in LDS (lane = channel, so LDS atomics never collide),
//                       passes until the candidate bucket is small, then an exact finish on the
//                       collected candidates.  No sort of the F values.
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
#include <string>
#include <vector>

#include "pfb_channelizer.h"

namespace {

constexpr int kTile = 512;        // frames per tile of the edge scan
constexpr int kCand = 2048;       // candidate capacity per channel for the exact median finish
constexpr int kPulseCache = 1024; // per-pulse values cached in LDS up to this many
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

// ---------------------------------------------------------------------------------
// noise floor: radix select of rank[col] over the column's magnitudes

// one 8-bit digit histogram pass. grid = (column groups of 64, row blocks); block = 256 (4 waves)
__global__ void __launch_bounds__(256) pdw_hist_kernel(const float2* y, long long F, int M, int pass,
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
      const unsigned long long k = dkey(mag2_of(y[r * M + col]));  // ordered like the magnitude, no sqrt
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

// ---------------------------------------------------------------------------------
// edges

// automaton step of create_pdws_channelized.m:85-135: inactive -> active on mag >= thr (:87),
// active -> inactive on mag <= thr (:94)
__device__ __forceinline__ int step_state(int active, double m, double thr) {
  return active ? (m > thr) : (m >= thr);
}

// tile summaries: fn[tile][col] = f(0) | f(1) << 1.  grid = (column groups, tile groups of 4), one tile per wave
__global__ void __launch_bounds__(256) pdw_tilefn_kernel(const float2* y, long long F, int M, const double* thr,
                                                         unsigned char* fn, long long ntiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  const long long tile = (long long)blockIdx.y * 4 + wave;
  if (col >= M || tile >= ntiles) return;
  const double t = thr[col];
  const long long r0 = tile * kTile, r1 = (r0 + kTile < F) ? r0 + kTile : F;
  int s0 = 0, s1 = 1;
  for (long long r = r0; r < r1; ++r) {
    const double m = mag_of(y[r * M + col]);
    s0 = step_state(s0, m, t);
    s1 = step_state(s1, m, t);
  }
  fn[tile * M + col] = (unsigned char)(s0 | (s1 << 1));
}

// per column: incoming state of every tile.  One wave per column: each lane composes the transition
// functions of a contiguous segment of tiles, the 64 segment functions are scanned across the wave,
// then each lane replays its segment from its incoming state.
__global__ void __launch_bounds__(64) pdw_tilescan_kernel(int M, long long ntiles, const unsigned char* fn,
                                                          unsigned char* state_in) {
  const int col = blockIdx.x, lane = threadIdx.x;
  const long long per = (ntiles + 63) / 64;
  const long long t0 = lane * per, t1 = (t0 + per < ntiles) ? t0 + per : ntiles;
  int f = 0x2;  // identity: f(0)=0, f(1)=1  -> bits (f0 | f1<<1) = 0b10
  for (long long t = t0; t < t1; ++t) {
    const int g = fn[t * M + col];  // apply g after f: h(s) = g(f(s))
    f = ((g >> (f & 1)) & 1) | (((g >> ((f >> 1) & 1)) & 1) << 1);
  }
  // inclusive scan of function composition across lanes (lane order = time order)
  int inc = f;
  for (int d = 1; d < 64; d <<= 1) {
    const int prev = __shfl_up(inc, d);
    if (lane >= d) inc = ((inc >> (prev & 1)) & 1) | (((inc >> ((prev >> 1) & 1)) & 1) << 1);
  }
  int exc = __shfl_up(inc, 1);
  if (lane == 0) exc = 0x2;
  int s = exc & 1;  // state entering my segment when the stream starts inactive: exc(0)
  for (long long t = t0; t < t1; ++t) {
    state_in[t * M + col] = (unsigned char)s;
    s = (fn[t * M + col] >> s) & 1;
  }
}

// replay a tile: count (EMIT=false) or write (EMIT=true) leading / trailing edge frame indices
template <bool EMIT>
__global__ void __launch_bounds__(256) pdw_edges_kernel(const float2* y, long long F, int M, const double* thr,
                                                        const unsigned char* state_in, long long ntiles,
                                                        unsigned* cnt_s, unsigned* cnt_e, const unsigned long long* off_s,
                                                        const unsigned long long* off_e, long long* starts, long long* ends) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  const long long tile = (long long)blockIdx.y * 4 + wave;
  if (col >= M || tile >= ntiles) return;
  const double t = thr[col];
  const long long r0 = tile * kTile, r1 = (r0 + kTile < F) ? r0 + kTile : F;
  int s = state_in[tile * M + col];
  unsigned ns = 0, ne = 0;
  unsigned long long os = 0, oe = 0;
  if (EMIT) { os = off_s[tile * M + col]; oe = off_e[tile * M + col]; }
  for (long long r = r0; r < r1; ++r) {
    const double m = mag_of(y[r * M + col]);
    const int n = step_state(s, m, t);
    if (n != s) {
      if (n) { if (EMIT) starts[os + ns] = r; ++ns; }
      else   { if (EMIT) ends[oe + ne] = r; ++ne; }
    }
    s = n;
  }
  if (!EMIT) { cnt_s[tile * M + col] = ns; cnt_e[tile * M + col] = ne; }
}

// per column exclusive prefix of the tile counts; column totals.  One wave per column, same segment
// decomposition as the state scan.
__global__ void __launch_bounds__(64) pdw_offsets_kernel(int M, long long ntiles, const unsigned* cnt_s,
                                                         const unsigned* cnt_e, unsigned long long* off_s,
                                                         unsigned long long* off_e, unsigned long long* tot_s,
                                                         unsigned long long* tot_e) {
  const int col = blockIdx.x, lane = threadIdx.x;
  const long long per = (ntiles + 63) / 64;
  const long long t0 = lane * per, t1 = (t0 + per < ntiles) ? t0 + per : ntiles;
  unsigned long long a = 0, b = 0;
  for (long long t = t0; t < t1; ++t) { a += cnt_s[t * M + col]; b += cnt_e[t * M + col]; }
  unsigned long long ia = a, ib = b;
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned long long pa = __shfl_up(ia, d), pb = __shfl_up(ib, d);
    if (lane >= d) { ia += pa; ib += pb; }
  }
  unsigned long long ea = ia - a, eb = ib - b;  // exclusive
  for (long long t = t0; t < t1; ++t) {
    off_s[t * M + col] = ea; off_e[t * M + col] = eb;
    ea += cnt_s[t * M + col]; eb += cnt_e[t * M + col];
  }
  if (lane == 63) { tot_s[col] = ia; tot_e[col] = ib; }
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

// k-th smallest of n doubles produced by get(i), by 8 passes of 8-bit digits; whole block cooperates.
template <class Get>
__device__ double block_select(Get get, long long n, long long k, unsigned* hist /* [256] shared */) {
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
    unsigned long long cum = 0;
    int d = 0;
    for (; d < 255; ++d) {  // every thread walks the same 256 counters: uniform result
      if (cum + hist[d] > (unsigned long long)k) break;
      cum += hist[d];
    }
    prefix |= (unsigned long long)d << shift;
    k -= (long long)cum;
    __syncthreads();
  }
  return dkey_inv(prefix);
}

template <class Get>
__device__ double block_median(Get get, long long n, unsigned* hist) {
  const double hi = block_select(get, n, n / 2, hist);
  if (n & 1) return hi;
  return 0.5 * (block_select(get, n, n / 2 - 1, hist) + hi);
}

__global__ void __launch_bounds__(256) pdw_pulse_kernel(const float2* y, int M, const long long* starts,
                                                        const long long* ends, const unsigned long long* base_s,
                                                        const unsigned long long* base_e, const double* nf, const double* bin_freqs, double fs, double fc,
                                                        double t0, unsigned flags, pfb_pdw* out, unsigned long long capacity) {
  __shared__ unsigned hist[256];
  __shared__ double cache[kPulseCache];
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
    amp = block_median([&](long long i) { return cache[i]; }, n, hist);
  } else {
    amp = block_median([&](long long i) { return mag_of(y[(toa + i) * M + b]); }, n, hist);
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
    med = block_median([&](long long i) { return cache[i]; }, n - 1, hist);
  } else {
    med = block_median(dphi, n - 1, hist);
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

}  // namespace

extern "C" const char* pfb_pdw_last_error_detail(void) { return g_pdw_detail.c_str(); }

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
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  const long long F = (long long)frames;
  const int Mi = (int)M;
  const long long ntiles = (F + kTile - 1) / kTile;
  const int cgroups = (Mi + 63) / 64;
  const double fs = fs_in / (double)decimation;  // :62
  int rc = PFB_OK;

  // device scratch
  float2* d_y_own = nullptr;
  const float2* d_y = nullptr;
  unsigned *d_hist = nullptr, *d_bucket = nullptr, *d_cand_n = nullptr, *d_cnt_s = nullptr, *d_cnt_e = nullptr;
  unsigned long long *d_prefix = nullptr, *d_rank = nullptr, *d_below = nullptr, *d_maxbelow = nullptr;
  unsigned long long *d_off_s = nullptr, *d_off_e = nullptr, *d_tot = nullptr, *d_base = nullptr;
  double *d_cand = nullptr, *d_nf = nullptr, *d_thr = nullptr, *d_binf = nullptr;
  unsigned char *d_fn = nullptr, *d_state = nullptr;
  long long *d_starts = nullptr, *d_ends = nullptr;
  pfb_pdw* d_out = nullptr;
  std::vector<unsigned> h_bucket(M);
  std::vector<unsigned long long> h_rank(M, (unsigned long long)(F / 2)), h_tot(2 * (size_t)M), h_base(2 * (size_t)M);
  std::vector<double> h_nf(M), h_thr(M), h_binf(M);
  unsigned long long total_s = 0, total_e = 0;
  int passes = 0;
  const int row_blocks = (int)std::min<long long>(1024, std::max<long long>(1, F / 256));

  if (mem == PFB_MEM_HOST) {
    PDW_TRY(hipMalloc((void**)&d_y_own, (size_t)F * M * sizeof(float2)));
    PDW_TRY(hipMemcpyAsync(d_y_own, y_in, (size_t)F * M * sizeof(float2), hipMemcpyHostToDevice, st));
    d_y = d_y_own;
  } else {
    d_y = static_cast<const float2*>(y_in);
  }
  PDW_TRY(hipMalloc((void**)&d_hist, (size_t)M * 256 * sizeof(unsigned)));
  PDW_TRY(hipMalloc((void**)&d_bucket, M * sizeof(unsigned)));
  PDW_TRY(hipMalloc((void**)&d_cand_n, M * sizeof(unsigned)));
  PDW_TRY(hipMalloc((void**)&d_prefix, M * sizeof(unsigned long long)));
  PDW_TRY(hipMalloc((void**)&d_rank, M * sizeof(unsigned long long)));
  PDW_TRY(hipMalloc((void**)&d_below, M * sizeof(unsigned long long)));
  PDW_TRY(hipMalloc((void**)&d_maxbelow, M * sizeof(unsigned long long)));
  PDW_TRY(hipMalloc((void**)&d_cand, (size_t)M * kCand * sizeof(double)));
  PDW_TRY(hipMalloc((void**)&d_nf, M * sizeof(double)));
  PDW_TRY(hipMalloc((void**)&d_thr, M * sizeof(double)));
  PDW_TRY(hipMalloc((void**)&d_binf, M * sizeof(double)));
  PDW_TRY(hipMemsetAsync(d_hist, 0, (size_t)M * 256 * sizeof(unsigned), st));
  PDW_TRY(hipMemsetAsync(d_prefix, 0, M * sizeof(unsigned long long), st));
  PDW_TRY(hipMemsetAsync(d_below, 0, M * sizeof(unsigned long long), st));
  PDW_TRY(hipMemsetAsync(d_maxbelow, 0, M * sizeof(unsigned long long), st));
  PDW_TRY(hipMemsetAsync(d_cand_n, 0, M * sizeof(unsigned), st));
  PDW_TRY(hipMemcpyAsync(d_rank, h_rank.data(), M * sizeof(unsigned long long), hipMemcpyHostToDevice, st));

  // ---- noise floor (:73): radix select of rank F/2, then the exact finish
  for (passes = 0; passes < 8;) {
    hipLaunchKernelGGL(pdw_hist_kernel, dim3(cgroups, row_blocks), dim3(256), 0, st, d_y, F, Mi, passes, d_prefix, d_hist);
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
  {
    const double gain = std::pow(10.0, snr_threshold_db / 10.0);  // :74-75 (dB applied to magnitude with /10)
    for (uint32_t b = 0; b < M; ++b) h_thr[b] = h_nf[b] * gain;
    pfb_center_frequencies(M, fs_in, h_binf.data());              // :42, before fs is decimated
    if (noise_floor_out) std::memcpy(noise_floor_out, h_nf.data(), M * sizeof(double));
  }
  PDW_TRY(hipMemcpyAsync(d_thr, h_thr.data(), M * sizeof(double), hipMemcpyHostToDevice, st));
  PDW_TRY(hipMemcpyAsync(d_binf, h_binf.data(), M * sizeof(double), hipMemcpyHostToDevice, st));

  // ---- edges (:85-135)
  PDW_TRY(hipMalloc((void**)&d_fn, (size_t)ntiles * M));
  PDW_TRY(hipMalloc((void**)&d_state, (size_t)ntiles * M));
  PDW_TRY(hipMalloc((void**)&d_cnt_s, (size_t)ntiles * M * sizeof(unsigned)));
  PDW_TRY(hipMalloc((void**)&d_cnt_e, (size_t)ntiles * M * sizeof(unsigned)));
  PDW_TRY(hipMalloc((void**)&d_off_s, (size_t)ntiles * M * sizeof(unsigned long long)));
  PDW_TRY(hipMalloc((void**)&d_off_e, (size_t)ntiles * M * sizeof(unsigned long long)));
  PDW_TRY(hipMalloc((void**)&d_tot, 2 * (size_t)M * sizeof(unsigned long long)));
  PDW_TRY(hipMalloc((void**)&d_base, 2 * (size_t)M * sizeof(unsigned long long)));
  {
    const dim3 tgrid(cgroups, (unsigned)((ntiles + 3) / 4));
    hipLaunchKernelGGL(pdw_tilefn_kernel, tgrid, dim3(256), 0, st, d_y, F, Mi, d_thr, d_fn, ntiles);
    hipLaunchKernelGGL(pdw_tilescan_kernel, dim3(Mi), dim3(64), 0, st, Mi, ntiles, d_fn, d_state);
    hipLaunchKernelGGL(pdw_edges_kernel<false>, tgrid, dim3(256), 0, st, d_y, F, Mi, d_thr, d_state, ntiles, d_cnt_s, d_cnt_e,
                       (const unsigned long long*)nullptr, (const unsigned long long*)nullptr, (long long*)nullptr,
                       (long long*)nullptr);
    hipLaunchKernelGGL(pdw_offsets_kernel, dim3(Mi), dim3(64), 0, st, Mi, ntiles, d_cnt_s, d_cnt_e, d_off_s, d_off_e,
                       d_tot, d_tot + M);
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
      PDW_TRY(hipMemcpyAsync(d_base, h_base.data(), 2 * (size_t)M * sizeof(unsigned long long), hipMemcpyHostToDevice, st));
      PDW_TRY(hipMalloc((void**)&d_starts, (size_t)std::max<unsigned long long>(total_s, 1) * sizeof(long long)));
      PDW_TRY(hipMalloc((void**)&d_ends, (size_t)total_e * sizeof(long long)));
      hipLaunchKernelGGL(pdw_rebase_kernel, dim3((unsigned)((ntiles * M + 255) / 256)), dim3(256), 0, st, Mi, ntiles, d_off_s,
                         d_off_e, d_base, d_base + M);
      hipLaunchKernelGGL(pdw_edges_kernel<true>, tgrid, dim3(256), 0, st, d_y, F, Mi, d_thr, d_state, ntiles, d_cnt_s, d_cnt_e,
                         (const unsigned long long*)d_off_s, (const unsigned long long*)d_off_e, d_starts, d_ends);
      if (n_out > 0) {
        PDW_TRY(hipMalloc((void**)&d_out, (size_t)n_out * sizeof(pfb_pdw)));
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
  (void)hipFree(d_y_own); (void)hipFree(d_hist); (void)hipFree(d_bucket); (void)hipFree(d_cand_n); (void)hipFree(d_prefix);
  (void)hipFree(d_rank); (void)hipFree(d_below); (void)hipFree(d_maxbelow); (void)hipFree(d_cand); (void)hipFree(d_nf);
  (void)hipFree(d_thr); (void)hipFree(d_binf); (void)hipFree(d_fn); (void)hipFree(d_state); (void)hipFree(d_cnt_s);
  (void)hipFree(d_cnt_e); (void)hipFree(d_off_s); (void)hipFree(d_off_e); (void)hipFree(d_tot); (void)hipFree(d_base);
  (void)hipFree(d_starts); (void)hipFree(d_ends); (void)hipFree(d_out);
  if (device_id >= 0 && device_id != prev_dev && prev_dev >= 0) (void)hipSetDevice(prev_dev);
  return rc;
}
