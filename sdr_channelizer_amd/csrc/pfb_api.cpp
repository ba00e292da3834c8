// pfb_api.cpp -- host side of the C ABI in include/pfb_channelizer.h.
//
// Owns: the handle (taps, twiddles, history, counters), kernel selection, the
// host-pointer staging path and the state blob.  All arithmetic is in
// pfb_kernels.hip; there is deliberately no CPU implementation here.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <utility>
#include <thread>
#include <vector>

#include "pfb_common.h"
#include "pfb_channelizer_dev.h"

namespace {

thread_local std::string g_detail;

int hip_fail(hipError_t e, const char* what) {
  char buf[256];
  std::snprintf(buf, sizeof(buf), "%s: %s (%d)", what, hipGetErrorString(e), (int)e);
  g_detail = buf;
  (void)hipGetLastError();  // clear the sticky error
  return (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorInsufficientDriver)
             ? PFB_ERR_NO_DEVICE
             : (e == hipErrorOutOfMemory ? PFB_ERR_NO_MEMORY : PFB_ERR_HIP);
}

#define HIP_TRY(expr)                                  \
  do {                                                 \
    const hipError_t e__ = (expr);                     \
    if (e__ != hipSuccess) return hip_fail(e__, #expr); \
  } while (0)

struct DeviceGuard {  // run on the handle's device, restore the caller's afterwards
  int prev = -1;
  bool switched = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = (hipSetDevice(dev) == hipSuccess);
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
};

constexpr uint32_t kStateMagic = 0x50464231u;  // "PFB1"

struct StateHeader {
  uint32_t magic, M, P, D, fmt, hist_samples, phase, reserved;
  uint64_t frame_index;
};

}  // namespace

struct pfb_handle {
  int M = 0, P = 0, D = 0, off = 0;
  int fmt = 0, bit_width = 0, layout = 0;
  unsigned flags = 0;
  int device = 0;
  int num_cus = 256;       // compute units of the device (slab sizing)
  int bps = 0;             // bytes per input sample
  int out_elem = 8;        // bytes per output element: complex64, or float32 with PFB_FLAG_MAGNITUDE
  int hist_samples = 0;    // M*P + D
  float* d_taps = nullptr;   // M*P, scaled by 2^-(bit_width-1)
  float2* d_tw = nullptr;    // M
  float* d_taps_lane = nullptr;   // fast kernels: per-column tap table
  float2* d_tw_lane = nullptr;    // fast kernels: inter-pass twiddle rows
  void* d_hist[2] = {nullptr, nullptr};
  int cur = 0;               // which history buffer is current
  uint32_t phase = 0;        // samples carried since the last frame boundary (0..D-1)
  uint64_t frame_index = 0;  // global index of the next frame
  hipStream_t stream = nullptr;
  const pfb::FastKernelInfo* fast = nullptr;
  // options
  int opt_kernel = 0;
  int opt_frames_per_block = 0;
  int64_t opt_host_chunk = 0;
  int opt_nontemporal = 0;
  int opt_xcd_remap = -1;  // -1: per schedule (on for 0..3, off for the wave-pair schedule)
  int opt_experiment = 0;
  int opt_variant = 0;
  int opt_schedule = -1;  // -1: the instantiation's measured default
  int opt_grid = 0;
  int opt_tile_waves = 8;
  int64_t opt_slab_frames = 0;  // channel-major by slabs: frames per slab (0 = ~32 MiB of output)
  void* d_slab = nullptr;       // frame-major scratch of the slab path
  size_t slab_bytes = 0;
  void* d_matrix = nullptr;     // pfb_pdw_from_iq_file: the record's channel matrix (grow-only)
  size_t matrix_bytes = 0;
  const char* last_kernel = "";
  // host staging: two sets, so chunk i+1 crosses PCIe inbound while chunk i is transformed and chunk i-1 goes out
  void* d_stage_in = nullptr;   // set 0 (also pfb_prime's scratch)
  void* d_stage_out = nullptr;
  void* d_stage_in2 = nullptr;  // set 1
  void* d_stage_out2 = nullptr;
  size_t stage_in_bytes = 0, stage_out_bytes = 0;
  hipStream_t s_in = nullptr, s_out = nullptr;  // copy streams of the host path
  hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_k[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
  // PFB_OPT_PROFILE: event pairs around each channelizer kernel launch
  int opt_profile = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;  // reusable pairs
  size_t ev_used = 0;
  // time sharding (pfb_shard_attach): this handle owns segment shard_rank of shard_world
  int shard_rank = 0, shard_world = 1;
  bool shard_ring = false;
  pfb_halo_exchange_fn shard_exchange = nullptr;
  void* shard_user = nullptr;
  void* d_halo = nullptr;            // hist_samples raw samples; the predecessor's tail lands in its last halo_samples
  hipStream_t s_halo = nullptr;      // side stream of the exchange
  hipEvent_t ev_seg = nullptr, ev_halo = nullptr;
  hipEvent_t ev_switch = nullptr;    // pfb_set_stream: orders the old stream's work in front of the new stream's
};

namespace {

void free_handle(pfb_handle* h) {
  if (!h) return;
  DeviceGuard g(h->device);
  (void)hipFree(h->d_taps);
  (void)hipFree(h->d_tw);
  (void)hipFree(h->d_taps_lane);
  (void)hipFree(h->d_tw_lane);
  (void)hipFree(h->d_hist[0]);
  (void)hipFree(h->d_hist[1]);
  (void)hipFree(h->d_stage_in);
  (void)hipFree(h->d_stage_out);
  (void)hipFree(h->d_stage_in2);
  (void)hipFree(h->d_stage_out2);
  (void)hipFree(h->d_slab);
  (void)hipFree(h->d_matrix);
  (void)hipFree(h->d_halo);
  if (h->s_in) (void)hipStreamDestroy(h->s_in);
  if (h->s_out) (void)hipStreamDestroy(h->s_out);
  if (h->s_halo) (void)hipStreamDestroy(h->s_halo);
  if (h->ev_seg) (void)hipEventDestroy(h->ev_seg);
  if (h->ev_halo) (void)hipEventDestroy(h->ev_halo);
  if (h->ev_switch) (void)hipEventDestroy(h->ev_switch);
  for (int i = 0; i < 2; ++i) {
    if (h->ev_in[i]) (void)hipEventDestroy(h->ev_in[i]);
    if (h->ev_k[i]) (void)hipEventDestroy(h->ev_k[i]);
    if (h->ev_out[i]) (void)hipEventDestroy(h->ev_out[i]);
  }
  for (auto& pr : h->ev_pool) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  delete h;
}

uint64_t frames_for(const pfb_handle* h, uint64_t n) { return (h->phase + n) / (uint64_t)h->D; }

// Launch the channelizer kernel(s) for local frames [f_begin, f_end) of a device buffer of n samples.  `hist` holds
// the hist_samples raw samples in front of d_iq[0]; it is only read when f_begin == 0 (a later range starts at
// least hist_samples into the buffer, so its "history" is the buffer itself).  Output row f lands where a call
// over the whole buffer would put it.  No state change, no host sync.
int launch_frames(pfb_handle* h, const void* d_iq, uint64_t n, const void* hist, void* d_out, uint64_t f_begin,
                  uint64_t f_end, int64_t out_ld, int64_t out_frame0) {
  if (f_end <= f_begin) return PFB_OK;
  if (f_begin > 0) {
    if (f_begin * (uint64_t)h->D < (uint64_t)h->hist_samples) return PFB_ERR_BAD_ARG;
    const size_t skip = (size_t)f_begin * h->D * h->bps;
    hist = static_cast<const char*>(d_iq) + skip - (size_t)h->hist_samples * h->bps;
    d_iq = static_cast<const char*>(d_iq) + skip;
    n -= f_begin * (uint64_t)h->D;
    if (h->layout == PFB_LAYOUT_FRAME_MAJOR) d_out = static_cast<char*>(d_out) + (size_t)f_begin * h->M * h->out_elem;
    else out_frame0 += (int64_t)f_begin;
  }
  const uint64_t frames = f_end - f_begin;
  pfb::KernelParams p{};
  p.in = d_iq;
  p.hist = hist;
  p.out = static_cast<float2*>(d_out);
  p.taps = h->d_taps;
  p.tw = h->d_tw;
  p.taps_lane = h->d_taps_lane;
  p.tw_lane = h->d_tw_lane;
  p.n_in = (long long)n;
  p.frames = (long long)frames;
  p.frame0 = (long long)(h->frame_index + f_begin);
  p.out_ld = out_ld;
  p.out_frame0 = out_frame0;
  p.base = (h->off - (int)h->phase) - (h->D - 1);
  p.hist_samples = h->hist_samples;
  p.M = h->M; p.P = h->P; p.D = h->D;
  p.fmt = h->fmt;
  p.layout = h->layout;
  p.flags = h->flags;
  p.nontemporal = h->opt_nontemporal;
  p.xcd_remap = h->opt_xcd_remap < 0 ? 1 : h->opt_xcd_remap;
  p.experiment = h->opt_experiment;
  p.grid_override = h->opt_grid;
  p.tile_waves = h->opt_tile_waves;
  const bool want_fast = h->fast && h->opt_kernel != 1;
  if (h->opt_kernel == 2 && !want_fast) return PFB_ERR_UNSUPPORTED;
  hipEvent_t ev_first = nullptr, ev_second = nullptr;
  if (h->opt_profile && h->ev_used < 4096) {
    if (h->ev_used == h->ev_pool.size()) {
      hipEvent_t a = nullptr, b = nullptr;
      HIP_TRY(hipEventCreate(&a));
      if (hipEventCreate(&b) != hipSuccess) { (void)hipEventDestroy(a); return hip_fail(hipGetLastError(), "hipEventCreate"); }
      h->ev_pool.emplace_back(a, b);
    }
    ev_first = h->ev_pool[h->ev_used].first;
    ev_second = h->ev_pool[h->ev_used].second;
    HIP_TRY(hipEventRecord(ev_first, h->stream));
    ++h->ev_used;  // only a pair whose first event was recorded counts as used
  }
  // Channel-major output of a fused shape is written by the kernel itself (its transposed-tile or plain
  // channel-major instantiation).  The team plans (M = 1024, 560) have none -- their chunks of 4 frames would be
  // 32-byte runs -- and go by slabs instead: the frame-major kernel fills a scratch slab, a transpose kernel
  // moves it into place (M = 1024: 2x the 8-wave plan's fused channel-major stores).  A slab must be long enough
  // to fill the chip with runs, so it does not fit the memory-side cache; PFB_OPT_SCHEDULE 9 forces the slabs on
  // any shape, PFB_OPT_SLAB_FRAMES sets their length.
  const bool cm = h->layout == PFB_LAYOUT_CHANNEL_MAJOR;
  const bool forced_fused = h->opt_schedule == 0 || h->opt_schedule == 2 || h->opt_schedule == 8;
  // (A fused route for the team plans -- the team kernel transposing its own tiles through an L2-resident scratch -- was
  // bit-identical but slower than the slabs, 7.4 against 6.4 ms per 2^30 samples at M = 1024, and was removed.)
  const bool by_slabs = cm && want_fast && (!h->fast->channel_major_ok || h->opt_schedule == 9);
  if (want_fast) {
    const int c = h->fast->chunk_frames;
    int fpb = h->opt_frames_per_block > 0 ? h->opt_frames_per_block : h->fast->default_frames_per_block;
    fpb = ((fpb + c - 1) / c) * c;
    p.schedule = (h->opt_schedule >= 0 && h->opt_schedule != 9) ? h->opt_schedule : h->fast->default_schedule;
    if (h->opt_schedule < 0 && (h->flags & PFB_FLAG_MAGNITUDE) && h->fast->magnitude_schedule >= 0 && !cm) {
      p.schedule = h->fast->magnitude_schedule;  // fused abs(): magnitudes staged in LDS, sliding runs
    }
    if (cm && !by_slabs)  // fused channel-major: 0 = sliding runs, 2 = tiles, 8 = short runs transposed in LDS, else the kernel's pick
      p.schedule = forced_fused ? h->opt_schedule : -1;
    if (p.schedule == 3 && h->opt_frames_per_block <= 0) fpb = 24;
    // Team kernels run one workgroup per CU, so their runs are dealt in rounds of num_cus, and a last round that is not
    // full costs a whole round: 683 593 frames of M = 1024 in the tuned 512-frame runs are 5.2 rounds = 6 (0.549 of the
    // roofline), in 672-frame runs 3.97 rounds (0.600).  Unless the caller fixed it, the run length is the call's frames
    // split evenly over k full rounds, k chosen for runs near twice the tuned length (full rounds of 1024-frame runs
    // measured +1.3 % over 512: half the pipeline fills and drains) -- short calls thereby spread over every CU instead
    // of filling a few.  (The slab route sizes its slabs as one 512-frame run per CU: already whole rounds.)
    if (p.schedule == 6 && h->opt_frames_per_block <= 0 && !by_slabs && frames > 0) {
      // (plans of <= 8 waves are built for several workgroups per CU, 16 waves in all: their rounds are that much wider)
      const long long slots = (long long)h->num_cus * std::max(1, 16 / ((h->fast->threads + 64 * c) / 64)), target = 2ll * fpb;
      const long long k = std::max<long long>(1, ((long long)frames + slots * target / 2) / (slots * target));
      const long long even = ((long long)frames + k * slots - 1) / (k * slots);
      fpb = (int)std::min<long long>(std::max<long long>(even, 2 * c), 4 * target);
    }
    // The other kernels with long runs (a wave pair, a lockstep workgroup or a single wave per run of 128-512 frames): a
    // short call must not leave most of the chip idle -- 2 * 10^7 samples of M = 1024 in 512-frame runs kept 39 of 256 CUs
    // busy (0.105 of the roofline; 0.456 spread over all of them).  When the tuned run length gives fewer runs than the
    // chip holds at once, the runs shrink until it is full (at least one chunk pair each).
    if ((p.schedule == 0 || p.schedule == 7 || p.schedule == 11 || p.schedule == 13) && h->opt_frames_per_block <= 0 && !by_slabs &&
        !cm && frames > 0) {
      const int waves = std::max(1, h->fast->threads / 64);
      const long long per_cu = p.schedule == 7 ? 6 : p.schedule == 13 ? 2 : std::max(1, 8 / waves);  // runs resident per CU
      const long long slots = h->num_cus * per_cu;
      if (((long long)frames + fpb - 1) / fpb < slots) {
        const long long even = ((long long)frames + slots - 1) / slots;
        fpb = (int)std::min<long long>(fpb, std::max<long long>(2 * c, (even + c - 1) / c * c));
      }
    }
    if (p.schedule == 6 || p.schedule == 7) fpb = ((fpb + 2 * c - 1) / (2 * c)) * (2 * c);  // these kernels walk chunks in pairs
    if (p.schedule == 4) {
      if (h->opt_frames_per_block <= 0) fpb = 64;
      if (h->opt_xcd_remap < 0) p.xcd_remap = 0;  // 512-frame workgroups: one dense sweep beats L2 halo hits
    }
    // short sliding runs in dispatch order already sweep the stream as one window: leave them round-robin over the XCDs
    if ((p.schedule == 0 || p.schedule == 11) && fpb <= 64 && h->opt_xcd_remap < 0) p.xcd_remap = 0;
    p.frames_per_block = fpb;
    const int cpt = h->fast->cols_per_thread;
    const int bmod = ((p.base % cpt) + cpt) % cpt;
    p.vec_ok = (bmod == 0) && (reinterpret_cast<uintptr_t>(d_iq) % (uintptr_t)(h->bps * cpt) == 0);
    if (by_slabs) {
      long long sf = h->opt_slab_frames > 0 ? h->opt_slab_frames : (long long)h->num_cus * fpb;  // one run per CU
      sf = std::max<long long>(64, (sf + 63) / 64 * 64);
      sf = std::max<long long>(sf, (h->hist_samples + h->D - 1) / h->D + 1);  // a later slab's window reaches back into the input, never into the history
      sf = std::min<long long>(sf, 65535ll * 64);  // the transpose kernel's grid: one row of 64 x 64 tiles per 64 frames
      sf = std::min<long long>(sf, ((long long)frames + 63) / 64 * 64);
      // (tried: two slabs and a side stream, slab k transposed while slab k + 1 is filled -- 6.9 ms instead of 6.4 per 2^30
      // samples at M = 1024: the two kernels slow each other down by more than the overlap buys.  One slab, one stream.)
      const size_t need = (size_t)sf * h->M * h->out_elem;
      if (need > h->slab_bytes) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        (void)hipFree(h->d_slab);
        h->d_slab = nullptr; h->slab_bytes = 0;
        HIP_TRY(hipMalloc(&h->d_slab, need));
        h->slab_bytes = need;
      }
      for (long long f0 = 0; f0 < (long long)frames; f0 += sf) {
        pfb::KernelParams q = p;
        q.layout = PFB_LAYOUT_FRAME_MAJOR;
        q.out = static_cast<float2*>(h->d_slab);
        q.frames = std::min<long long>(sf, (long long)frames - f0);
        q.frame0 = p.frame0 + f0;
        q.in = static_cast<const char*>(d_iq) + (size_t)f0 * h->D * h->bps;
        q.n_in = (long long)n - f0 * h->D;
        if (f0 > 0)  // "history" of a later slab = the input samples in front of it
          q.hist = static_cast<const char*>(q.in) - (size_t)h->hist_samples * h->bps;
        HIP_TRY(h->fast->launch(q, h->stream));
        HIP_TRY(pfb::launch_transpose_slab(h->d_slab, q.frames, h->M, d_out, out_ld, out_frame0 + f0, h->out_elem, h->stream));
      }
    } else {
      HIP_TRY(h->fast->launch(p, h->stream));
    }
    h->last_kernel = h->fast->name;
  } else {
    HIP_TRY(pfb::launch_generic(p, h->stream));
    h->last_kernel = "pfb_generic";
  }
  if (ev_second) HIP_TRY(hipEventRecord(ev_second, h->stream));
  return PFB_OK;
}

// carry the last hist_samples raw samples of [history | d_iq] and advance the counters
int advance_state(pfb_handle* h, const void* d_iq, uint64_t n, uint64_t frames) {
  if (n > 0) {
    HIP_TRY(pfb::launch_update_history(h->d_hist[h->cur], d_iq, (long long)n, h->d_hist[h->cur ^ 1],
                                       h->hist_samples, h->bps, h->stream));
    h->cur ^= 1;
  }
  h->phase = (uint32_t)((h->phase + n) % (uint64_t)h->D);
  h->frame_index += frames;
  return PFB_OK;
}

// enqueue kernel + history update for device-resident buffers; no host sync
int enqueue(pfb_handle* h, const void* d_iq, uint64_t n, void* d_out, uint64_t frames, int64_t out_ld,
            int64_t out_frame0) {
  const int rc = launch_frames(h, d_iq, n, h->d_hist[h->cur], d_out, 0, frames, out_ld, out_frame0);
  if (rc != PFB_OK) return rc;
  return advance_state(h, d_iq, n, frames);
}

int ensure_stage(pfb_handle* h, size_t in_bytes, size_t out_bytes, bool both_sets = false) {
  if (in_bytes > h->stage_in_bytes || (both_sets && in_bytes > 0 && !h->d_stage_in2)) {
    const size_t nb = std::max(in_bytes, h->stage_in_bytes);
    (void)hipFree(h->d_stage_in);
    (void)hipFree(h->d_stage_in2);
    h->d_stage_in = h->d_stage_in2 = nullptr; h->stage_in_bytes = 0;
    HIP_TRY(hipMalloc(&h->d_stage_in, nb));
    if (both_sets) HIP_TRY(hipMalloc(&h->d_stage_in2, nb));
    h->stage_in_bytes = nb;
  }
  if (out_bytes > h->stage_out_bytes || (both_sets && out_bytes > 0 && !h->d_stage_out2)) {
    const size_t nb = std::max(out_bytes, h->stage_out_bytes);
    (void)hipFree(h->d_stage_out);
    (void)hipFree(h->d_stage_out2);
    h->d_stage_out = h->d_stage_out2 = nullptr; h->stage_out_bytes = 0;
    HIP_TRY(hipMalloc(&h->d_stage_out, nb));
    if (both_sets) HIP_TRY(hipMalloc(&h->d_stage_out2, nb));
    h->stage_out_bytes = nb;
  }
  return PFB_OK;
}

// channel-major: `out` is the whole M x out_ld matrix and this call fills rows [out_row0, out_row0 + frames_total) of
// every column (pfb_process: out_ld = frames_total, out_row0 = 0; the .iq front end walks out_row0 through a record)
// device_out: `out` is device memory (frame-major rows follow each other; channel-major as above): nothing is
// copied back and the call returns once the input has left the host buffer, the kernels still queued on the
// handle's stream.
int process_host_pipeline(pfb_handle* h, const void* iq, uint64_t n, void* out, uint64_t frames_total, uint64_t out_ld,
                          uint64_t out_row0, bool device_out);

int process_host(pfb_handle* h, const void* iq, uint64_t n, void* out, uint64_t frames_total, uint64_t out_ld,
                 uint64_t out_row0, bool device_out = false) {
  const int rc = process_host_pipeline(h, iq, n, out, frames_total, out_ld, out_row0, device_out);
  if (rc != PFB_OK) {
    // a failed step returned from the middle of the pipeline: copies to and from the CALLER's buffers may still be
    // in flight on the three streams -- drain them before the caller is told it may free or reuse those buffers
    if (h->s_in) (void)hipStreamSynchronize(h->s_in);
    (void)hipStreamSynchronize(h->stream);
    if (h->s_out) (void)hipStreamSynchronize(h->s_out);
    (void)hipGetLastError();
  }
  return rc;
}

int process_host_pipeline(pfb_handle* h, const void* iq, uint64_t n, void* out, uint64_t frames_total, uint64_t out_ld,
                          uint64_t out_row0, bool device_out) {
  // Stage through device buffers in chunks (multiples of D so chunks never change the carried phase
  // pattern mid-call beyond what the stream semantics already define).  Three streams and two buffer
  // sets: chunk i+1 is copied in while chunk i is transformed and chunk i-1 is copied out, so a caller
  // whose buffers are page-locked (pfb_host_alloc) sees both PCIe directions busy at once; with pageable
  // memory the runtime's own staging serialises the copies and this degrades to the plain sequence.
  uint64_t chunk = h->opt_host_chunk > 0 ? (uint64_t)h->opt_host_chunk : (uint64_t)1 << 24;
  chunk = ((chunk + h->D - 1) / h->D) * h->D;
  const uint64_t max_frames = chunk / h->D + 1;
  const int rc = ensure_stage(h, (size_t)std::min<uint64_t>(chunk, n ? n : 1) * h->bps,
                              device_out ? 0
                                         : (size_t)std::min<uint64_t>(max_frames, frames_total ? frames_total : 1) * h->M *
                                               sizeof(float2),
                              true);
  if (rc != PFB_OK) return rc;
  if (!h->s_in) {
    HIP_TRY(hipStreamCreateWithFlags(&h->s_in, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&h->s_out, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
      HIP_TRY(hipEventCreateWithFlags(&h->ev_in[i], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&h->ev_k[i], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&h->ev_out[i], hipEventDisableTiming));
    }
  }
  void* const st_in[2] = {h->d_stage_in, h->d_stage_in2};
  void* const st_out[2] = {h->d_stage_out, h->d_stage_out2};
  const char* src = static_cast<const char*>(iq);
  char* dst = static_cast<char*>(out);
  uint64_t done = 0, frames_done = 0;
  // whatever the caller queued on the handle's stream comes first
  HIP_TRY(hipEventRecord(h->ev_k[1], h->stream));
  HIP_TRY(hipStreamWaitEvent(h->s_in, h->ev_k[1], 0));
  int rc2 = PFB_OK;
  for (uint64_t i = 0; done < n; ++i) {
    const int b = (int)(i & 1);
    const uint64_t m = std::min<uint64_t>(chunk, n - done);
    const uint64_t f = frames_for(h, m);
    if (i >= 2) HIP_TRY(hipStreamWaitEvent(h->s_in, h->ev_k[b], 0));  // the kernel of chunk i-2 has read this input buffer
    HIP_TRY(hipMemcpyAsync(st_in[b], src + done * h->bps, (size_t)m * h->bps, hipMemcpyHostToDevice, h->s_in));
    HIP_TRY(hipEventRecord(h->ev_in[b], h->s_in));
    HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_in[b], 0));
    if (device_out) {
      if (h->layout == PFB_LAYOUT_FRAME_MAJOR)
        rc2 = enqueue(h, st_in[b], m, dst + frames_done * h->M * h->out_elem, f, (int64_t)f, 0);
      else
        rc2 = enqueue(h, st_in[b], m, dst, f, (int64_t)out_ld, (int64_t)(out_row0 + frames_done));
      if (rc2 != PFB_OK) break;
      HIP_TRY(hipEventRecord(h->ev_k[b], h->stream));
      done += m;
      frames_done += f;
      continue;
    }
    if (i >= 2) HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_out[b], 0));  // chunk i-2 has left this output buffer
    rc2 = enqueue(h, st_in[b], m, st_out[b], f, (int64_t)f, 0);
    if (rc2 != PFB_OK) break;
    HIP_TRY(hipEventRecord(h->ev_k[b], h->stream));
    HIP_TRY(hipStreamWaitEvent(h->s_out, h->ev_k[b], 0));
    if (f > 0) {
      if (h->layout == PFB_LAYOUT_FRAME_MAJOR) {
        HIP_TRY(hipMemcpyAsync(dst + frames_done * h->M * h->out_elem, st_out[b], (size_t)f * h->M * h->out_elem,
                               hipMemcpyDeviceToHost, h->s_out));
      } else {  // column k of this chunk -> rows [frames_done, frames_done+f) of column k of the call
        HIP_TRY(hipMemcpy2DAsync(dst + (out_row0 + frames_done) * h->out_elem, (size_t)out_ld * h->out_elem, st_out[b],
                                 (size_t)f * h->out_elem, (size_t)f * h->out_elem, (size_t)h->M, hipMemcpyDeviceToHost,
                                 h->s_out));
      }
    }
    HIP_TRY(hipEventRecord(h->ev_out[b], h->s_out));
    done += m;
    frames_done += f;
  }
  HIP_TRY(hipStreamSynchronize(h->s_in));
  if (device_out) return rc2;
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipStreamSynchronize(h->s_out));
  return rc2;
}

}  // namespace

extern "C" {

int pfb_abi_version(void) { return PFB_ABI_VERSION; }

const char* pfb_last_error_detail(void) { return g_detail.c_str(); }

const char* pfb_strerror(int status) {
  switch (status) {
    case PFB_OK: return "ok";
    case PFB_ERR_BAD_ARG: return "bad argument";
    case PFB_ERR_BAD_FORMAT: return "bad sample or record format";
    case PFB_ERR_UNSUPPORTED: return "unsupported configuration";
    case PFB_ERR_NO_DEVICE: return "no HIP device";
    case PFB_ERR_HIP: return "HIP runtime error";
    case PFB_ERR_NO_MEMORY: return "out of memory";
    case PFB_ERR_CAPACITY: return "output buffer too small";
    case PFB_ERR_INTERNAL: return "internal error (C++ exception stopped at the ABI)";
    case PFB_ERR_COMM: return "halo exchange callback failed";
    default: return "unknown status";
  }
}

int pfb_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

int pfb_center_frequencies(uint32_t M, double fs, double* out) {
  if (!out || M == 0) return PFB_ERR_BAD_ARG;
  for (uint32_t k = 0; k < M; ++k) {
    const int64_t kk = (k < (M + 1) / 2) ? (int64_t)k : (int64_t)k - (int64_t)M;
    out[k] = (double)kk * fs / (double)M;
  }
  return PFB_OK;
}

int pfb_center_frequencies_ordered(uint32_t M, double fs, uint32_t order, double* out) {
  if (!out || M == 0 || order > PFB_FREQ_ORDER_CENTERED) return PFB_ERR_BAD_ARG;
  if (order == PFB_FREQ_ORDER_FFT) return pfb_center_frequencies(M, fs, out);
  for (uint32_t c = 0; c < M; ++c)  // column c of fftshift(out,2) is FFT column (c + ceil(M/2)) mod M
    out[c] = ((double)c - (double)(M / 2)) * fs / (double)M;
  return PFB_OK;
}

int pfb_selftest_exception_guard(int kind) {
  return pfb::abi_guard([&]() -> int {
    if (kind == 0) throw std::bad_alloc();
    if (kind == 1) throw std::runtime_error("pfb_selftest_exception_guard");
    if (kind == 2) throw 42;
    return PFB_OK;
  });
}

static double bessel_i0(double x) {
  double sum = 1.0, term = 1.0;
  const double q = x * x / 4.0;
  for (int k = 1; k < 200; ++k) {
    term *= q / ((double)k * (double)k);
    sum += term;
    if (term < 1e-17 * sum) break;
  }
  return sum;
}

int pfb_design_prototype(uint32_t M, uint32_t P, double atten_db, float* taps) {
  if (!taps || M == 0 || P == 0) return PFB_ERR_BAD_ARG;
  const double pi = 3.14159265358979323846;
  const int L = (int)(M * P);
  double beta = 0.0;
  if (atten_db > 50.0) beta = 0.1102 * (atten_db - 8.7);
  else if (atten_db >= 21.0) beta = 0.5842 * std::pow(atten_db - 21.0, 0.4) + 0.07886 * (atten_db - 21.0);
  const double i0b = bessel_i0(beta);
  for (int n = 0; n < L; ++n) {
    const double t = ((double)n - (double)L / 2.0) / (double)M;
    const double s = (t == 0.0) ? 1.0 : std::sin(pi * t) / (pi * t);
    const double r = 2.0 * (double)n / (double)L - 1.0;
    const double w = bessel_i0(beta * std::sqrt(std::fmax(0.0, 1.0 - r * r))) / i0b;
    taps[n] = (float)(s / (double)M * w);
  }
  return PFB_OK;
}

int pfb_create(const pfb_config* cfg, pfb_handle** out) {
  return pfb::abi_guard([&]() -> int {
  if (!cfg || !out) return PFB_ERR_BAD_ARG;
  *out = nullptr;
  if (cfg->struct_size != sizeof(pfb_config) || !cfg->taps) return PFB_ERR_BAD_ARG;
  const uint32_t M = cfg->num_channels, P_given = cfg->taps_per_channel;
  const uint32_t D = cfg->decimation ? cfg->decimation : M;
  if (M < 2 || P_given < 1 || D < 1 || D > M) return PFB_ERR_BAD_ARG;
  if (M > 4096 || P_given > 64) return PFB_ERR_UNSUPPORTED;
  // A prototype with fewer taps per channel than a fused shape has is the same filter with zero taps appended
  // (h[n] = 0 for n >= M * P adds exact zeros to every branch sum), so it runs on that shape's kernel -- which is
  // memory-bound anyway -- instead of falling to the generic one.  From here on the handle simply has P taps per
  // channel (history, state blob and halo sizes follow).
  uint32_t P = P_given;
  if (cfg->sample_format <= PFB_FMT_CF32 && !pfb::find_fast_kernel((int)M, (int)P, (int)D, (int)cfg->sample_format)) {
    for (uint32_t p2 = P_given + 1; p2 <= 16; ++p2)
      if (pfb::find_fast_kernel((int)M, (int)p2, (int)D, (int)cfg->sample_format)) { P = p2; break; }
  }
  if (cfg->sample_format > PFB_FMT_CF32) return PFB_ERR_BAD_FORMAT;
  if (cfg->output_layout > PFB_LAYOUT_CHANNEL_MAJOR) return PFB_ERR_BAD_ARG;
  if ((cfg->flags & PFB_FLAG_POWER) && !(cfg->flags & PFB_FLAG_MAGNITUDE)) return PFB_ERR_BAD_ARG;
  int bw = (int)cfg->bit_width;
  if (cfg->sample_format == PFB_FMT_INT8_IQ && (bw < 1 || bw > 8)) return PFB_ERR_BAD_FORMAT;
  if (cfg->sample_format == PFB_FMT_INT16_IQ && (bw < 1 || bw > 16)) return PFB_ERR_BAD_FORMAT;
  if (cfg->sample_format == PFB_FMT_CF32) bw = 1;  // scale 1
  const int off = cfg->input_offset < 0 ? (int)D - 1 : cfg->input_offset;
  if (off >= (int)D) return PFB_ERR_BAD_ARG;

  int ndev = 0;
  const hipError_t ce = hipGetDeviceCount(&ndev);
  if (ce != hipSuccess || ndev <= 0) {
    g_detail = std::string("hipGetDeviceCount: ") + (ce == hipSuccess ? "0 devices" : hipGetErrorString(ce));
    (void)hipGetLastError();
    return PFB_ERR_NO_DEVICE;
  }
  int dev = cfg->device_id;
  if (dev < 0) HIP_TRY(hipGetDevice(&dev));
  if (dev >= ndev) return PFB_ERR_BAD_ARG;

  pfb_handle* h = new (std::nothrow) pfb_handle();
  if (!h) return PFB_ERR_NO_MEMORY;
  h->M = (int)M; h->P = (int)P; h->D = (int)D; h->off = off;
  h->fmt = (int)cfg->sample_format; h->bit_width = bw; h->layout = (int)cfg->output_layout;
  h->flags = cfg->flags;
  h->out_elem = (cfg->flags & PFB_FLAG_MAGNITUDE) ? 4 : 8;
  h->device = dev;
  h->bps = pfb::bytes_per_sample(h->fmt);
  h->hist_samples = (int)(M * P + D);
  // (a channel-major handle takes the shape's default plan too: plans without a channel-major instantiation of
  // their own go through frame-major slabs, see enqueue)
  h->fast = pfb::find_fast_kernel(h->M, h->P, h->D, h->fmt, 0);

  DeviceGuard g(dev);
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) h->num_cus = cus;
    else (void)hipGetLastError();
  }
  const size_t L = (size_t)M * P, L_given = (size_t)M * P_given;
  std::vector<float> taps(L, 0.0f);
  const float scale = std::ldexp(1.0f, -(bw - 1));  // power of two: h*scale is exact
  for (size_t i = 0; i < L_given; ++i) taps[i] = cfg->taps[i] * scale;
  std::vector<float2> tw(M);
  const double two_pi = 6.283185307179586476925286766559;
  for (uint32_t m = 0; m < M; ++m) {
    tw[m].x = (float)std::cos(two_pi * (double)m / (double)M);
    tw[m].y = (float)std::sin(two_pi * (double)m / (double)M);
  }
  const size_t hist_bytes = (size_t)h->hist_samples * h->bps;
  hipError_t e = hipMalloc((void**)&h->d_taps, L * sizeof(float));
  if (e == hipSuccess) e = hipMalloc((void**)&h->d_tw, M * sizeof(float2));
  if (e == hipSuccess) e = hipMalloc(&h->d_hist[0], hist_bytes);
  if (e == hipSuccess) e = hipMalloc(&h->d_hist[1], hist_bytes);
  if (e == hipSuccess) e = hipMemcpy(h->d_taps, taps.data(), L * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(h->d_tw, tw.data(), M * sizeof(float2), hipMemcpyHostToDevice);
  if (e == hipSuccess && h->fast) {
    e = hipMalloc((void**)&h->d_taps_lane, (size_t)h->fast->taps_lane_floats * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&h->d_tw_lane, (size_t)h->fast->tw_lane_elems * sizeof(float2));
    if (e == hipSuccess) e = h->fast->init_tables(h->d_taps, h->d_tw, h->d_taps_lane, h->d_tw_lane, nullptr);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
  }
  if (e == hipSuccess) e = hipMemset(h->d_hist[0], 0, hist_bytes);
  if (e == hipSuccess) e = hipMemset(h->d_hist[1], 0, hist_bytes);
  if (e != hipSuccess) {
    const int rc = hip_fail(e, "pfb_create allocation");
    free_handle(h);
    return rc;
  }
  *out = h;
  return PFB_OK;
  });
}

int pfb_destroy(pfb_handle* h) {
  return pfb::abi_guard([&]() -> int {
  if (!h) return PFB_ERR_BAD_ARG;
  free_handle(h);
  return PFB_OK;
  });
}

int pfb_reset(pfb_handle* h) {
  return pfb::abi_guard([&]() -> int {
  if (!h) return PFB_ERR_BAD_ARG;
  DeviceGuard g(h->device);
  const size_t hist_bytes = (size_t)h->hist_samples * h->bps;
  HIP_TRY(hipMemsetAsync(h->d_hist[h->cur], 0, hist_bytes, h->stream));
  h->phase = 0;
  h->frame_index = 0;
  return PFB_OK;
  });
}

int pfb_set_stream(pfb_handle* h, void* hip_stream) {
  return pfb::abi_guard([&]() -> int {
  if (!h) return PFB_ERR_BAD_ARG;
  hipStream_t next = static_cast<hipStream_t>(hip_stream);
  if (next != h->stream) {  // what the old stream still has queued for this handle (kernels, the history update) comes first
    DeviceGuard g(h->device);
    if (!h->ev_switch) HIP_TRY(hipEventCreateWithFlags(&h->ev_switch, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(h->ev_switch, h->stream));
    HIP_TRY(hipStreamWaitEvent(next, h->ev_switch, 0));
    h->stream = next;
  }
  return PFB_OK;
  });
}

int pfb_frames_for(const pfb_handle* h, uint64_t n, uint64_t* frames_out) {
  if (!h || !frames_out) return PFB_ERR_BAD_ARG;
  *frames_out = frames_for(h, n);
  return PFB_OK;
}

int pfb_process_async(pfb_handle* h, const void* d_iq, uint64_t n, void* d_out, uint64_t cap,
                      uint64_t* frames_out) {
  return pfb::abi_guard([&]() -> int {
  if (!h || (n > 0 && !d_iq)) return PFB_ERR_BAD_ARG;
  const uint64_t f = frames_for(h, n);
  if (frames_out) *frames_out = f;
  if (f > cap) return PFB_ERR_CAPACITY;
  if (f > 0 && !d_out) return PFB_ERR_BAD_ARG;
  DeviceGuard g(h->device);
  return enqueue(h, d_iq, n, d_out, f, (int64_t)f, 0);
  });
}

int pfb_sync(pfb_handle* h) {
  return pfb::abi_guard([&]() -> int {
  if (!h) return PFB_ERR_BAD_ARG;
  DeviceGuard g(h->device);
  HIP_TRY(hipStreamSynchronize(h->stream));
  return PFB_OK;
  });
}

int pfb_process(pfb_handle* h, const void* iq, uint64_t n, void* out, uint64_t cap, uint64_t* frames_out,
                uint32_t mem) {
  return pfb::abi_guard([&]() -> int {
  if (!h || (n > 0 && !iq) || mem > PFB_MEM_DEVICE) return PFB_ERR_BAD_ARG;
  const uint64_t f = frames_for(h, n);
  if (frames_out) *frames_out = f;
  if (f > cap) return PFB_ERR_CAPACITY;
  if (f > 0 && !out) return PFB_ERR_BAD_ARG;
  DeviceGuard g(h->device);
  if (mem == PFB_MEM_DEVICE) {
    const int rc = enqueue(h, iq, n, out, f, (int64_t)f, 0);
    if (rc != PFB_OK) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    return PFB_OK;
  }
  return process_host(h, iq, n, out, f, f, 0);
  });
}

namespace {

// read exactly `bytes` at `offset` of fd into dst with `nthreads` concurrent pread streams (one thread copies out
// of the page cache at ~7 GB/s, less than PCIe takes; four keep the link busy)
bool pread_parallel(int fd, char* dst, size_t bytes, off_t offset, int nthreads) {
  auto read_range = [fd](char* d, size_t len, off_t off) {
    while (len > 0) {
      const ssize_t r = ::pread(fd, d, len, off);
      if (r <= 0) return false;
      d += r; len -= (size_t)r; off += r;
    }
    return true;
  };
  if (nthreads <= 1 || bytes < ((size_t)4 << 20)) return read_range(dst, bytes, offset);
  const size_t part = ((bytes / (size_t)nthreads) + 4095) & ~(size_t)4095;
  std::vector<std::thread> th;
  std::vector<char> ok((size_t)nthreads, 1);
  for (int t = 0; t < nthreads; ++t) {
    const size_t lo = std::min(bytes, part * (size_t)t), hi = std::min(bytes, part * (size_t)(t + 1));
    if (hi > lo) th.emplace_back([&, t, lo, hi] { ok[(size_t)t] = read_range(dst + lo, hi - lo, offset + (off_t)lo); });
  }
  for (auto& x : th) x.join();
  for (char c : ok) if (!c) return false;
  return true;
}

}  // namespace

namespace {

// open a record, parse and check its header (payload length; format and bit width against the handle, if any)
int open_record(pfb_handle* h, const char* path, int* fd_out, pfb_iq_info* info) {
  const int fd = ::open(path, O_RDONLY);
  if (fd < 0) { g_detail = std::string("cannot open ") + path; return PFB_ERR_BAD_ARG; }
  unsigned char head[PFB_IQ_HEADER_BYTES];
  const ssize_t got = ::pread(fd, head, sizeof(head), 0);
  int rc = pfb_iq_parse_header(head, got > 0 ? (size_t)got : 0, info);
  if (rc == PFB_OK && h && ((int)info->sample_format != h->fmt || (int)info->packet.bitWidth != h->bit_width))
    rc = PFB_ERR_BAD_FORMAT;  // the handle's scale / unpack would not match this record
  if (rc == PFB_OK) {
    struct stat st;
    const long long size = ::fstat(fd, &st) == 0 ? (long long)st.st_size : -1;
    if (size - (long long)info->header_bytes != (long long)info->packet.numSamples * (long long)info->bytes_per_sample)
      rc = PFB_ERR_BAD_FORMAT;
  }
  if (rc != PFB_OK) { ::close(fd); return rc; }
  *fd_out = fd;
  return PFB_OK;
}

// Stream the record's payload through the channelizer.  Two page-locked chunk buffers: reader threads fill one
// from the file while the other crosses PCIe and is transformed (the staged host path overlaps its own copy-in /
// transform / copy-out underneath).  device_out: `out` is device memory and nothing comes back.
int stream_record(pfb_handle* h, int fd, const pfb_iq_info& info, void* out, uint64_t need, bool device_out,
                  uint64_t* frames_out) {
  const uint64_t n = info.packet.numSamples;
  const uint64_t chunk = (((uint64_t)1 << 24) / h->D) * h->D;  // whole frames, 64 MB of int16 I/Q
  const size_t chunk_bytes = (size_t)std::min<uint64_t>(chunk, n ? n : 1) * h->bps;
  char* bufs[2] = {static_cast<char*>(pfb_host_alloc(chunk_bytes)), static_cast<char*>(pfb_host_alloc(chunk_bytes))};
  if (!bufs[0] || !bufs[1]) {
    pfb_host_free(bufs[0]);
    pfb_host_free(bufs[1]);
    return PFB_ERR_NO_MEMORY;
  }
  const unsigned hw = std::thread::hardware_concurrency();
  const int readers = (int)std::max(1u, std::min(4u, hw ? hw / 2 : 1u));
  auto read_chunk = [&](char* dst, uint64_t first, uint64_t m) {
    return pread_parallel(fd, dst, (size_t)m * h->bps, (off_t)info.header_bytes + (off_t)(first * h->bps), readers);
  };
  // four staging steps per chunk, so that the copy-in / transform / copy-out pipeline of the host path has stages
  // to overlap inside each call (measured: 3.3 -> 3.7 GS/s; larger file chunks lose more at the ends than they gain)
  const int64_t user_host_chunk = h->opt_host_chunk;
  if (user_host_chunk <= 0) h->opt_host_chunk = (int64_t)1 << 22;
  int rc = PFB_OK;
  uint64_t done = 0, frames_done = 0;
  bool have = n > 0 && read_chunk(bufs[0], 0, std::min<uint64_t>(chunk, n));
  if (n > 0 && !have) rc = PFB_ERR_BAD_FORMAT;
  for (uint64_t i = 0; done < n && rc == PFB_OK; ++i) {
    const uint64_t m = std::min<uint64_t>(chunk, n - done);
    const uint64_t m_next = std::min<uint64_t>(chunk, n - done - m);
    bool next_ok = true;
    std::thread reader;
    if (m_next > 0) reader = std::thread([&, i, m_next] { next_ok = read_chunk(bufs[(i + 1) & 1], done + m, m_next); });
    const uint64_t fr = frames_for(h, m);
    {
      DeviceGuard g(h->device);
      if (h->layout == PFB_LAYOUT_FRAME_MAJOR)
        rc = process_host(h, bufs[i & 1], m, static_cast<char*>(out) + frames_done * h->M * h->out_elem, fr, fr, 0, device_out);
      else  // channel-major: one M x `need` matrix for the whole record, this chunk fills rows frames_done...
        rc = process_host(h, bufs[i & 1], m, out, fr, need, frames_done, device_out);
    }
    if (reader.joinable()) reader.join();
    if (rc == PFB_OK && !next_ok) rc = PFB_ERR_BAD_FORMAT;
    done += m;
    frames_done += fr;
  }
  h->opt_host_chunk = user_host_chunk;
  pfb_host_free(bufs[0]);
  pfb_host_free(bufs[1]);
  if (frames_out) *frames_out = frames_done;
  return rc;
}

}  // namespace

int pfb_process_iq_file(pfb_handle* h, const char* path, void* out, uint64_t cap, uint64_t* frames_out,
                        pfb_iq_info* info_out) {
  return pfb::abi_guard([&]() -> int {
  if (!h || !path) return PFB_ERR_BAD_ARG;
  int fd = -1;
  pfb_iq_info info{};  // zeroed: copied out below even when the record could not be opened
  int rc = open_record(h, path, &fd, &info);
  if (info_out) *info_out = info;
  if (rc != PFB_OK) return rc;
  const uint64_t need = frames_for(h, info.packet.numSamples);
  if (frames_out) *frames_out = need;
  if (need > cap) { ::close(fd); return PFB_ERR_CAPACITY; }
  if (need > 0 && !out) { ::close(fd); return PFB_ERR_BAD_ARG; }
  rc = stream_record(h, fd, info, out, need, false, frames_out);
  ::close(fd);
  return rc;
  });
}

int pfb_pdw_from_iq_file(pfb_handle* h, const char* path, double snr_threshold_db, uint32_t pdw_flags, pfb_pdw* out,
                         uint64_t capacity, uint64_t* count, double* noise_floor_out, pfb_iq_info* info_out) {
  return pfb::abi_guard([&]() -> int {
  if (!h || !path || !count || (capacity > 0 && !out)) return PFB_ERR_BAD_ARG;
  // the PDW stage reads a frame-major complex matrix (mag = abs(iq), phase = angle(iq), :67-68)
  if (h->layout != PFB_LAYOUT_FRAME_MAJOR || (h->flags & PFB_FLAG_MAGNITUDE)) return PFB_ERR_UNSUPPORTED;
  int fd = -1;
  pfb_iq_info info{};  // zeroed: copied out below even when the record could not be opened
  int rc = open_record(h, path, &fd, &info);
  if (info_out) *info_out = info;
  if (rc != PFB_OK) return rc;
  const uint64_t need = frames_for(h, info.packet.numSamples);
  const size_t bytes = (size_t)std::max<uint64_t>(need, 1) * h->M * sizeof(float2);
  {
    DeviceGuard g(h->device);
    if (bytes > h->matrix_bytes) {
      if (hipStreamSynchronize(h->stream) != hipSuccess) { ::close(fd); return PFB_ERR_HIP; }
      (void)hipFree(h->d_matrix);
      h->d_matrix = nullptr; h->matrix_bytes = 0;
      if (hipMalloc(&h->d_matrix, bytes) != hipSuccess) { (void)hipGetLastError(); ::close(fd); return PFB_ERR_NO_MEMORY; }
      h->matrix_bytes = bytes;
    }
  }
  uint64_t frames = 0;
  rc = stream_record(h, fd, info, h->d_matrix, need, true, &frames);
  ::close(fd);
  if (rc != PFB_OK) return rc;
  if (frames == 0) {  // a record shorter than one frame holds no pulse
    *count = 0;
    return PFB_OK;
  }
  // the kernels are queued on the handle's stream; the extraction runs behind them on the same stream
  return pfb_pdw_extract(h->d_matrix, frames, (uint32_t)h->M, (uint32_t)h->D, (double)info.packet.sampleRateSps,
                         (double)info.packet.frequencyHz, info.packet.sampleStartTime, snr_threshold_db, pdw_flags, out,
                         capacity, count, noise_floor_out, PFB_MEM_DEVICE, h->device, h->stream);
  });
}

int pfb_pdw_raw_from_iq_file(const char* path, double snr_threshold_db, double trailing_threshold_db, pfb_pdw* out,
                             uint64_t capacity, uint64_t* count, double* noise_floor_out, pfb_iq_info* info_out,
                             int32_t device_id) {
  return pfb::abi_guard([&]() -> int {
  if (!path || !count || (capacity > 0 && !out)) return PFB_ERR_BAD_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); return PFB_ERR_NO_DEVICE; }
  int dev = device_id;
  if (dev < 0) HIP_TRY(hipGetDevice(&dev));
  if (dev >= ndev) return PFB_ERR_BAD_ARG;
  int fd = -1;
  pfb_iq_info info{};
  int rc = open_record(nullptr, path, &fd, &info);
  if (info_out) *info_out = info;
  if (rc != PFB_OK) return rc;
  DeviceGuard g(dev);
  const uint64_t n = info.packet.numSamples;
  const size_t bps = info.bytes_per_sample;
  if (n == 0) { ::close(fd); return PFB_ERR_BAD_ARG; }
  // the record goes to the device through two page-locked 64 MB buffers: four pread streams fill one while the
  // other crosses PCIe; the extraction then runs on the device-resident stream
  void* d_iq = nullptr;
  hipStream_t st = nullptr;
  const size_t chunk = (size_t)64 << 20;
  char* bufs[2] = {static_cast<char*>(pfb_host_alloc(std::min(chunk, n * bps))),
                   static_cast<char*>(pfb_host_alloc(std::min(chunk, n * bps)))};
  auto cleanup = [&] {
    pfb_host_free(bufs[0]);
    pfb_host_free(bufs[1]);
    if (st) (void)hipStreamDestroy(st);
    (void)hipFree(d_iq);
    ::close(fd);
  };
  if (!bufs[0] || !bufs[1] || hipMalloc(&d_iq, n * bps) != hipSuccess || hipStreamCreate(&st) != hipSuccess) {
    (void)hipGetLastError();
    cleanup();
    return PFB_ERR_NO_MEMORY;
  }
  const unsigned hw = std::thread::hardware_concurrency();
  const int readers = (int)std::max(1u, std::min(4u, hw ? hw / 2 : 1u));
  const size_t total = n * bps;
  auto read_chunk = [&](char* dst, size_t first, size_t len) {
    return pread_parallel(fd, dst, len, (off_t)info.header_bytes + (off_t)first, readers);
  };
  bool ok = read_chunk(bufs[0], 0, std::min(chunk, total));
  size_t done = 0;
  for (size_t i = 0; done < total && ok; ++i) {
    const size_t len = std::min(chunk, total - done), next = std::min(chunk, total - done - len);
    bool next_ok = true;
    std::thread reader;
    if (next > 0) reader = std::thread([&, i, next] { next_ok = read_chunk(bufs[(i + 1) & 1], done + len, next); });
    const hipError_t e = hipMemcpyAsync(static_cast<char*>(d_iq) + done, bufs[i & 1], len, hipMemcpyHostToDevice, st);
    const hipError_t e2 = hipStreamSynchronize(st);
    if (reader.joinable()) reader.join();
    if (e != hipSuccess || e2 != hipSuccess) { cleanup(); return hip_fail(e != hipSuccess ? e : e2, "pfb_pdw_raw_from_iq_file"); }
    ok = next_ok;
    done += len;
  }
  if (!ok) { cleanup(); return PFB_ERR_BAD_FORMAT; }
  rc = pfb_pdw_extract_raw(d_iq, n, info.sample_format, info.packet.bitWidth, (double)info.packet.sampleRateSps,
                           (double)info.packet.frequencyHz, info.packet.sampleStartTime, snr_threshold_db,
                           trailing_threshold_db, out, capacity, count, noise_floor_out, PFB_MEM_DEVICE, dev, st);
  cleanup();
  return rc;
  });
}

uint64_t pfb_history_samples(const pfb_handle* h) { return h ? (uint64_t)h->hist_samples : 0; }

int pfb_prime(pfb_handle* h, const void* iq, uint64_t n, uint32_t mem) {
  return pfb::abi_guard([&]() -> int {
  if (!h || (n > 0 && !iq) || mem > PFB_MEM_DEVICE) return PFB_ERR_BAD_ARG;
  if (n == 0) return PFB_OK;
  DeviceGuard g(h->device);
  const uint64_t f = frames_for(h, n);
  // only the trailing hist_samples matter
  const uint64_t keep = std::min<uint64_t>(n, (uint64_t)h->hist_samples);
  const char* tail = static_cast<const char*>(iq) + (n - keep) * h->bps;
  const void* d_tail = tail;
  if (mem == PFB_MEM_HOST) {
    const int rc = ensure_stage(h, (size_t)keep * h->bps, 0);
    if (rc != PFB_OK) return rc;
    HIP_TRY(hipMemcpyAsync(h->d_stage_in, tail, (size_t)keep * h->bps, hipMemcpyHostToDevice, h->stream));
    d_tail = h->d_stage_in;
  }
  HIP_TRY(pfb::launch_update_history(h->d_hist[h->cur], d_tail, (long long)keep, h->d_hist[h->cur ^ 1],
                                     h->hist_samples, h->bps, h->stream));
  h->cur ^= 1;
  h->phase = (uint32_t)((h->phase + n) % (uint64_t)h->D);
  h->frame_index += f;
  if (mem == PFB_MEM_HOST) HIP_TRY(hipStreamSynchronize(h->stream));
  return PFB_OK;
  });
}

uint64_t pfb_halo_samples(const pfb_handle* h) { return h ? (uint64_t)(h->M * h->P - 1 - h->off) : 0; }

uint64_t pfb_shard_head_frames(const pfb_handle* h) {
  return h ? (uint64_t)((h->hist_samples + h->D - 1) / h->D) : 0;
}

void* pfb_halo_recv_buffer(pfb_handle* h) {
  if (!h || !h->d_halo) return nullptr;
  return static_cast<char*>(h->d_halo) + ((size_t)h->hist_samples - (size_t)pfb_halo_samples(h)) * h->bps;
}

int pfb_shard_attach(pfb_handle* h, const pfb_shard_config* cfg) {
  return pfb::abi_guard([&]() -> int {
  if (!h || !cfg || cfg->struct_size != sizeof(pfb_shard_config)) return PFB_ERR_BAD_ARG;
  if (cfg->world < 1 || cfg->rank < 0 || cfg->rank >= cfg->world) return PFB_ERR_BAD_ARG;
  if (cfg->world > 1 && !cfg->exchange) return PFB_ERR_BAD_ARG;
  DeviceGuard g(h->device);
  if (!h->d_halo) {
    // sized like the history so the kernels index it the same way; only its last pfb_halo_samples() are ever read
    const size_t bytes = (size_t)h->hist_samples * h->bps;
    HIP_TRY(hipMalloc(&h->d_halo, bytes));
    HIP_TRY(hipMemset(h->d_halo, 0, bytes));
    HIP_TRY(hipStreamCreateWithFlags(&h->s_halo, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&h->ev_seg, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&h->ev_halo, hipEventDisableTiming));
  }
  h->shard_rank = cfg->rank;
  h->shard_world = cfg->world;
  h->shard_ring = cfg->ring != 0;
  h->shard_exchange = cfg->exchange;
  h->shard_user = cfg->user;
  return PFB_OK;
  });
}

int pfb_process_shard_async(pfb_handle* h, const void* d_seg, uint64_t n, void* d_out, uint64_t cap, uint64_t* frames_out) {
  return pfb::abi_guard([&]() -> int {
  if (!h || !d_seg) return PFB_ERR_BAD_ARG;
  // a shard is cut on frame boundaries: whole frames in, no carried tail before or after
  if (h->phase != 0 || n % (uint64_t)h->D != 0) return PFB_ERR_BAD_ARG;
  const uint64_t F = n / (uint64_t)h->D, head = pfb_shard_head_frames(h);
  if (frames_out) *frames_out = F;
  // a segment shorter than its head frames has no interior: everything waits for the halo; it must still hold the tail
  // the next shard needs (and the history the handle keeps)
  if (F == 0 || n < (uint64_t)h->hist_samples) return PFB_ERR_BAD_ARG;
  if (F > cap) return PFB_ERR_CAPACITY;
  if (!d_out) return PFB_ERR_BAD_ARG;
  DeviceGuard g(h->device);
  const int world = h->shard_world, rank = h->shard_rank;
  const bool receiving = world > 1 && (h->shard_ring || rank > 0);
  const bool sending = world > 1 && (h->shard_ring || rank + 1 < world);
  const size_t halo_bytes = (size_t)pfb_halo_samples(h) * h->bps;
  if (sending || receiving) {
    // side stream: the exchange starts once everything queued so far on the handle's stream (the caller's writes of
    // the segment, the previous call's head frames that still read the landing zone) has finished
    HIP_TRY(hipEventRecord(h->ev_seg, h->stream));
    HIP_TRY(hipStreamWaitEvent(h->s_halo, h->ev_seg, 0));
    // What I pass on: the tail of THIS segment -- except the last rank of a ring, whose successor (rank 0) works on the
    // NEXT call's first segment: it gets the tail of my PREVIOUS segment, i.e. my carried state (zeros after a reset, so
    // the very first segment of the stream starts from zero state like a fresh dsp.Channelizer).  A matched transport
    // pairs rank 0's receive in call i with my send in call i; sending the current tail there would hand rank 0 samples
    // from its own future.
    const bool pass_state = h->shard_ring && rank == world - 1;
    const char* tail = pass_state
                           ? static_cast<const char*>(h->d_hist[h->cur]) + ((size_t)h->hist_samples * h->bps - halo_bytes)
                           : static_cast<const char*>(d_seg) + (size_t)n * h->bps - halo_bytes;
    const int crc = h->shard_exchange(h->shard_user, sending ? tail : nullptr, receiving ? pfb_halo_recv_buffer(h) : nullptr,
                                      halo_bytes, sending ? (rank + 1) % world : -1,
                                      receiving ? (rank + world - 1) % world : -1, h->s_halo);
    if (crc != 0) {
      char buf[96];
      std::snprintf(buf, sizeof(buf), "halo exchange callback returned %d", crc);
      g_detail = buf;
      return PFB_ERR_COMM;
    }
    HIP_TRY(hipEventRecord(h->ev_halo, h->s_halo));
  }
  // main stream: every frame whose window lies inside the segment starts now ...
  const uint64_t nhead = head < F ? head : F;
  int rc = launch_frames(h, d_seg, n, nullptr, d_out, nhead, F, (int64_t)F, 0);
  if (rc != PFB_OK) return rc;
  // ... the head frames once the halo has landed (a wait on the GPU, not on the host); a shard that receives
  // nothing continues from the handle's own state.  Waiting for the event also puts the SEND in front of
  // whatever the caller queues next on this stream, so pfb_sync() covers both transfers.
  if (sending || receiving) HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_halo, 0));
  rc = launch_frames(h, d_seg, n, receiving ? h->d_halo : h->d_hist[h->cur], d_out, 0, nhead, (int64_t)F, 0);
  if (rc != PFB_OK) return rc;
  return advance_state(h, d_seg, n, F);
  });
}

int pfb_get_state(pfb_handle* h, void* buf, size_t* bytes) {
  return pfb::abi_guard([&]() -> int {
  if (!h || !bytes) return PFB_ERR_BAD_ARG;
  const size_t hist_bytes = (size_t)h->hist_samples * h->bps;
  const size_t need = sizeof(StateHeader) + hist_bytes;
  if (!buf) { *bytes = need; return PFB_OK; }
  if (*bytes < need) { *bytes = need; return PFB_ERR_CAPACITY; }
  DeviceGuard g(h->device);
  StateHeader sh{kStateMagic, (uint32_t)h->M, (uint32_t)h->P, (uint32_t)h->D, (uint32_t)h->fmt,
                 (uint32_t)h->hist_samples, h->phase, 0u, h->frame_index};
  std::memcpy(buf, &sh, sizeof(sh));
  HIP_TRY(hipMemcpyAsync(static_cast<char*>(buf) + sizeof(sh), h->d_hist[h->cur], hist_bytes, hipMemcpyDeviceToHost,
                         h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  *bytes = need;
  return PFB_OK;
  });
}

int pfb_set_state(pfb_handle* h, const void* buf, size_t bytes) {
  return pfb::abi_guard([&]() -> int {
  if (!h || !buf || bytes < sizeof(StateHeader)) return PFB_ERR_BAD_ARG;
  StateHeader sh;
  std::memcpy(&sh, buf, sizeof(sh));
  const size_t hist_bytes = (size_t)h->hist_samples * h->bps;
  if (sh.magic != kStateMagic || sh.M != (uint32_t)h->M || sh.P != (uint32_t)h->P || sh.D != (uint32_t)h->D ||
      sh.fmt != (uint32_t)h->fmt || sh.hist_samples != (uint32_t)h->hist_samples || sh.phase >= (uint32_t)h->D ||
      bytes < sizeof(sh) + hist_bytes)
    return PFB_ERR_BAD_ARG;
  DeviceGuard g(h->device);
  HIP_TRY(hipMemcpyAsync(h->d_hist[h->cur], static_cast<const char*>(buf) + sizeof(sh), hist_bytes,
                         hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->phase = sh.phase;
  h->frame_index = sh.frame_index;
  return PFB_OK;
  });
}

int pfb_set_frame_index(pfb_handle* h, uint64_t next_frame) {
  if (!h) return PFB_ERR_BAD_ARG;
  h->frame_index = next_frame;
  return PFB_OK;
}

int pfb_get_frame_index(const pfb_handle* h, uint64_t* next_frame) {
  if (!h || !next_frame) return PFB_ERR_BAD_ARG;
  *next_frame = h->frame_index;
  return PFB_OK;
}

int pfb_set_option(pfb_handle* h, int option, int64_t value) {
  return pfb::abi_guard([&]() -> int {
  if (!h) return PFB_ERR_BAD_ARG;
  switch (option) {
    case PFB_OPT_KERNEL:
      if (value < 0 || value > 2) return PFB_ERR_BAD_ARG;
      h->opt_kernel = (int)value;
      return PFB_OK;
    case PFB_OPT_FRAMES_PER_BLOCK:
      if (value < 0 || value > (1 << 24)) return PFB_ERR_BAD_ARG;
      h->opt_frames_per_block = (int)value;
      return PFB_OK;
    case PFB_OPT_HOST_CHUNK_SAMPLES:
      if (value < 0) return PFB_ERR_BAD_ARG;
      h->opt_host_chunk = value;
      return PFB_OK;
    case PFB_OPT_NONTEMPORAL:
      h->opt_nontemporal = value ? 1 : 0;
      return PFB_OK;
    case PFB_OPT_SCHEDULE:
      if (value < -1 || value > 13 || value == 1 || value == 5 || value == 10 || value == 12) return PFB_ERR_BAD_ARG;  // (removed studies)
      h->opt_schedule = (int)value;
      return PFB_OK;
    case PFB_OPT_TILE_WAVES:
      if (value < 1 || value > 16) return PFB_ERR_BAD_ARG;
      h->opt_tile_waves = (int)value;
      return PFB_OK;
    case PFB_OPT_GRID:
      if (value < 0 || value > (1 << 22)) return PFB_ERR_BAD_ARG;
      h->opt_grid = (int)value;
      return PFB_OK;
    case PFB_OPT_XCD_REMAP:
      if (value < -1 || value > (1 << 20)) return PFB_ERR_BAD_ARG;
      h->opt_xcd_remap = (int)value;
      return PFB_OK;
    case PFB_OPT_EXPERIMENT:
      if (value < 0 || value > 0xffff) return PFB_ERR_BAD_ARG;
      h->opt_experiment = (int)value;
      return PFB_OK;
    case PFB_OPT_SLAB_FRAMES:
      if (value < 0 || value > (1ll << 32)) return PFB_ERR_BAD_ARG;
      h->opt_slab_frames = value;
      return PFB_OK;
    case PFB_OPT_VARIANT: {
      if (value < 0 || value > 16) return PFB_ERR_BAD_ARG;
      if ((int)value == h->opt_variant) return PFB_OK;
      const pfb::FastKernelInfo* f =
          pfb::find_fast_kernel(h->M, h->P, h->D, h->fmt, (int)value);
      if (!f) return PFB_ERR_UNSUPPORTED;
      DeviceGuard g(h->device);
      HIP_TRY(hipStreamSynchronize(h->stream));  // the old tables may still be in use
      float* tl = nullptr;
      float2* tw = nullptr;
      hipError_t e = hipMalloc((void**)&tl, (size_t)f->taps_lane_floats * sizeof(float));
      if (e == hipSuccess) e = hipMalloc((void**)&tw, (size_t)f->tw_lane_elems * sizeof(float2));
      if (e == hipSuccess) e = f->init_tables(h->d_taps, h->d_tw, tl, tw, nullptr);
      if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
      if (e != hipSuccess) {
        (void)hipFree(tl);
        (void)hipFree(tw);
        return hip_fail(e, "pfb_set_option(PFB_OPT_VARIANT)");
      }
      (void)hipFree(h->d_taps_lane);
      (void)hipFree(h->d_tw_lane);
      h->d_taps_lane = tl;
      h->d_tw_lane = tw;
      h->fast = f;
      h->opt_variant = (int)value;
      return PFB_OK;
    }
    case PFB_OPT_PROFILE:
      h->opt_profile = value ? 1 : 0;
      h->ev_used = 0;
      return PFB_OK;
    default:
      return PFB_ERR_BAD_ARG;
  }
  });
}

const char* pfb_last_kernel(const pfb_handle* h) { return h ? h->last_kernel : ""; }

int pfb_get_device(const pfb_handle* h, int* device_id) {
  if (!h || !device_id) return PFB_ERR_BAD_ARG;
  *device_id = h->device;
  return PFB_OK;
}

int pfb_get_kernel_times(pfb_handle* h, float* ms_out, int capacity, int* count) {
  return pfb::abi_guard([&]() -> int {
  if (!h || !count || (capacity > 0 && !ms_out)) return PFB_ERR_BAD_ARG;
  DeviceGuard g(h->device);
  HIP_TRY(hipStreamSynchronize(h->stream));
  int n = 0;
  for (size_t i = 0; i < h->ev_used && n < capacity; ++i, ++n)
    HIP_TRY(hipEventElapsedTime(&ms_out[n], h->ev_pool[i].first, h->ev_pool[i].second));
  h->ev_used = 0;
  *count = n;
  return PFB_OK;
  });
}

void* pfb_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return p;
}

void pfb_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

int pfb_measure_stream_copy(int device_id, uint64_t bytes_in, int iters, double* bytes_per_sec) {
  return pfb::abi_guard([&]() -> int {
  if (!bytes_per_sec || iters < 1 || bytes_in < 512) return PFB_ERR_BAD_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    (void)hipGetLastError();
    return PFB_ERR_NO_DEVICE;
  }
  int dev = device_id;
  if (dev < 0) HIP_TRY(hipGetDevice(&dev));
  DeviceGuard g(dev);
  const long long nvec = (long long)(bytes_in / 512) * 32;  // whole row pairs (512 bytes of input per wave)
  void *in = nullptr, *out = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = PFB_OK;
  float ms = 0.f;
  hipError_t e = hipMalloc(&in, (size_t)nvec * 16);
  if (e == hipSuccess) e = hipMalloc(&out, (size_t)nvec * 32);
  if (e == hipSuccess) e = hipMemset(in, 1, (size_t)nvec * 16);
  if (e == hipSuccess) e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  if (e == hipSuccess) e = pfb::launch_stream_copy(in, out, nvec, nullptr);  // warm-up
  if (e == hipSuccess) e = hipEventRecord(e0, nullptr);
  for (int i = 0; i < iters && e == hipSuccess; ++i) e = pfb::launch_stream_copy(in, out, nvec, nullptr);
  if (e == hipSuccess) e = hipEventRecord(e1, nullptr);
  if (e == hipSuccess) e = hipEventSynchronize(e1);
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  if (e != hipSuccess) rc = hip_fail(e, "pfb_measure_stream_copy");
  else *bytes_per_sec = (double)nvec * 48.0 * iters / ((double)ms * 1e-3);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(in);
  (void)hipFree(out);
  return rc;
  });
}

int pfb_measure_mix_copy(int device_id, uint64_t bytes_in, uint32_t write_ratio, uint32_t rows_per_wave, int iters,
                         double* bytes_per_sec) {
  return pfb::abi_guard([&]() -> int {
  if (!bytes_per_sec || iters < 1 || (write_ratio != 2 && write_ratio != 4) || rows_per_wave < 2 || (rows_per_wave & 1))
    return PFB_ERR_BAD_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    (void)hipGetLastError();
    return PFB_ERR_NO_DEVICE;
  }
  int dev = device_id;
  if (dev < 0) HIP_TRY(hipGetDevice(&dev));
  DeviceGuard g(dev);
  const long long rows = (long long)(bytes_in / 256) / (4ll * rows_per_wave) * (4ll * rows_per_wave);  // whole workgroups
  if (rows <= 0) return PFB_ERR_BAD_ARG;
  void *in = nullptr, *out = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = PFB_OK;
  float ms = 0.f;
  hipError_t e = hipMalloc(&in, (size_t)rows * 256);
  if (e == hipSuccess) e = hipMalloc(&out, (size_t)rows * 256 * write_ratio);
  if (e == hipSuccess) e = hipMemset(in, 1, (size_t)rows * 256);
  if (e == hipSuccess) e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  if (e == hipSuccess) e = pfb::launch_mix_copy(in, out, rows, (int)write_ratio, (int)rows_per_wave, nullptr);  // warm-up
  if (e == hipSuccess) e = hipEventRecord(e0, nullptr);
  for (int i = 0; i < iters && e == hipSuccess; ++i) e = pfb::launch_mix_copy(in, out, rows, (int)write_ratio, (int)rows_per_wave, nullptr);
  if (e == hipSuccess) e = hipEventRecord(e1, nullptr);
  if (e == hipSuccess) e = hipEventSynchronize(e1);
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  if (e != hipSuccess) rc = hip_fail(e, "pfb_measure_mix_copy");
  else *bytes_per_sec = (double)rows * 256.0 * (1.0 + write_ratio) * iters / ((double)ms * 1e-3);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(in);
  (void)hipFree(out);
  return rc;
  });
}

}  // extern "C"
