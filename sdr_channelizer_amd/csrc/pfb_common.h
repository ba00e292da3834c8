// pfb_common.h -- shared between the host API (pfb_api.cpp) and the HIP kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <new>

#include "pfb_channelizer.h"

namespace pfb {

// Every extern "C" entry point that can allocate host memory, build a std::string / std::vector, take a mutex
// or start a thread runs its body under this guard: the header promises "never throws", and an exception
// unwinding through a C caller (the recorders' loop, MATLAB's loadlibrary, ctypes) is undefined behaviour.
template <class F>
static inline int abi_guard(F&& body) noexcept {
  try {
    return body();
  } catch (const std::bad_alloc&) {
    return PFB_ERR_NO_MEMORY;
  } catch (...) {
    return PFB_ERR_INTERNAL;
  }
}

// Kernel-side view of one pfb_process call.  "Row r" is the D input samples
// whose newest member is the newest sample of local frame r:
//   sample index of (row r, column c) = r*D + base + c,   c = 0..D-1
// relative to in[0]; negative indices live in the history buffer, whose last
// element hist[hist_samples-1] is the sample just before in[0].
struct KernelParams {
  const void* in;          // this call's samples (device), cfg.sample_format
  const void* hist;        // hist_samples samples of history (device)
  float2* out;             // frames*M complex64
  const float* taps;       // M*P taps, pre-scaled by 2^-(bit_width-1) (exact)
  const float2* tw;        // tw[m] = exp(+j 2 pi m / M), m = 0..M-1 (from float64)
  const float* taps_lane;  // fast kernels: taps_lane[c][j] = taps[(D-1-c) + D*j], rows padded to float4s
  const float2* tw_lane;   // fast kernels: inter-pass twiddle rows, see pfb_fast.hpp
  long long n_in;          // samples in `in`
  long long frames;        // frames to produce
  long long frame0;        // global index of local frame 0 (PFB_FLAG_DEROTATE)
  long long out_ld;        // CHANNEL_MAJOR: column stride in frames
  long long out_frame0;    // CHANNEL_MAJOR: first row this call writes
  int base;                // eff_off - (D-1), eff_off = input_offset - carried
  int hist_samples;        // M*P + D
  int frames_per_block;    // fast kernels: run length per workgroup (multiple of C)
  int vec_ok;              // fast kernels: vector loads are aligned
  int M, P, D;             // generic kernel only
  int fmt;                 // pfb_sample_format (generic kernel only)
  int layout;              // pfb_output_layout
  unsigned flags;          // PFB_FLAG_*
  int nontemporal;         // nontemporal output stores
  int xcd_remap;           // fast kernels: 1 = consecutive runs on one XCD, G>1 = in groups of G
  int experiment;          // bit mask of timing experiments (0 in production)
  int schedule;            // fast kernels: the list at the top of pfb_fast.hpp
  int tile_waves;          // schedules 2 / 3 / 8: waves, 4 / 7: wave pairs per workgroup
  int grid_override;       // schedule 13: workgroups to launch, each walking runs b, b + G, ... (0 = one per run)
};

// sample traits -----------------------------------------------------------------
template <int FMT> struct SampleT;
template <> struct SampleT<PFB_FMT_INT16_IQ> {
  using raw_t = uint32_t;  // {I:int16, Q:int16}, little endian
  static constexpr int kBytes = 4;
  static __device__ __forceinline__ void cvt(raw_t v, float& re, float& im) {
    re = (float)(int16_t)(v & 0xffffu);
    im = (float)((int32_t)v >> 16);
  }
};
template <> struct SampleT<PFB_FMT_INT8_IQ> {
  using raw_t = uint16_t;  // {I:int8, Q:int8}
  static constexpr int kBytes = 2;
  static __device__ __forceinline__ void cvt(raw_t v, float& re, float& im) {
    re = (float)(int8_t)(v & 0xffu);
    im = (float)(int8_t)(v >> 8);
  }
};
template <> struct SampleT<PFB_FMT_CF32> {
  using raw_t = float2;
  static constexpr int kBytes = 8;
  static __device__ __forceinline__ void cvt(raw_t v, float& re, float& im) {
    re = v.x;
    im = v.y;
  }
};

// the float32 a PFB_FLAG_MAGNITUDE handle stores for an output y: |y|, or |y|^2 with PFB_FLAG_POWER
static __device__ __forceinline__ float mag_out(float re, float im, unsigned flags) {
  const float m2 = re * re + im * im;
  return (flags & PFB_FLAG_POWER) ? m2 : sqrtf(m2);
}

static inline int bytes_per_sample(int fmt) {
  return fmt == PFB_FMT_INT8_IQ ? 2 : (fmt == PFB_FMT_INT16_IQ ? 4 : 8);
}

// Launchers implemented in pfb_kernels.hip ------------------------------------------
// Returns nullptr if there is no fast kernel for this configuration.
using FastLaunchFn = hipError_t (*)(const KernelParams&, hipStream_t);
using FastInitFn = hipError_t (*)(const float* taps, const float2* tw, float* taps_lane, float2* tw_lane, hipStream_t);
struct FastKernelInfo {
  FastLaunchFn launch;
  FastInitFn init_tables;
  int taps_lane_floats;
  int tw_lane_elems;
  const char* name;
  int chunk_frames;        // C: frames_per_block must be a multiple of this
  int default_frames_per_block;
  int cols_per_thread;     // CPT: vector-load alignment requirement
  int default_schedule;    // measured best schedule for this instantiation (PFB_OPT_SCHEDULE = -1)
  bool channel_major_ok;   // has a channel-major instantiation
  int magnitude_schedule;  // measured best schedule with PFB_FLAG_MAGNITUDE, -1 = default_schedule
  int threads;             // threads of the plan's FIR workgroup (run-length policy for short calls)
};
const FastKernelInfo* find_fast_kernel(int M, int P, int D, int fmt, int variant = 0, bool channel_major = false);

hipError_t launch_generic(const KernelParams& p, hipStream_t s);
hipError_t launch_update_history(const void* old_hist, const void* in, long long n_in, void* new_hist,
                                 int hist_samples, int bytes_per_sample, hipStream_t s);
hipError_t launch_transpose_slab(const void* slab, long long frames, int M, void* out, long long out_ld,
                                 long long out_frame0, int elem_bytes, hipStream_t s);
hipError_t launch_stream_copy(const void* in, void* out, long long n_vec16, hipStream_t s);
hipError_t launch_mix_copy(const void* in, void* out, long long rows, int write_ratio, int spw, hipStream_t s);

}  // namespace pfb
