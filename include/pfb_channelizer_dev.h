/* pfb_channelizer_dev.h -- measurement and kernel-study entry points of libpfb_channelizer.
 *
 * NOT part of the drop-in boundary: nothing here replaces a reference interface, and an integrator
 * binds pfb_channelizer.h and pfb_iq_packet.h only.  These are the hooks bench.py, tools/ and the tests use to put a
 * yardstick next to a kernel's rate (copy kernels with the channelizer's byte mix and no arithmetic), to select a
 * registered kernel plan other than the default one for A/B runs, and to prove a property of the ABI itself (no C++
 * exception crosses it).  Same shared library, same C ABI rules (plain pointers and sizes, status codes, never throws).
 */
#ifndef PFB_CHANNELIZER_DEV_H
#define PFB_CHANNELIZER_DEV_H

#include "pfb_channelizer.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Options of pfb_set_option (pfb_channelizer.h) for kernel studies; every combination a kernel accepts produces the
 * same bits as the handle's default plan (tests/test_gpu_parity.py walks them).  All are per handle. */
enum pfb_dev_option {
  PFB_OPT_GRID = 7,        /* schedule 13: resident workgroups to launch, each walking runs b, b + G, ... (0 = one per run) */
  PFB_OPT_TILE_WAVES = 8,  /* schedules 2 / 3 / 8: waves, 4 / 7: wave pairs per workgroup (1 ... 16)                       */
  PFB_OPT_EXPERIMENT = 9,  /* timing experiments, 0 in production.  bit 0: nontemporal row loads in the one-dword-per-lane */
                           /* kernels; bits 8-15: extra dynamic LDS in KiB for schedule 3 (an occupancy throttle).  Neither */
                           /* changes a result bit.                                                                         */
  PFB_OPT_VARIANT = 10     /* n-th fused kernel registered for this shape (0 = the default plan; PFB_ERR_UNSUPPORTED past   */
                           /* the last).  Rebuilds the handle's per-lane tables for the new plan; the filter state stays.   */
};

/* Device stream copy, read 1 : write 2 (configuration 2's traffic: int16 I/Q in, complex64 out), timed with HIP events
 * over `iters` launches after a warm-up: bytes moved per second, the "measured peak" next to the nominal roofline.
 * Allocates bytes_in of input and 2 * bytes_in of output scratch on device_id (-1 = current) for the call.  The kernel
 * is the fastest 1:2 shape tools/membench2 found on MI355X (short-lived 4-wave workgroups, two 256-byte rows per wave,
 * one 16-byte store per lane), so no channelizer kernel with this byte mix should beat the figure. */
int pfb_measure_stream_copy(int device_id, uint64_t bytes_in, int iters, double* bytes_per_sec);

/* The same for either byte mix of the channelizer and a given wave lifetime: a copy kernel with no arithmetic that
 * reads bytes_in and writes write_ratio x bytes_in (2: int16 I/Q -> complex64 at D = M; 4: int8 I/Q, or int16 at
 * D = M/2), every wave owning rows_per_wave consecutive 256-byte rows (even, >= 2).  What the memory system gives a
 * byte mix depends on how short-lived the waves are (0.78 / 0.75 of the nominal 8 TB/s at 2 rows, 0.63 / 0.64 at 512):
 * the bound bench.py prints next to each shape's fraction (roofline.copy_kernel_frac_by_byte_mix). */
int pfb_measure_mix_copy(int device_id, uint64_t bytes_in, uint32_t write_ratio, uint32_t rows_per_wave, int iters,
                         double* bytes_per_sec);

/* Diagnostic: throws a C++ exception of the given kind (0 = std::bad_alloc, 1 = std::runtime_error,
 * 2 = a non-std type) INSIDE the guard every entry point runs under and returns what the guard
 * returns (PFB_ERR_NO_MEMORY / PFB_ERR_INTERNAL): proof that nothing thrown crosses the C ABI. */
int pfb_selftest_exception_guard(int kind);

#ifdef __cplusplus
}
#endif
#endif /* PFB_CHANNELIZER_DEV_H */
