// pfb_table.h -- one row of the fused-kernel table; the table is split over four translation units
// (pfb_kernels.hip: M = 64; pfb_kernels_mid.hip: the other tuned single-wave shapes; pfb_kernels_big.hip: M = 1024 and
// 560; pfb_kernels_mixed.hip: the other plausible radio rates, M = 2^a 3^b 5^c 7^d) so that they compile in parallel.
#pragma once
#include "pfb_fast.hpp"

namespace pfb {

struct FastEntry { int M, P, D, fmt; FastKernelInfo info; };

template <class K>
constexpr FastEntry entry(const char* name, int default_fpb, int default_schedule) {
  return FastEntry{K::M, K::P, K::D, K::FMT,
                   FastKernelInfo{&launch_fast<K>, &init_tables<K>, K::TAPS_LANE_FLOATS, K::TW_LANE_ELEMS, name, K::C,
                                  default_fpb, K::CPT, default_schedule, kChannelMajorOk<K>, kMagnitudeSchedule<K>, K::NT}};
}

// small banks: SegKernel, 64 / M segments of the run per wave; frames_per_block is a multiple of C * SEG
template <class K>
constexpr FastEntry seg_entry(const char* name, int default_fpb) {
  return FastEntry{K::M, K::P, K::D, K::FMT,
                   FastKernelInfo{&launch_seg<K>, &init_tables<K>, K::TAPS_LANE_FLOATS, K::TW_LANE_ELEMS, name,
                                  SegKernel<K>::CT, default_fpb, 1, 0, true, -1, 64}};
}

struct FastTablePart { const FastEntry* rows; int count; };
FastTablePart fast_table_m64();
FastTablePart fast_table_mid();
FastTablePart fast_table_big();
FastTablePart fast_table_mixed();

}  // namespace pfb
