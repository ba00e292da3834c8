// pfb_kernels_big.hip -- fused-kernel table, part 3: the multi-wave shapes (cfg4 M = 1024, M = 560) (see pfb_table.h)
#include <hip/hip_runtime.h>

#include "pfb_table.h"

namespace pfb {

//                 M    P   D  CPT FMT               C NP R0 R1 R2 RS0 RS1 RS2 FS  PP     MINW  TW_TABLE
// cfg4: 512 threads x 2 adjacent columns (8-byte loads), 1024 = 16 x 16 x 4 in place in one 68 KB chunk
// buffer (read - barrier - write), twiddles from the L1-resident table, conflict-free padding
using Cfg1024x16i16 =
    FastCfg<1024, 16, 1024, 2, PFB_FMT_INT16_IQ, 8, 3, 16, 16, 4, 64, 68, 260, 1088, false, 2, true>;
// cfg4, second plan: 1024 threads x 1 column (16 waves per CU instead of 8: the window is half as many
// registers per thread), 1024 = 8 x 8 x 16, conflict-free padding
using Cfg1024x16i16b =
    FastCfg<1024, 16, 1024, 1, PFB_FMT_INT16_IQ, 8, 3, 8, 8, 16, 128, 128, 65, 1040, false, 1, true>;
// the reference's training-set band count: numBands = round(fs / 0.1e6) = 560 at fs = 56 MHz
// (generate_channelized_training_iq.m:95-96).  560 = 10 x 8 x 7; 560 of 576 threads own columns; chunks
// of 7 frames make every pass one iteration (7 * 80 = 560 final-pass items); at most 2-way LDS conflicts
// on about a third of the accesses (no padding removes them for this size, tools/fft_plan_model.py)
using Cfg560x12i16 =
    FastCfg<560, 12, 560, 1, PFB_FMT_INT16_IQ, 7, 3, 10, 8, 7, 56, 71, 82, 600, false, 3, true>;
using Cfg560x12i8 =
    FastCfg<560, 12, 560, 1, PFB_FMT_INT8_IQ, 7, 3, 10, 8, 7, 56, 71, 82, 600, false, 3, true>;
// cfg4, team plan (the default): 512 FIR threads x 2 columns filter chunks of 4 frames and run the last pass
// + stores of the chunk before the previous one; 4 FFT waves take one frame each for the first two passes
// (16 x 16 x 4, wave-local, twiddles in registers); 12 waves per workgroup, three LDS chunk buffers, one
// workgroup barrier per chunk (pfb_fast.hpp, schedule T)
using Cfg1024x16i16t =
    FastCfg<1024, 16, 1024, 2, PFB_FMT_INT16_IQ, 4, 3, 16, 16, 4, 64, 68, 260, 1088, false, 3, false>;
// M = 560, team plan (the default): 280 FIR threads x 2 columns (5 waves) + 4 FFT waves, chunks of 4 frames,
// 560 = 14 x 10 x 4 so that both FFT-team passes are one item per lane (40 and 56 of 64 lanes; with 10 x 8 x 7 the
// second pass needed two iterations and the FFT team was the bottleneck: 1.00 -> 0.83-0.87 ms)
using Cfg560x12i16t =
    FastCfg<560, 12, 560, 2, PFB_FMT_INT16_IQ, 4, 3, 14, 10, 4, 40, 60, 140, 600, false, 3, false>;
using Cfg560x12i8t =
    FastCfg<560, 12, 560, 2, PFB_FMT_INT8_IQ, 4, 3, 14, 10, 4, 40, 60, 140, 600, false, 3, false>;
// the same teams on chunks of TWO frames: 5 FIR + 2 FFT waves, a 13-row window and 29 KB of LDS per workgroup, so that
// TWO workgroups fit a CU (14 waves at <= 128 registers), neither synchronised with the other
using Cfg560x12i16t2 =
    FastCfg<560, 12, 560, 2, PFB_FMT_INT16_IQ, 2, 3, 14, 10, 4, 40, 60, 140, 600, false, 4, false>;
using Cfg560x12i8t2 =
    FastCfg<560, 12, 560, 2, PFB_FMT_INT8_IQ, 2, 3, 14, 10, 4, 40, 60, 140, 600, false, 4, false>;

// complex float32 input at the training script's band count (generate_channelized_training_iq.m:95-100 channelizes data
// that only exists as complex doubles in MATLAB's memory) and at cfg4's: the lockstep plans, whose one column per thread
// leaves room for 8-byte samples (the 4-frame team plans' two-column FIR window spills 40-100 registers with them; the
// 2-frame teams further down do not, and are the defaults since round 3: these two stay as variant 1)
using Cfg560x12f32 = FastCfg<560, 12, 560, 1, PFB_FMT_CF32, 7, 3, 10, 8, 7, 56, 71, 82, 600, false, 3, true>;
using Cfg1024x16f32b = FastCfg<1024, 16, 1024, 1, PFB_FMT_CF32, 8, 3, 8, 8, 16, 128, 128, 65, 1040, false, 1, true>;

// cfg4, schedule W (pfb_fast.hpp): independent workgroups whose waves filter their columns and then transform whole
// frames by themselves; the window stays packed (raw int16 pairs).  Three shapes of it:
//   twin    512 threads x 2 columns, chunks of 8 frames (one per wave), 128 registers, 78 KB of LDS: 2 workgroups per CU
//   triple  256 threads x 4 columns (16-byte loads), chunks of 4 frames, 168 registers, 44 KB: 3 workgroups per CU
//   duo     256 threads x 4 columns, chunks of 8 frames (two per wave), 256 registers -- taps resident --, 78 KB: 2 per CU
using Cfg1024x16i16w =
    FastCfg<1024, 16, 1024, 2, PFB_FMT_INT16_IQ, 8, 3, 16, 16, 4, 64, 68, 260, 1088, false, 4, true, true>;
using Cfg1024x16i16q =
    FastCfg<1024, 16, 1024, 4, PFB_FMT_INT16_IQ, 4, 3, 16, 16, 4, 64, 68, 260, 1088, false, 3, true, true>;
using Cfg1024x16i16d =
    FastCfg<1024, 16, 1024, 4, PFB_FMT_INT16_IQ, 8, 3, 16, 16, 4, 64, 68, 260, 1088, false, 2, true, true>;

// complex float32 input on the team plans: with chunks of TWO frames the FIR thread's state (13- / 17-row window, two row
// sets of 8-byte samples in flight) fits where the 4-frame teams spilled 40-100 registers
// (the same 2-frame teams on int16 cfg4 -- 10 waves, 128 registers, still one workgroup per CU -- : 0.52 against 0.60 for the
// 4-frame teams; not registered)
using Cfg560x12f32t2 =
    FastCfg<560, 12, 560, 2, PFB_FMT_CF32, 2, 3, 14, 10, 4, 40, 60, 140, 600, false, 4, false>;
using Cfg1024x16f32t2 =
    FastCfg<1024, 16, 1024, 2, PFB_FMT_CF32, 2, 3, 16, 16, 4, 64, 68, 260, 1088, false, 3, false>;
// (M = 560 on schedule 13 -- 5 waves per workgroup, chunks of 5 frames, one frame per wave, 124 registers, three workgroups
// per CU -- is bit-identical to the team plans and slower: 0.37-0.39 of the roofline against 0.524 in one process; removed)
static const FastEntry kRows[] = {
    entry<Cfg1024x16i16t>("pfb_fast<M1024,P16,D1024,int16>", 512, 6),
    entry<Cfg1024x16i16b>("pfb_fast<M1024,P16,D1024,int16,16w>", 256, 0),
    entry<Cfg1024x16i16>("pfb_fast<M1024,P16,D1024,int16,8w>", 256, 0),
    // M = 560: teams on 2-frame chunks, two workgroups per CU (0.515 of the roofline against 0.476 for one 9-wave workgroup
    // on 4-frame chunks, same process), then the 4-frame teams and the 9-wave lockstep plan
    entry<Cfg560x12i16t2>("pfb_fast<M560,P12,D560,int16>", 512, 6),
    entry<Cfg560x12i16t>("pfb_fast<M560,P12,D560,int16,4f>", 512, 6),
    entry<Cfg560x12i16>("pfb_fast<M560,P12,D560,int16,9w>", 252, 0),
    // 8-bit samples: the same order (the 4-frame team plan, spill-free since its loop issues the same memory operations on
    // every path, 0.217; the lockstep plan, the default while the team plan spilled 30 registers, 0.180)
    entry<Cfg560x12i8t2>("pfb_fast<M560,P12,D560,int8>", 512, 6),
    entry<Cfg560x12i8t>("pfb_fast<M560,P12,D560,int8,4f>", 512, 6),
    entry<Cfg560x12i8>("pfb_fast<M560,P12,D560,int8,9w>", 252, 0),
    // complex float32: the 2-frame teams (0.61 / 0.63 of the roofline) ahead of the lockstep plans (0.47 / 0.42)
    entry<Cfg560x12f32t2>("pfb_fast<M560,P12,D560,cf32>", 512, 6),
    entry<Cfg560x12f32>("pfb_fast<M560,P12,D560,cf32,9w>", 252, 0),
    entry<Cfg1024x16f32t2>("pfb_fast<M1024,P16,D1024,cf32>", 512, 6),
    entry<Cfg1024x16f32b>("pfb_fast<M1024,P16,D1024,cf32,16w>", 256, 0),
    entry<Cfg1024x16i16d>("pfb_fast<M1024,P16,D1024,int16,duo>", 256, 13),
};

FastTablePart fast_table_big() { return FastTablePart{kRows, (int)(sizeof(kRows) / sizeof(kRows[0]))}; }

}  // namespace pfb
