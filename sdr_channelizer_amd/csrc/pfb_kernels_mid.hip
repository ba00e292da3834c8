// pfb_kernels_mid.hip -- fused-kernel table, part 2: the other single-wave shapes (cfg3, cfg5, M = 56, the small banks) (see pfb_table.h)
#include <hip/hip_runtime.h>

#include "pfb_table.h"

namespace pfb {

//                 M    P   D  CPT FMT               C NP R0 R1 R2 RS0 RS1 RS2 FS  PP     MINW  TW_TABLE
// cfg5: 2x oversampled, 24 taps per column, 128 = 16 x 8 (final-pass LDS reads are 2-way conflicted:
// no single frame stride serves both passes, tools/fft_plan_model.py); schedule 7 (6 FIR/FFT wave pairs per
// workgroup) measured within noise of one wave doing both (+0..3 %), so the default stays schedule 0
using Cfg128x12os2i16 = FastCfg<128, 12, 64, 1, PFB_FMT_INT16_IQ, 8, 2, 16, 8, 1, 8, 17, 0, 136, false, 2>;
// cfg3: 4 adjacent columns per lane (8-byte loads of int8 I/Q), 256 = 16 x 16, rows of both passes padded to 17
// (RS0 = 16 made every FIR write a 4-way LDS conflict: lanes 4 elements apart, sixteen lanes per LDS cycle over
// float2 addresses mod 16; SQ_LDS_BANK_CONFLICT 43 % of the LDS cycles -> 0); short sliding runs (32
// frames, 22 % more row reads of a stream that is 80 % writes) keep the chip's active window small: +4 % over 256
using Cfg256x8i8  = FastCfg<256, 8, 256, 4, PFB_FMT_INT8_IQ,  4, 2, 16, 16, 1, 17, 17, 0, 272, false, 2>;
using Cfg256x8i16 = FastCfg<256, 8, 256, 4, PFB_FMT_INT16_IQ, 4, 2, 16, 16, 1, 17, 17, 0, 272, false, 2>;
// the reference's own band count: numBands = fs*1e-6 = 56 (channelizer_example.m:29, generate_pulsed_iq.m:12),
// 56 = 8 x 7; 56 of the wave's 64 lanes own columns (2-way LDS conflicts on about half the accesses); default
// schedule 7 (8 FIR/FFT wave pairs per workgroup over sliding runs of 512 frames, +17 %)
using Cfg56x12i16 = FastCfg<56, 12, 56, 1, PFB_FMT_INT16_IQ, 8, 2, 8, 7, 1, 7, 9, 0, 71, false, 4>;
using Cfg56x12i8  = FastCfg<56, 12, 56, 1, PFB_FMT_INT8_IQ,  8, 2, 8, 7, 1, 7, 9, 0, 71, false, 4>;
// small banks (numBands = fs * 1e-6 at 8 / 16 / 32 Msps, channelizer_example.m:29).  M = 32: the ordinary kernel with
// half the lanes idle through the FIR (52 %; two segments per wave measured 47 %).  M = 16 and 8: SegKernel, 64 / M
// segments of the run per wave so that every lane filters (36 -> 49 %, 24 -> 56 %); ping-pong LDS buffers (PP = true).
// M = 8 is 4 x 2, not 2 x 4: the last pass then leaves a lane group 4 adjacent channels (32-byte runs), worth 40 -> 56 %
using Cfg32x12i16 = FastCfg<32, 12, 32, 1, PFB_FMT_INT16_IQ, 8, 2, 8, 4, 1, 4, 9, 0, 36, false, 4>;
using Cfg16x12i16 = FastCfg<16, 12, 16, 1, PFB_FMT_INT16_IQ, 8, 2, 4, 4, 1, 4, 5, 0, 20, true, 4>;
using Cfg8x12i16 = FastCfg<8, 12, 8, 1, PFB_FMT_INT16_IQ, 8, 2, 4, 2, 1, 2, 5, 0, 10, true, 3>;
// (MIN_WAVES 3 for M = 8, 10 and the int8 M = 16: at 4 waves per SIMD these spilled 12-30 registers; without the spills
// M = 8 56 -> 58 %, int8 35 -> 43 %, cf32 57 -> 66 %, M = 16 int8 32 -> 36 %, M = 10 38 -> 46 %)
using Cfg32x12i8 = FastCfg<32, 12, 32, 1, PFB_FMT_INT8_IQ, 8, 2, 8, 4, 1, 4, 9, 0, 36, false, 4>;
using Cfg16x12i8 = FastCfg<16, 12, 16, 1, PFB_FMT_INT8_IQ, 8, 2, 4, 4, 1, 4, 5, 0, 20, true, 3>;
using Cfg8x12i8 = FastCfg<8, 12, 8, 1, PFB_FMT_INT8_IQ, 8, 2, 4, 2, 1, 2, 5, 0, 10, true, 3>;
using Cfg8x12f32 = FastCfg<8, 12, 8, 1, PFB_FMT_CF32, 8, 2, 4, 2, 1, 2, 5, 0, 10, true, 3>;
// numBands at the bladeRF's round rates 10 / 20 / 40 Msps: 5 x 2, 10 x 2, 8 x 5 (5- and 10-point DFTs); 6, 3 and 1
// segments per wave (60 / 60 / 40 of 64 lanes work).  M = 20 as 10 x 2, not 4 x 5: 80-byte instead of 32-byte store
// runs, 36 -> 51 %; M = 10 as 5 x 2, not 2 x 5 (odd first radix: SegKernel reads those twiddle rows element by
// element): 30 -> 39 %; M = 40 as 10 x 4 measured half the speed of 8 x 5
using Cfg10x12i16 = FastCfg<10, 12, 10, 1, PFB_FMT_INT16_IQ, 8, 2, 5, 2, 1, 2, 5, 0, 10, true, 3>;
using Cfg20x12i16 = FastCfg<20, 12, 20, 1, PFB_FMT_INT16_IQ, 8, 2, 10, 2, 1, 2, 15, 0, 42, true, 3>;
using Cfg40x12i16 = FastCfg<40, 12, 40, 1, PFB_FMT_INT16_IQ, 8, 2, 8, 5, 1, 5, 9, 0, 45, true, 4>;

// complex float32 input (what a MATLAB caller holds after normalising, channelizer_example.m:44-56) for the reference's
// own band count, cfg5's and cfg3's shapes: the same plans with 8-byte samples (no spills; M = 64 and M = 8 have theirs
// next to the integer ones, M = 560 and 1024 in pfb_kernels_big.hip).  Everything else in cf32 takes the generic kernel.
using Cfg56x12f32 = FastCfg<56, 12, 56, 1, PFB_FMT_CF32, 8, 2, 8, 7, 1, 7, 9, 0, 71, false, 4>;
using Cfg128x12os2f32 = FastCfg<128, 12, 64, 1, PFB_FMT_CF32, 8, 2, 16, 8, 1, 8, 17, 0, 136, false, 2>;
using Cfg256x8f32 = FastCfg<256, 8, 256, 4, PFB_FMT_CF32, 4, 2, 16, 16, 1, 17, 17, 0, 272, false, 2>;

// M = 32 with 8-bit samples: in the ordinary kernel half the lanes of every 2-byte row load are idle, and sub-dword
// loads cost the memory pipeline as much as full ones -- 25 % of roofline.  Two segments per wave (SegKernel) fill
// the loads: 47 % (with 4-byte samples the ordinary kernel stays ahead, 54 vs 51 %)
using Cfg32x12i8seg = FastCfg<32, 12, 32, 1, PFB_FMT_INT8_IQ, 8, 2, 8, 4, 1, 4, 9, 0, 36, true, 3>;

static const FastEntry kRows[] = {
    // schedule 11 (sliding runs, the next chunk's FIR scheduled into this chunk's FFT, rows two chunks ahead) over short
    // runs: cfg5 1.125 -> 1.048 ms per 2^28 samples (59.6 -> 64.0 % of roofline), cfg3 2.104 -> 2.073 ms (63.8 -> 64.8 %)
    entry<Cfg128x12os2i16>("pfb_fast<M128,P12,D64,int16>", 32, 11),
    // cfg3: schedule 0 again -- once its loop kept the compiler's wait counts exact (FastKernel::run_impl) it read 0.68
    // against 0.64 for schedule 11, whose 4-column register budget spills inside the chunk loop
    entry<Cfg256x8i8>("pfb_fast<M256,P8,D256,int8>", 32, 0),
    entry<Cfg256x8i16>("pfb_fast<M256,P8,D256,int16>", 32, 0),
    entry<Cfg32x12i16>("pfb_fast<M32,P12,D32,int16>", 512, 0),
    seg_entry<Cfg16x12i16>("pfb_fast<M16,P12,D16,int16>", 1024),
    seg_entry<Cfg8x12i16>("pfb_fast<M8,P12,D8,int16>", 1024),
    seg_entry<Cfg32x12i8seg>("pfb_fast<M32,P12,D32,int8>", 1024),
    entry<Cfg32x12i8>("pfb_fast<M32,P12,D32,int8,1seg>", 512, 0),
    seg_entry<Cfg16x12i8>("pfb_fast<M16,P12,D16,int8>", 1024),
    seg_entry<Cfg8x12i8>("pfb_fast<M8,P12,D8,int8>", 1024),
    seg_entry<Cfg8x12f32>("pfb_fast<M8,P12,D8,cf32>", 1024),
    seg_entry<Cfg10x12i16>("pfb_fast<M10,P12,D10,int16>", 1008),
    seg_entry<Cfg20x12i16>("pfb_fast<M20,P12,D20,int16>", 1008),
    seg_entry<Cfg40x12i16>("pfb_fast<M40,P12,D40,int16>", 1024),
    entry<Cfg56x12i16>("pfb_fast<M56,P12,D56,int16>", 512, 7),
    entry<Cfg56x12i8>("pfb_fast<M56,P12,D56,int8>", 512, 7),
    entry<Cfg56x12f32>("pfb_fast<M56,P12,D56,cf32>", 512, 7),
    entry<Cfg128x12os2f32>("pfb_fast<M128,P12,D64,cf32>", 64, 0),
    entry<Cfg256x8f32>("pfb_fast<M256,P8,D256,cf32>", 32, 0),
};

FastTablePart fast_table_mid() { return FastTablePart{kRows, (int)(sizeof(kRows) / sizeof(kRows[0]))}; }

}  // namespace pfb
