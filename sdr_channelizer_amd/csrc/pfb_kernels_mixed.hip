// pfb_kernels_mixed.hip -- fused-kernel table, part 4: band counts M = 2^a 3^b 5^c 7^d that are plausible radio rates
// (see pfb_table.h).
//
// The reference's band count is whatever the recorder's sample rate is in MHz (numBands = fs * 1e-6,
// /root/reference/matlab/channelizer_example.m:29, create_pdws_channelized.m:31) or in units of 0.1 MHz
// (generate_channelized_training_iq.m:95-96): 12, 24, 25, 30, 48, 50, 80, 96, 100, 112, 120, 160, 200, 250, 280, 320,
// 400, 500, 512 besides the shapes of the other tables.  All of them are instantiations of the same templates as the
// tuned shapes (pfb_fast.hpp), with radices from {2 .. 16 \ 9, 11, 13, 15} and the LDS paddings
// tools/fft_plan_search.py picks (modelled bank conflicts minimal, last radix small so that the final pass stores long
// runs of adjacent channels):
//   M < 64           SegKernel: 64 / M segments of the run per wave, two passes
//   64 < M <= 160    one wave (two in lockstep at M = 160), 2 adjacent columns per lane, two passes
//   M >= 200         4-5 waves in lockstep, 1-2 columns per lane, three passes in place, twiddles from the table
// 12 taps per band (dsp.Channelizer's default, the only value the reference uses); shorter prototypes are zero-padded
// onto these kernels at create, everything else -- and every other M -- takes the generic kernel.
#include <hip/hip_runtime.h>

#include "pfb_table.h"

namespace pfb {

//                 M    P   D  CPT FMT               C NP R0 R1 R2 RS0 RS1 RS2 FS  PP     MINW  TW_TABLE
using Cfg12x12i16 = FastCfg<12, 12, 12, 1, PFB_FMT_INT16_IQ, 8, 2, 6, 2, 1, 2, 7, 0, 38, true, 2>;  // (two waves per SIMD: its 168-register state does not fit three)
using Cfg24x12i16 = FastCfg<24, 12, 24, 1, PFB_FMT_INT16_IQ, 8, 2, 12, 2, 1, 2, 13, 0, 26, true, 4>;
// (M = 25: 5 x 5 leaves the final pass 5-channel = 40-byte store runs; 0.34-0.39 of roofline -- another frame stride,
// chunks of 16 frames per segment: no better)
using Cfg25x12i16 = FastCfg<25, 12, 25, 1, PFB_FMT_INT16_IQ, 8, 2, 5, 5, 1, 5, 5, 0, 25, true, 3>;
using Cfg30x12i16 = FastCfg<30, 12, 30, 1, PFB_FMT_INT16_IQ, 8, 2, 10, 3, 1, 3, 13, 0, 39, true, 4>;
// M = 48 / 50 as ordinary single-wave plans (48 / 50 of 64 lanes own a column, like M = 56): 0.52 / 0.53 against
// 0.37 / 0.46 as one-segment SegKernel shapes
using Cfg48x12i16 = FastCfg<48, 12, 48, 1, PFB_FMT_INT16_IQ, 8, 2, 16, 3, 1, 3, 19, 0, 57, false, 4>;
using Cfg50x12i16 = FastCfg<50, 12, 50, 1, PFB_FMT_INT16_IQ, 8, 2, 10, 5, 1, 5, 11, 0, 55, false, 4>;

using Cfg80x12i16 = FastCfg<80, 12, 80, 2, PFB_FMT_INT16_IQ, 8, 2, 16, 5, 1, 5, 17, 0, 101, false, 2>;
using Cfg96x12i16 = FastCfg<96, 12, 96, 2, PFB_FMT_INT16_IQ, 8, 2, 16, 6, 1, 6, 17, 0, 102, false, 2>;
using Cfg100x12i16 = FastCfg<100, 12, 100, 2, PFB_FMT_INT16_IQ, 4, 2, 10, 10, 1, 10, 10, 0, 106, false, 2>;
using Cfg112x12i16 = FastCfg<112, 12, 112, 2, PFB_FMT_INT16_IQ, 8, 2, 16, 7, 1, 7, 17, 0, 135, false, 2>;
using Cfg120x12i16 = FastCfg<120, 12, 120, 2, PFB_FMT_INT16_IQ, 4, 2, 12, 10, 1, 10, 13, 0, 138, false, 2>;
// (M = 160 with 4 columns per lane on one wave needs 256 registers and still spills: 2 columns, two waves in lockstep)
using Cfg160x12i16 = FastCfg<160, 12, 160, 2, PFB_FMT_INT16_IQ, 8, 2, 16, 10, 1, 10, 17, 0, 170, false, 2, true>;

using Cfg200x12i16 = FastCfg<200, 12, 200, 1, PFB_FMT_INT16_IQ, 8, 3, 10, 10, 2, 20, 20, 104, 210, false, 4, true>;
using Cfg250x12i16 = FastCfg<250, 12, 250, 1, PFB_FMT_INT16_IQ, 4, 3, 10, 5, 5, 25, 51, 50, 274, false, 2, true>;
using Cfg280x12i16 = FastCfg<280, 12, 280, 1, PFB_FMT_INT16_IQ, 8, 3, 7, 10, 4, 40, 28, 70, 296, false, 2, true>;
// (M = 320 as 16 x 10 x 2: 0.37 -- the FFT team's two passes then use 20 and 32 of a wave's 64 lanes; 8 x 10 x 4: 40 and 32)
using Cfg320x12i16 = FastCfg<320, 12, 320, 1, PFB_FMT_INT16_IQ, 8, 3, 8, 10, 4, 40, 34, 84, 340, false, 2, true>;
// (M = 400 with 2 columns per lane and chunks of 4: 0.39 in lockstep, 0.34 as teams; 1 column, chunks of 8, teams: 0.46)
using Cfg400x12i16 = FastCfg<400, 12, 400, 1, PFB_FMT_INT16_IQ, 8, 3, 10, 10, 4, 40, 40, 100, 424, false, 2, true>;
using Cfg500x12i16 = FastCfg<500, 12, 500, 2, PFB_FMT_INT16_IQ, 4, 3, 10, 10, 5, 50, 51, 102, 516, false, 2, true>;
using Cfg512x12i16 = FastCfg<512, 12, 512, 2, PFB_FMT_INT16_IQ, 8, 3, 16, 16, 2, 32, 33, 264, 528, false, 2, true>;

// M = 250 / 500 as teams on 2-frame chunks, two columns per FIR thread: 4 / 6 waves per workgroup, so that four / two
// unsynchronised workgroups fit a CU (what took M = 560 from 0.476 to 0.515): 0.398 -> 0.502 and 0.409 -> 0.516, the
// defaults.  The same plan lost on 200 (0.380 vs 0.386), 280 (0.370 vs 0.422), 320 (0.407 vs 0.460), 400 (0.426 vs
// 0.497) and 512 (0.438 vs 0.497) against their 8-frame teams / lockstep plans: profiles/r03_mixed_radix_two_frame_teams.txt
using Cfg250x12i16t2 = FastCfg<250, 12, 250, 2, PFB_FMT_INT16_IQ, 2, 3, 10, 5, 5, 25, 51, 50, 274, false, 4, false>;
using Cfg500x12i16t2 = FastCfg<500, 12, 500, 2, PFB_FMT_INT16_IQ, 2, 3, 10, 10, 5, 50, 51, 102, 516, false, 4, false>;

static const FastEntry kRows[] = {
    // default run lengths and schedules: the best of tools/mixed_probe.py's sweep (profiles/r03_mixed_radix_shapes.txt);
    // the three-pass shapes also have the FIR-team / FFT-team instantiation (schedule 6), the default where it won
    seg_entry<Cfg12x12i16>("pfb_fast<M12,P12,D12,int16>", 1000),
    seg_entry<Cfg24x12i16>("pfb_fast<M24,P12,D24,int16>", 256),
    seg_entry<Cfg25x12i16>("pfb_fast<M25,P12,D25,int16>", 64),
    seg_entry<Cfg30x12i16>("pfb_fast<M30,P12,D30,int16>", 256),
    entry<Cfg48x12i16>("pfb_fast<M48,P12,D48,int16>", 24, 11),  // one PERIOD per run: the ring variant (0.561 against 0.535 at 128)
    entry<Cfg50x12i16>("pfb_fast<M50,P12,D50,int16>", 256, 7),
    entry<Cfg80x12i16>("pfb_fast<M80,P12,D80,int16>", 256, 0),
    entry<Cfg96x12i16>("pfb_fast<M96,P12,D96,int16>", 256, 0),
    entry<Cfg100x12i16>("pfb_fast<M100,P12,D100,int16>", 256, 7),
    entry<Cfg112x12i16>("pfb_fast<M112,P12,D112,int16>", 128, 0),
    entry<Cfg120x12i16>("pfb_fast<M120,P12,D120,int16>", 512, 7),
    entry<Cfg160x12i16>("pfb_fast<M160,P12,D160,int16>", 128, 0),
    entry<Cfg200x12i16>("pfb_fast<M200,P12,D200,int16>", 256, 0),
    entry<Cfg250x12i16t2>("pfb_fast<M250,P12,D250,int16>", 512, 6),
    entry<Cfg250x12i16>("pfb_fast<M250,P12,D250,int16,lockstep>", 256, 0),
    entry<Cfg280x12i16>("pfb_fast<M280,P12,D280,int16>", 256, 6),
    entry<Cfg320x12i16>("pfb_fast<M320,P12,D320,int16>", 256, 6),
    entry<Cfg400x12i16>("pfb_fast<M400,P12,D400,int16>", 256, 6),
    entry<Cfg500x12i16t2>("pfb_fast<M500,P12,D500,int16>", 512, 6),
    entry<Cfg500x12i16>("pfb_fast<M500,P12,D500,int16,lockstep>", 128, 0),
    entry<Cfg512x12i16>("pfb_fast<M512,P12,D512,int16>", 512, 0),
};

FastTablePart fast_table_mixed() { return FastTablePart{kRows, (int)(sizeof(kRows) / sizeof(kRows[0]))}; }

}  // namespace pfb
