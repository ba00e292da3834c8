/*
 * pfb_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Float64 restatement of the channelizer path of cwozny/sdr_channelizer
 * (reference paths below are relative to /root/reference):
 *
 *   int -> complex normalise   matlab/channelizer_example.m:18-21,
 *                              matlab/create_pdws_channelized.m:35-38
 *   (conjugate) transpose      matlab/channelizer_example.m:23 vs
 *                              matlab/create_pdws_channelized.m:44-48
 *   truncate to multiple of M  matlab/create_pdws_channelized.m:52-54
 *   dsp.Channelizer(M) call    matlab/channelizer_example.m:31,56
 *                              matlab/create_pdws_channelized.m:33,57
 *   fftshift(.,2)              matlab/channelizer_example.m:58
 *                              matlab/create_pdws_channelized.m:60
 *   centerFrequencies          matlab/channelizer_example.m:60
 *                              matlab/create_pdws_channelized.m:42
 *   .iq record header          cpp/IqPacket.h:9-25,
 *                              matlab/convert_my_iq_to_mat.m:40-102
 *
 * PARITY UNPINNED for the filterbank arithmetic: the reference delegates it to
 * MathWorks' closed-source dsp.Channelizer (DSP System Toolbox, no version
 * pinned anywhere in the reference; not present under /root/reference; no
 * MATLAB/Octave in the build image; the reference ships no tests, fixtures or
 * golden vectors).  The arithmetic below restates the documented polyphase
 * analysis filter bank (channel k = prototype modulated to +k*fs/M, maximally
 * decimated or D = M/2), in two independent formulations that must agree to
 * ~1e-12, and with scipy.signal.upfirdn applied per channel (tests/test_oracle.py).
 * The record header / filename pieces ARE
 * pinned: oracle/_ref builds the reference's own IqPacket.h / Helper.cpp and
 * tests compare against it.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * use anything in this directory.
 *
 * Canonical definition:
 *
 *   x[n]   = (I[n] + j Q[n]) / 2^(bitWidth-1)          n >= 0 ; x[n<0] = 0
 *   y_k[m] = sum_{n=0}^{MP-1} h[n] e^{+j 2 pi k n / M} x[m D + off - n]
 *            k = 0..M-1,  m = 0..F-1,  F = floor(N / D),  0 <= off < D
 *
 * off defaults to D-1 (frame m consumes input samples [mD, mD+D), newest
 * sample meets h[0]).  Output is frame-major: y[m*M + k].
 */
#ifndef PFB_ORACLE_H
#define PFB_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  int M;              /* channels                                        */
  int P;              /* taps per channel (prototype has M*P taps)       */
  int D;              /* decimation: M (critical) or any divisor-ish D   */
  int off;            /* input_offset, 0 <= off < D ; -1 -> D-1          */
  int conj_input;     /* channelizer_example.m:23 quirk (iq')            */
  int derotate;       /* multiply y_k[m] by e^{-j 2 pi k m D / M}        */
  int fftshift;       /* reorder channels like fftshift(out,2)           */
} pfbo_cfg;

/* (I + jQ) / 2^(bit_width-1); interleaved int payload as the recorders hold it
 * (cpp/blade_record_iq_12bit.cpp:268,322). */
void pfbo_unpack_int16(const int16_t* iq, size_t n, int bit_width, double* xr, double* xi);
void pfbo_unpack_int8(const int8_t* iq, size_t n, int bit_width, double* xr, double* xi);

/* Formulation A: per-channel complex band-pass FIR (h[n] e^{j2pi kn/M}) then
 * decimate by D.  O(F * M * MP).  yr/yi hold F*M doubles, frame-major. */
size_t pfbo_channelize_direct(const double* xr, const double* xi, size_t n,
                              const double* h, const pfbo_cfg* cfg,
                              double* yr, double* yi);

/* Formulation B: M polyphase branches then an M-point inverse-type DFT per
 * frame (plain O(M^2) DFT, any M). */
size_t pfbo_channelize_polyphase(const double* xr, const double* xi, size_t n,
                                 const double* h, const pfbo_cfg* cfg,
                                 double* yr, double* yi);

/* Formulation B': polyphase branches + iterative radix-2 FFT (M = 2^k only).
 * Used where B would be too slow (M = 1024). Returns 0 frames if M is not 2^k. */
size_t pfbo_channelize_polyphase_fft(const double* xr, const double* xi, size_t n,
                                     const double* h, const pfbo_cfg* cfg,
                                     double* yr, double* yi);

/* centerFrequencies(channelizer, fs): unshifted order
 * [0, 1, ..., ceil(M/2)-1, -floor(M/2), ..., -1] * fs/M. */
void pfbo_center_frequencies(int M, double fs, double* out);

/* Candidate default prototype ("unverifiable"):
 * h[n] = sinc((n - MP/2)/M)/M * kaiser(MP+1, 0.1102*(A-8.7))[n], n=0..MP-1. */
void pfbo_design_prototype(int M, int P, double atten_db, double* h);

/* ---- fp32 multi-threaded CPU port (bench.py cpu_baseline only) ----------- */
/* int16 interleaved input, fp32 arithmetic, OpenMP over frame blocks.
 * out: F*M interleaved (re,im) floats, frame-major. Returns frames. */
size_t pfbo_channelize_f32_i16(const int16_t* iq, size_t n, int bit_width,
                               const float* h, int M, int P, int D, int off,
                               float* out, int num_threads);
int pfbo_max_threads(void);

/* ---- .iq record header (cpp/IqPacket.h:9-25) ------------------------------ */
typedef struct {
  uint32_t marker;        /* "endianness" word                               */
  int      file_format;   /* 1, 2 or 3 (convert_my_iq_to_mat.m:42-57)       */
  uint32_t link_speed;
  uint64_t frequency_hz;
  uint32_t bandwidth_hz;
  uint32_t sample_rate_sps;
  double   rx_gain_db;    /* f32 for fmt>=3, u32 otherwise (:73-77)         */
  uint32_t num_samples;
  uint32_t bit_width;
  uint32_t spare0;
  char     board_name[17];
  char     serial_number[17];
  char     fpga_version[17];
  char     fw_version[17];
  double   sample_start_time;
  uint32_t header_bytes;  /* 104 (fmt 1) or 112 (fmt >= 2)                   */
  uint32_t bytes_per_sample; /* 2 (int8 IQ) or 4 (int16 IQ)                  */
} pfbo_iq_header;

/* Returns 0 on success, <0 on error (unknown marker, short buffer, bad width). */
int pfbo_parse_iq_header(const uint8_t* bytes, size_t len, pfbo_iq_header* out);

/* ---- channelized PDW extraction (matlab/create_pdws_channelized.m:64-143) -- */
typedef struct {
  double toa;   /* UTC seconds (:98)   */
  double freq;  /* Hz (:122)           */
  double pw;    /* seconds (:110)      */
  double snr;   /* dB (:105)           */
  int    sat;   /* (:130-132)          */
  int    bin;   /* 0-based shifted column the pulse was found in (extra) */
  double mag;   /* median magnitude over the pulse: thisAmp (:101, not stored by the channelized
                   script) / thisMag (create_pdws.m:70,96) */
} pfbo_pdw;

/* y: F x M complex, frame-major, already fftshift-ed (:60). fs_in is the rate
 * BEFORE decimation; the reference divides by M (fs <- fs/M, :62); decim carries the true
 * decimation so the 2x oversampled bank (decim = M/2) gets its real frame rate.  matlab_quirks is a bit mask:
 *   bit 0  phase(toa:jj) linear-indexes column 1 of the phase matrix (:114) instead of the pulse's column;
 *   bit 1  binFreqs(bin) (:80) comes from the FFT-ordered (unshifted) centre-frequency list indexed with the
 *          shifted column -- what the script computes IF MathWorks' centerFrequencies(channelizer,fs) (:42)
 *          returns the unshifted list; unpinned (closed toolbox; channelizer_example.m:58-66 plots
 *          fftshift(out,2) against that list, which suggests it is already centred).  Without the bit the
 *          column's true centre frequency is used.
 * Returns the number of PDWs found; writes at most max_out of them. */
size_t pfbo_extract_pdws(const double* yr, const double* yi, size_t F, int M, int decim,
                         double fs_in, double fc, double sample_start_time,
                         double snr_threshold_db, int matlab_quirks,
                         pfbo_pdw* out, size_t max_out);

/* ---- raw (un-channelized) PDW extraction (matlab/create_pdws.m:30-105) ---------------- */
/* x: n complex samples already normalised to [-1, 1) (:30-33).  One noise floor = median |x| (:44);
 * leading edge at |x| >= NF*10^(snr_db/10) (:45-46,57), trailing edge at |x| <= NF*10^(trail_db/10)
 * (:47,63); the reference uses 18 and 3.  fs is the sample rate (no decimation).  bin is 0.
 * If noise_floor is not NULL it receives NF.  Returns the number of PDWs found. */
size_t pfbo_extract_pdws_raw(const double* xr, const double* xi, size_t n, double fs, double fc,
                             double sample_start_time, double snr_threshold_db, double trailing_threshold_db,
                             pfbo_pdw* out, size_t max_out, double* noise_floor);

#ifdef __cplusplus
}
#endif
#endif
