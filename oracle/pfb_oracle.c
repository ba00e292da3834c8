/*
 * pfb_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).
 * See pfb_oracle.h for scope, reference citations and the "parity unpinned"
 * statement.  Plain C, float64, written for clarity rather than speed (the one
 * exception is pfbo_channelize_f32_i16, the fp32/OpenMP CPU port that
 * bench.py times as cpu_baseline).
 */
#include "pfb_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------------- */
/* a3: int -> complex normalise (channelizer_example.m:18-21)                */

void pfbo_unpack_int16(const int16_t* iq, size_t n, int bit_width, double* xr, double* xi) {
  const double max_val = ldexp(1.0, bit_width - 1); /* maxVal = 2^(bitWidth-1) */
  for (size_t i = 0; i < n; ++i) {
    xr[i] = (double)iq[2 * i] / max_val;     /* row 1 = I (convert_my_iq_to_mat.m:92-98) */
    xi[i] = (double)iq[2 * i + 1] / max_val; /* row 2 = Q */
  }
}

void pfbo_unpack_int8(const int8_t* iq, size_t n, int bit_width, double* xr, double* xi) {
  const double max_val = ldexp(1.0, bit_width - 1);
  for (size_t i = 0; i < n; ++i) {
    xr[i] = (double)iq[2 * i] / max_val;
    xi[i] = (double)iq[2 * i + 1] / max_val;
  }
}

/* ------------------------------------------------------------------------- */

static int cfg_off(const pfbo_cfg* c) { return c->off < 0 ? c->D - 1 : c->off; }

/* fftshift(out,2) destination column of channel k: the shifted row is
 * [k = ceil(M/2) .. M-1, 0 .. ceil(M/2)-1], i.e. dst = (k + floor(M/2)) mod M. */
static int shifted_col(int k, int M, int do_shift) {
  return do_shift ? (k + M / 2) % M : k;
}

static void post_store(const pfbo_cfg* c, size_t m, int k, double re, double im,
                       double* yr, double* yi) {
  const int M = c->M;
  if (c->derotate) {
    /* e^{-j 2 pi k m D / M}; reduce the integer phase first to stay exact */
    const uint64_t ph = ((uint64_t)k * ((m * (uint64_t)c->D) % (uint64_t)M)) % (uint64_t)M;
    const double a = -2.0 * M_PI * (double)ph / (double)M;
    const double cr = cos(a), ci = sin(a);
    const double tr = re * cr - im * ci, ti = re * ci + im * cr;
    re = tr;
    im = ti;
  }
  const int col = shifted_col(k, M, c->fftshift);
  yr[m * (size_t)M + col] = re;
  yi[m * (size_t)M + col] = im;
}

/* x[s] with zero history and the optional conjugation quirk */
static inline void sample(const double* xr, const double* xi, long long s, int conj,
                          double* re, double* im) {
  if (s < 0) {
    *re = 0.0;
    *im = 0.0;
  } else {
    *re = xr[s];
    *im = conj ? -xi[s] : xi[s];
  }
}

/* ------------------------------------------------------------------------- */
/* Formulation A: filter with h_k[n] = h[n] e^{+j2pi k n/M}, keep every D-th.  */

size_t pfbo_channelize_direct(const double* xr, const double* xi, size_t n,
                              const double* h, const pfbo_cfg* cfg,
                              double* yr, double* yi) {
  const int M = cfg->M, P = cfg->P, D = cfg->D, off = cfg_off(cfg);
  const size_t F = n / (size_t)D; /* create_pdws_channelized.m:52-54 truncation */
  const int L = M * P;
  for (size_t m = 0; m < F; ++m) {
    for (int k = 0; k < M; ++k) {
      double ar = 0.0, ai = 0.0;
      for (int t = 0; t < L; ++t) {
        double sr, si;
        sample(xr, xi, (long long)(m * (size_t)D) + off - t, cfg->conj_input, &sr, &si);
        /* modulated tap: angle reduced mod M before the trig call */
        const double a = 2.0 * M_PI * (double)(((long long)k * t) % M) / (double)M;
        const double hr = h[t] * cos(a), hi = h[t] * sin(a);
        ar += hr * sr - hi * si;
        ai += hr * si + hi * sr;
      }
      post_store(cfg, m, k, ar, ai, yr, yi);
    }
  }
  return F;
}

/* ------------------------------------------------------------------------- */
/* Polyphase branches u_p[m] = sum_q h[p + Mq] x[mD + off - p - Mq]           */

static void branches(const double* xr, const double* xi, size_t m, const double* h,
                     const pfbo_cfg* cfg, double* ur, double* ui) {
  const int M = cfg->M, P = cfg->P, D = cfg->D, off = cfg_off(cfg);
  for (int p = 0; p < M; ++p) {
    double ar = 0.0, ai = 0.0;
    for (int q = 0; q < P; ++q) {
      double sr, si;
      sample(xr, xi, (long long)(m * (size_t)D) + off - p - (long long)M * q,
             cfg->conj_input, &sr, &si);
      ar += h[p + M * q] * sr;
      ai += h[p + M * q] * si;
    }
    ur[p] = ar;
    ui[p] = ai;
  }
}

/* Formulation B: y_k = sum_p e^{+j2pi kp/M} u_p by plain DFT */
size_t pfbo_channelize_polyphase(const double* xr, const double* xi, size_t n,
                                 const double* h, const pfbo_cfg* cfg,
                                 double* yr, double* yi) {
  const int M = cfg->M, D = cfg->D;
  const size_t F = n / (size_t)D;
  double* ur = (double*)malloc(sizeof(double) * 4 * (size_t)M);
  if (!ur) return 0;
  double* ui = ur + M;
  double* wr = ui + M;
  double* wi = wr + M;
  for (int j = 0; j < M; ++j) {
    wr[j] = cos(2.0 * M_PI * j / M);
    wi[j] = sin(2.0 * M_PI * j / M);
  }
  for (size_t m = 0; m < F; ++m) {
    branches(xr, xi, m, h, cfg, ur, ui);
    for (int k = 0; k < M; ++k) {
      double ar = 0.0, ai = 0.0;
      for (int p = 0; p < M; ++p) {
        const int j = (int)(((long long)k * p) % M);
        ar += ur[p] * wr[j] - ui[p] * wi[j];
        ai += ur[p] * wi[j] + ui[p] * wr[j];
      }
      post_store(cfg, m, k, ar, ai, yr, yi);
    }
  }
  free(ur);
  return F;
}

/* in-place iterative radix-2, e^{+j...} kernel, natural in / natural out */
static void fft_pos_f64(double* re, double* im, int M, const double* wr, const double* wi) {
  for (int i = 1, j = 0; i < M; ++i) { /* bit reversal */
    int bit = M >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) {
      double t = re[i]; re[i] = re[j]; re[j] = t;
      t = im[i]; im[i] = im[j]; im[j] = t;
    }
  }
  for (int len = 2; len <= M; len <<= 1) {
    const int half = len >> 1, step = M / len;
    for (int i = 0; i < M; i += len) {
      for (int j = 0; j < half; ++j) {
        const double cr = wr[j * step], ci = wi[j * step];
        const double br = re[i + j + half] * cr - im[i + j + half] * ci;
        const double bi = re[i + j + half] * ci + im[i + j + half] * cr;
        re[i + j + half] = re[i + j] - br;
        im[i + j + half] = im[i + j] - bi;
        re[i + j] += br;
        im[i + j] += bi;
      }
    }
  }
}

size_t pfbo_channelize_polyphase_fft(const double* xr, const double* xi, size_t n,
                                     const double* h, const pfbo_cfg* cfg,
                                     double* yr, double* yi) {
  const int M = cfg->M, D = cfg->D;
  if (M < 2 || (M & (M - 1))) return 0;
  const size_t F = n / (size_t)D;
  double* ur = (double*)malloc(sizeof(double) * 4 * (size_t)M);
  if (!ur) return 0;
  double* ui = ur + M;
  double* wr = ui + M;
  double* wi = wr + M;
  for (int j = 0; j < M; ++j) {
    wr[j] = cos(2.0 * M_PI * j / M);
    wi[j] = sin(2.0 * M_PI * j / M);
  }
  for (size_t m = 0; m < F; ++m) {
    branches(xr, xi, m, h, cfg, ur, ui);
    fft_pos_f64(ur, ui, M, wr, wi);
    for (int k = 0; k < M; ++k) post_store(cfg, m, k, ur[k], ui[k], yr, yi);
  }
  free(ur);
  return F;
}

/* ------------------------------------------------------------------------- */
/* a10: centerFrequencies(channelizer, fs), unshifted order                   */

void pfbo_center_frequencies(int M, double fs, double* out) {
  for (int k = 0; k < M; ++k) {
    /* bins at or above the Nyquist bin wrap negative: [0..ceil(M/2)-1, -floor(M/2)..-1] */
    const int kk = (k < (M + 1) / 2) ? k : k - M;
    out[k] = (double)kk * fs / (double)M;
  }
}

/* ------------------------------------------------------------------------- */
/* Candidate prototype (convenience generator, unverifiable vs MathWorks)     */

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

void pfbo_design_prototype(int M, int P, double atten_db, double* h) {
  const int L = M * P; /* window has L+1 points, last one dropped */
  double beta;
  if (atten_db > 50.0) beta = 0.1102 * (atten_db - 8.7);
  else if (atten_db >= 21.0) beta = 0.5842 * pow(atten_db - 21.0, 0.4) + 0.07886 * (atten_db - 21.0);
  else beta = 0.0;
  const double i0b = bessel_i0(beta);
  for (int n = 0; n < L; ++n) {
    const double t = ((double)n - (double)L / 2.0) / (double)M;
    const double s = (t == 0.0) ? 1.0 : sin(M_PI * t) / (M_PI * t);
    const double r = 2.0 * (double)n / (double)L - 1.0; /* kaiser(L+1) abscissa */
    const double w = bessel_i0(beta * sqrt(fmax(0.0, 1.0 - r * r))) / i0b;
    h[n] = s / (double)M * w;
  }
}

/* ------------------------------------------------------------------------- */
/* fp32 OpenMP CPU port: the "MATLAB/CPU reference" stand-in that bench.py     */
/* times on the GPU box's host cores (kind = "port").                          */

static void fft_pos_f32(float* re, float* im, int M, const float* wr, const float* wi,
                        const int* rev) {
  for (int i = 0; i < M; ++i) {
    const int j = rev[i];
    if (i < j) {
      float t = re[i]; re[i] = re[j]; re[j] = t;
      t = im[i]; im[i] = im[j]; im[j] = t;
    }
  }
  for (int len = 2; len <= M; len <<= 1) {
    const int half = len >> 1, step = M / len;
    for (int i = 0; i < M; i += len) {
      for (int j = 0; j < half; ++j) {
        const float cr = wr[j * step], ci = wi[j * step];
        const float br = re[i + j + half] * cr - im[i + j + half] * ci;
        const float bi = re[i + j + half] * ci + im[i + j + half] * cr;
        re[i + j + half] = re[i + j] - br;
        im[i + j + half] = im[i + j] - bi;
        re[i + j] += br;
        im[i + j] += bi;
      }
    }
  }
}

int pfbo_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

size_t pfbo_channelize_f32_i16(const int16_t* iq, size_t n, int bit_width,
                               const float* h, int M, int P, int D, int off,
                               float* out, int num_threads) {
  if (M < 2 || (M & (M - 1))) return 0;
  if (off < 0) off = D - 1;
  const size_t F = n / (size_t)D;
  const float scale = (float)ldexp(1.0, -(bit_width - 1));
  float* wr = (float*)malloc(sizeof(float) * 2 * (size_t)M);
  int* rev = (int*)malloc(sizeof(int) * (size_t)M);
  if (!wr || !rev) { free(wr); free(rev); return 0; }
  float* wi = wr + M;
  for (int j = 0; j < M; ++j) {
    wr[j] = (float)cos(2.0 * M_PI * j / M);
    wi[j] = (float)sin(2.0 * M_PI * j / M);
  }
  rev[0] = 0;
  for (int i = 1, j = 0; i < M; ++i) {
    int bit = M >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    rev[i] = j;
  }
#ifdef _OPENMP
  if (num_threads > 0) omp_set_num_threads(num_threads);
#else
  (void)num_threads;
#endif
#pragma omp parallel
  {
    float* ur = (float*)malloc(sizeof(float) * 2 * (size_t)M);
    float* ui = ur + M;
#pragma omp for schedule(static)
    for (long long mm = 0; mm < (long long)F; ++mm) {
      const size_t m = (size_t)mm;
      for (int p = 0; p < M; ++p) {
        float ar = 0.f, ai = 0.f;
        for (int q = 0; q < P; ++q) {
          const long long s = (long long)(m * (size_t)D) + off - p - (long long)M * q;
          if (s >= 0) {
            const float c = h[p + M * q];
            ar += c * (float)iq[2 * s];
            ai += c * (float)iq[2 * s + 1];
          }
        }
        ur[p] = ar * scale;
        ui[p] = ai * scale;
      }
      fft_pos_f32(ur, ui, M, wr, wi, rev);
      float* o = out + 2 * m * (size_t)M;
      for (int k = 0; k < M; ++k) {
        o[2 * k] = ur[k];
        o[2 * k + 1] = ui[k];
      }
    }
    free(ur);
  }
  free(wr);
  free(rev);
  return F;
}

/* ------------------------------------------------------------------------- */
/* .iq header parse: restates matlab/convert_my_iq_to_mat.m:40-98 field by    */
/* field over the layout of cpp/IqPacket.h:9-25.                              */

static uint32_t rd_u32(const uint8_t* p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
static uint64_t rd_u64(const uint8_t* p) {
  return (uint64_t)rd_u32(p) | ((uint64_t)rd_u32(p + 4) << 32);
}

int pfbo_parse_iq_header(const uint8_t* b, size_t len, pfbo_iq_header* o) {
  if (!b || !o || len < 4) return -1;
  memset(o, 0, sizeof(*o));
  o->marker = rd_u32(b);
  switch (o->marker) { /* :42-57 */
    case 0x01010101u: o->file_format = 1; break;
    case 0x02020202u: o->file_format = 2; break;
    case 0x03030303u: o->file_format = 3; break;
    case 0x00000000u: o->file_format = 2; break; /* :43-45: announced as big endian, read as format 2 in native order */
    default: return -2;
  }
  o->header_bytes = (o->file_format == 1) ? 104u : 112u;
  if (len < o->header_bytes) return -3;
  size_t pos = 4;
  o->link_speed = rd_u32(b + pos); pos += 4;
  if (o->file_format == 1) { o->frequency_hz = rd_u32(b + pos); pos += 4; }   /* :63-65 */
  else { o->frequency_hz = rd_u64(b + pos); pos += 8; }                       /* :66-67 */
  o->bandwidth_hz = rd_u32(b + pos); pos += 4;
  o->sample_rate_sps = rd_u32(b + pos); pos += 4;
  if (o->file_format >= 3) {                                                  /* :73-77 */
    float g; uint32_t u = rd_u32(b + pos); memcpy(&g, &u, 4); o->rx_gain_db = (double)g;
  } else {
    o->rx_gain_db = (double)rd_u32(b + pos);
  }
  pos += 4;
  o->num_samples = rd_u32(b + pos); pos += 4;
  o->bit_width = rd_u32(b + pos); pos += 4;
  if (o->file_format >= 2) { o->spare0 = rd_u32(b + pos); pos += 4; }        /* :82-84 */
  memcpy(o->board_name, b + pos, 16); pos += 16;
  memcpy(o->serial_number, b + pos, 16); pos += 16;
  memcpy(o->fpga_version, b + pos, 16); pos += 16;
  memcpy(o->fw_version, b + pos, 16); pos += 16;
  uint64_t t = rd_u64(b + pos); memcpy(&o->sample_start_time, &t, 8); pos += 8;
  if (pos != o->header_bytes) return -4;
  if (o->bit_width > 0 && o->bit_width <= 8) o->bytes_per_sample = 2;        /* :92-98 */
  else if (o->bit_width > 8 && o->bit_width <= 16) o->bytes_per_sample = 4;
  else return -5;
  return 0;
}

/* ------------------------------------------------------------------------- */
/* Channelized PDW extraction: matlab/create_pdws_channelized.m:64-143         */

static int cmp_double(const void* a, const void* b) {
  const double x = *(const double*)a, y = *(const double*)b;
  return (x > y) - (x < y);
}

/* MATLAB median: mean of the two middle elements for even counts */
static double median_of(double* tmp, size_t n) {
  qsort(tmp, n, sizeof(double), cmp_double);
  return (n & 1) ? tmp[n / 2] : 0.5 * (tmp[n / 2 - 1] + tmp[n / 2]);
}

size_t pfbo_extract_pdws(const double* yr, const double* yi, size_t F, int M, int decim,
                         double fs_in, double fc, double sample_start_time,
                         double snr_threshold_db, int matlab_quirks,
                         pfbo_pdw* out, size_t max_out) {
  const double fs = fs_in / (double)decim; /* :62 (decim = M in the reference) */
  double* tmp = (double*)malloc(sizeof(double) * (F + 1));
  double* nf = (double*)malloc(sizeof(double) * (size_t)M * 2);
  double* bin_freqs = nf + M;
  if (!tmp || !nf) { free(tmp); free(nf); return 0; }
  pfbo_center_frequencies(M, fs_in, bin_freqs); /* :42, taken before fs is decimated */
#define MAG(j, b) hypot(yr[(j) * (size_t)M + (b)], yi[(j) * (size_t)M + (b)])
#define PHASE(j, b) (atan2(yi[(j) * (size_t)M + (b)], yr[(j) * (size_t)M + (b)]) * (180.0 / M_PI))
  for (int b = 0; b < M; ++b) { /* :73 column median */
    for (size_t j = 0; j < F; ++j) tmp[j] = MAG(j, b);
    nf[b] = median_of(tmp, F);
  }
  const double gain = pow(10.0, snr_threshold_db / 10.0); /* :74-75 (dB on magnitude with /10) */
  size_t count = 0;
  for (int b = 0; b < M; ++b) {
    /* :80 binFreqs(bin) with `bin` the column of the fftshift-ed matrix.  Bit 1: binFreqs is the FFT-ordered
     * (unshifted) list -- what :42 yields IF MathWorks' centerFrequencies returns that order (unpinned);
     * otherwise the column's true centre frequency (= the centred list indexed by the shifted column). */
    const double f_bin = (matlab_quirks & 2) ? bin_freqs[b] : bin_freqs[(b + (M + 1) / 2) % M];
    const double fc_chan = fc + f_bin;
    const double thr = nf[b] * gain;
    int active = 0, saturated = 0;
    size_t toa = 0; /* 0-based here; the reference's 1-based toa = toa+1 */
    for (size_t j = 0; j < F; ++j) {
      const double mg = MAG(j, b);
      if (!active) {
        if (mg >= thr) { active = 1; toa = j; saturated = 0; } /* :87-91 */
      } else if (mg <= thr) {                                  /* :94 trailing edge */
        active = 0;
        const size_t len = j - toa + 1;
        for (size_t t = 0; t < len; ++t) tmp[t] = MAG(toa + t, b);
        const double amp = median_of(tmp, len);                /* :101 */
        const int pcol = (matlab_quirks & 1) ? 0 : b;          /* :114 linear index -> column 1 (bit 0) */
        for (size_t t = 0; t + 1 < len; ++t) {
          double d = PHASE(toa + t + 1, pcol) - PHASE(toa + t, pcol);
          if (d < -180.0) d += 360.0;                          /* :115 */
          if (d > 180.0) d -= 360.0;                           /* :116 */
          tmp[t] = d;
        }
        const double med = median_of(tmp, len - 1);            /* :117 */
        if (count < max_out) {
          pfbo_pdw* o = &out[count];
          o->toa = ((double)(toa + 1) / fs) + sample_start_time; /* :98, 1-based index */
          o->snr = 10.0 * log10(amp / nf[b]);                  /* :105 */
          o->pw = (double)(j - toa) / fs;                      /* :110 */
          o->freq = fc_chan + (fs / (360.0 / med));            /* :122 */
          o->sat = saturated;
          o->bin = b;
          o->mag = amp;
        }
        ++count;
      } else {
        if (fabs(yr[j * (size_t)M + b]) >= 0.9999 || fabs(yi[j * (size_t)M + b]) >= 0.9999)
          saturated = 1;                                       /* :130-132 */
      }
    }
  }
#undef MAG
#undef PHASE
  free(tmp);
  free(nf);
  return count;
}

/* matlab/create_pdws.m:30-105: the same state machine on the raw stream, with hysteresis. */
size_t pfbo_extract_pdws_raw(const double* xr, const double* xi, size_t n, double fs, double fc,
                             double sample_start_time, double snr_threshold_db, double trailing_threshold_db,
                             pfbo_pdw* out, size_t max_out, double* noise_floor) {
  double* tmp = (double*)malloc(sizeof(double) * (n + 1));
  if (!tmp) return 0;
#define MAG(j) hypot(xr[j], xi[j])                         /* :38 */
#define PHASE(j) (atan2(xi[j], xr[j]) * (180.0 / M_PI))    /* :39 */
  for (size_t j = 0; j < n; ++j) tmp[j] = MAG(j);
  const double nf = median_of(tmp, n);                                        /* :44 */
  const double lead = nf * pow(10.0, snr_threshold_db / 10.0);                /* :45-46 */
  const double trail = nf * pow(10.0, trailing_threshold_db / 10.0);          /* :47 */
  if (noise_floor) *noise_floor = nf;
  size_t count = 0, toa = 0;
  int active = 0, saturated = 0;
  for (size_t j = 0; j < n; ++j) {
    const double mg = MAG(j);
    if (!active) {
      if (mg >= lead) { active = 1; toa = j; saturated = 0; }                 /* :57-60 */
    } else if (mg <= trail) {                                                 /* :63 */
      active = 0;
      const size_t len = j - toa + 1;
      for (size_t t = 0; t < len; ++t) tmp[t] = MAG(toa + t);
      const double amp = median_of(tmp, len);                                 /* :70 */
      for (size_t t = 0; t + 1 < len; ++t) {
        double d = PHASE(toa + t + 1) - PHASE(toa + t);                       /* :83 */
        if (d < -180.0) d += 360.0;                                           /* :84 */
        if (d > 180.0) d -= 360.0;                                            /* :85 */
        tmp[t] = d;
      }
      const double med = median_of(tmp, len - 1);                             /* :86 */
      if (count < max_out) {
        pfbo_pdw* o = &out[count];
        o->toa = ((double)(toa + 1) / fs) + sample_start_time;                /* :67, 1-based index */
        o->mag = amp;
        o->snr = 10.0 * log10(amp / nf);                                      /* :74 */
        o->pw = (double)(j - toa) / fs;                                       /* :79 */
        o->freq = fc + (fs / (360.0 / med));                                  /* :91 */
        o->sat = saturated;
        o->bin = 0;
      }
      ++count;
    } else if (fabs(xr[j]) >= 0.9999 || fabs(xi[j]) >= 0.9999) {             /* :100-102 */
      saturated = 1;
    }
  }
#undef MAG
#undef PHASE
  free(tmp);
  return count;
}
