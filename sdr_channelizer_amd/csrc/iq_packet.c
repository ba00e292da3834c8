/*
 * iq_packet.c -- .iq record header parse / fill / filename (see include/pfb_iq_packet.h).
 * Restates /root/reference/matlab/convert_my_iq_to_mat.m:40-98 (reader),
 * /root/reference/cpp/blade_record_iq_12bit.cpp:246-261 (writer fields) and
 * /root/reference/cpp/Helper.cpp:6-23 (file name).  Plain C, no GPU.
 */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <string.h>
#include <time.h>

#include "pfb_channelizer.h"

static uint32_t rd32(const unsigned char* p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
static uint64_t rd64(const unsigned char* p) { return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32); }

int pfb_iq_parse_header(const void* bytes, size_t len, pfb_iq_info* out) {
  if (!bytes || !out || len < 4) return PFB_ERR_BAD_ARG;
  const unsigned char* b = (const unsigned char*)bytes;
  memset(out, 0, sizeof(*out));
  pfb_iq_packet* k = &out->packet;
  k->endianness = rd32(b);
  switch (k->endianness) { /* convert_my_iq_to_mat.m:42-57 */
    case PFB_IQ_MARKER_FMT1: out->file_format = 1; break;
    case PFB_IQ_MARKER_FMT2: out->file_format = 2; break;
    case PFB_IQ_MARKER_FMT3: out->file_format = 3; break;
    case PFB_IQ_MARKER_ZERO: out->file_format = 2; break; /* :43-45 "assume latest file format"; fields stay little-endian */
    default: return PFB_ERR_BAD_FORMAT;                   /* :55-56 error(...) */
  }
  out->header_bytes = out->file_format == 1 ? PFB_IQ_HEADER_BYTES_FMT1 : PFB_IQ_HEADER_BYTES;
  if (len < out->header_bytes) return PFB_ERR_BAD_ARG;
  size_t pos = 4;
  k->linkSpeed = rd32(b + pos); pos += 4;
  if (out->file_format == 1) { k->frequencyHz = rd32(b + pos); pos += 4; } /* :63-65 */
  else { k->frequencyHz = rd64(b + pos); pos += 8; }                       /* :66-67 */
  k->bandwidthHz = rd32(b + pos); pos += 4;
  k->sampleRateSps = rd32(b + pos); pos += 4;
  {
    const uint32_t g = rd32(b + pos); pos += 4;
    memcpy(&k->rxGainDb, &g, 4);
    /* :73-77: only fmt >= 3 is read as float32; older markers are read as uint32 */
    out->rx_gain_as_read = out->file_format >= 3 ? (double)k->rxGainDb : (double)g;
    if (out->file_format < 3) k->rxGainDb = (float)g;
  }
  k->numSamples = rd32(b + pos); pos += 4;
  k->bitWidth = rd32(b + pos); pos += 4;
  if (out->file_format >= 2) { k->spare0 = rd32(b + pos); pos += 4; }     /* :82-84 */
  memcpy(k->boardName, b + pos, 16); pos += 16;
  memcpy(k->serialNumber, b + pos, 16); pos += 16;
  memcpy(k->fpgaVersion, b + pos, 16); pos += 16;
  memcpy(k->fwVersion, b + pos, 16); pos += 16;
  {
    const uint64_t t = rd64(b + pos); pos += 8;
    memcpy(&k->sampleStartTime, &t, 8);
  }
  if (k->bitWidth > 0 && k->bitWidth <= 8) {                               /* :92-98 */
    out->bytes_per_sample = 2; out->sample_format = PFB_FMT_INT8_IQ;
  } else if (k->bitWidth > 8 && k->bitWidth <= 16) {
    out->bytes_per_sample = 4; out->sample_format = PFB_FMT_INT16_IQ;
  } else {
    return PFB_ERR_BAD_FORMAT;
  }
  return PFB_OK;
}

void pfb_iq_fill_packet(pfb_iq_packet* p, uint32_t marker, uint64_t frequency_hz, uint32_t bandwidth_hz,
                        uint32_t sample_rate_sps, float rx_gain_db, uint32_t num_samples, uint32_t bit_width,
                        const char* board_name, const char* serial_number, double sample_start_time) {
  memset(p, 0, sizeof(*p));
  p->endianness = marker;
  p->frequencyHz = frequency_hz;
  p->bandwidthHz = bandwidth_hz;
  p->sampleRateSps = sample_rate_sps;
  p->rxGainDb = rx_gain_db;
  p->numSamples = num_samples;
  p->bitWidth = bit_width;
  if (board_name) strncpy(p->boardName, board_name, sizeof(p->boardName) - 1);
  if (serial_number) strncpy(p->serialNumber, serial_number, sizeof(p->serialNumber) - 1);
  p->sampleStartTime = sample_start_time;
}

int pfb_iq_filename(int64_t epoch_ms, char* out, int out_len) {
  if (!out || out_len <= 0) return PFB_ERR_BAD_ARG;
  int64_t secs = epoch_ms / 1000, ms = epoch_ms % 1000;
  if (ms < 0) { ms += 1000; secs -= 1; }
  const time_t tt = (time_t)secs;
  struct tm utc;
  gmtime_r(&tt, &utc);
  return snprintf(out, (size_t)out_len, "%04d_%02d_%02d_%02d_%02d_%02d_%03d.iq", utc.tm_year + 1900,
                  utc.tm_mon + 1, utc.tm_mday, utc.tm_hour, utc.tm_min, utc.tm_sec, (int)ms);
}
