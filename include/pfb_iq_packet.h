/*
 * pfb_iq_packet.h -- the reference's .iq record format, restated.
 *
 * Replaces / stays byte-compatible with:
 *   struct IqPacket               /root/reference/cpp/IqPacket.h:9-25
 *   record writer                 /root/reference/cpp/blade_record_iq_12bit.cpp:318-323
 *                                 /root/reference/cpp/usrp_record_iq_12bit.cpp:224-227
 *   record reader                 /root/reference/matlab/convert_my_iq_to_mat.m:40-102
 *   getFilenameStr / FILENAME_LENGTH
 *                                 /root/reference/cpp/Helper.cpp:6-23, cpp/Helper.h:7
 *
 * A record is this 112-byte little-endian header followed by numSamples
 * interleaved I,Q pairs: int8 pairs when bitWidth <= 8, int16 pairs when
 * 8 < bitWidth <= 16 (convert_my_iq_to_mat.m:92-98).  File-format 1 records
 * (marker 0x01010101, hand-written by matlab/generate_training_iq.m:109-123)
 * have a 104-byte header: 32-bit frequency, integer gain and no spare0 word.
 *
 * Plain C, no dependencies; usable from the recorders' C++ as-is.
 */
#ifndef PFB_IQ_PACKET_H
#define PFB_IQ_PACKET_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PFB_IQ_MARKER_FMT1 0x01010101u /* matlab/generate_training_iq.m:109, cpp/tx_rx_pulses_usrp.cpp:182 */
#define PFB_IQ_MARKER_FMT2 0x02020202u /* cpp/blade_record_iq_12bit.cpp:248 */
#define PFB_IQ_MARKER_FMT3 0x03030303u /* cpp/usrp_record_iq_12bit.cpp:159 */
/* matlab/convert_my_iq_to_mat.m:42-45: marker 0 is announced as "big endian" and read as file format 2 --
 * but the script never re-opens the file with another byte order, so every field is still read in native
 * (little-endian) order.  No writer in the reference produces it; the parser does what the reader does. */
#define PFB_IQ_MARKER_ZERO 0x00000000u
#define PFB_IQ_HEADER_BYTES 112u
#define PFB_IQ_HEADER_BYTES_FMT1 104u
#define PFB_FILENAME_LENGTH 80 /* cpp/Helper.h:7 */

/* Same field order, types and natural alignment as cpp/IqPacket.h:11-24. */
typedef struct pfb_iq_packet {
  uint32_t endianness;      /* @0   marker word, selects the file format      */
  uint32_t linkSpeed;       /* @4                                              */
  uint64_t frequencyHz;     /* @8                                              */
  uint32_t bandwidthHz;     /* @16                                             */
  uint32_t sampleRateSps;   /* @20                                             */
  float    rxGainDb;        /* @24                                             */
  uint32_t numSamples;      /* @28  complex samples in the payload             */
  uint32_t bitWidth;        /* @32  8, 12 or 16: scale is 2^(bitWidth-1)       */
  uint32_t spare0;          /* @36                                             */
  char     boardName[16];   /* @40                                             */
  char     serialNumber[16];/* @56                                             */
  char     fpgaVersion[16]; /* @72                                             */
  char     fwVersion[16];   /* @88                                             */
  double   sampleStartTime; /* @104 UTC seconds                                */
} pfb_iq_packet;

#if defined(__cplusplus)
static_assert(sizeof(pfb_iq_packet) == PFB_IQ_HEADER_BYTES, "IqPacket is 112 bytes");
static_assert(offsetof(pfb_iq_packet, frequencyHz) == 8, "offset");
static_assert(offsetof(pfb_iq_packet, rxGainDb) == 24, "offset");
static_assert(offsetof(pfb_iq_packet, bitWidth) == 32, "offset");
static_assert(offsetof(pfb_iq_packet, boardName) == 40, "offset");
static_assert(offsetof(pfb_iq_packet, sampleStartTime) == 104, "offset");
#else
_Static_assert(sizeof(pfb_iq_packet) == PFB_IQ_HEADER_BYTES, "IqPacket is 112 bytes");
_Static_assert(offsetof(pfb_iq_packet, frequencyHz) == 8, "offset");
_Static_assert(offsetof(pfb_iq_packet, rxGainDb) == 24, "offset");
_Static_assert(offsetof(pfb_iq_packet, bitWidth) == 32, "offset");
_Static_assert(offsetof(pfb_iq_packet, boardName) == 40, "offset");
_Static_assert(offsetof(pfb_iq_packet, sampleStartTime) == 104, "offset");
#endif

/* Parsed view of any of the three on-disk header variants. */
typedef struct pfb_iq_info {
  pfb_iq_packet packet;       /* fields widened into the current (fmt 3) struct   */
  int32_t  file_format;       /* 1, 2 or 3 (marker 0 reads as 2)                  */
  uint32_t header_bytes;      /* 104 or 112: payload starts here                  */
  uint32_t bytes_per_sample;  /* 2 (int8 I,Q) or 4 (int16 I,Q)                    */
  uint32_t sample_format;     /* PFB_FMT_INT8_IQ or PFB_FMT_INT16_IQ              */
  double   rx_gain_as_read;   /* what convert_my_iq_to_mat.m:73-77 would report:  */
                              /* fmt<3 reads the gain word as uint32              */
} pfb_iq_info;

/* Parse a header from the first bytes of a record.  Returns PFB_OK or a negative
 * pfb_status (PFB_ERR_BAD_FORMAT for an unknown marker / unsupported bit width,
 * PFB_ERR_BAD_ARG for a short buffer).  Mirrors convert_my_iq_to_mat.m:40-98. */
int pfb_iq_parse_header(const void* bytes, size_t len, pfb_iq_info* out);

/* Fill a fmt-2/3 header the way the recorders do before fout.write(&packet)
 * (cpp/blade_record_iq_12bit.cpp:246-261).  Strings are truncated to 15 chars. */
void pfb_iq_fill_packet(pfb_iq_packet* p, uint32_t marker, uint64_t frequency_hz,
                        uint32_t bandwidth_hz, uint32_t sample_rate_sps, float rx_gain_db,
                        uint32_t num_samples, uint32_t bit_width, const char* board_name,
                        const char* serial_number, double sample_start_time);

/* UTC "YYYY_MM_DD_hh_mm_ss_mmm.iq" from milliseconds since the epoch
 * (cpp/Helper.cpp:6-23).  Returns the number of characters written. */
int pfb_iq_filename(int64_t epoch_ms, char* out, int out_len);

#ifdef __cplusplus
}
#endif
#endif
