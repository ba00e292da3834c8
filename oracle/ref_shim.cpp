// ref_shim.cpp -- ORACLE-SIDE glue (test infrastructure, NOT the product).
//
// Compiles the reference's own record-format sources *where they lie* under
// /root/reference (cpp/IqPacket.h, cpp/Helper.h, cpp/Helper.cpp -- the only part
// of the hot path the reference owns as compilable code) and
// exposes them through a tiny C ABI so tests can pin our restatements
// (include/pfb_iq_packet.h, sdr_channelizer_amd/iqfile.py) against the real
// struct layout and filename format.  Built by oracle/Makefile into
// oracle/_ref/libref_iqpacket.so; nothing from the reference is copied into the repo.
#include <chrono>
#include <cstddef>
#include <cstdint>
#include <cstring>

#include "Helper.h"    // -I/root/reference/cpp
#include "IqPacket.h"  // -I/root/reference/cpp

extern "C" {

unsigned ref_sizeof_iqpacket() { return static_cast<unsigned>(sizeof(IqPacket)); }

// offsets in declaration order (cpp/IqPacket.h:11-24); returns the field count
unsigned ref_iqpacket_offsets(unsigned* out) {
  unsigned i = 0;
  out[i++] = offsetof(IqPacket, endianness);
  out[i++] = offsetof(IqPacket, linkSpeed);
  out[i++] = offsetof(IqPacket, frequencyHz);
  out[i++] = offsetof(IqPacket, bandwidthHz);
  out[i++] = offsetof(IqPacket, sampleRateSps);
  out[i++] = offsetof(IqPacket, rxGainDb);
  out[i++] = offsetof(IqPacket, numSamples);
  out[i++] = offsetof(IqPacket, bitWidth);
  out[i++] = offsetof(IqPacket, spare0);
  out[i++] = offsetof(IqPacket, boardName);
  out[i++] = offsetof(IqPacket, serialNumber);
  out[i++] = offsetof(IqPacket, fpgaVersion);
  out[i++] = offsetof(IqPacket, fwVersion);
  out[i++] = offsetof(IqPacket, sampleStartTime);
  return i;
}

// Fill a real IqPacket the way the recorders do (cpp/blade_record_iq_12bit.cpp:
// 248-261,296,314) and copy its bytes out, exactly what fout.write(&packet) emits.
unsigned ref_make_header(unsigned marker, unsigned link_speed, unsigned long long freq_hz,
                         unsigned bw_hz, unsigned fs_sps, float gain_db, unsigned num_samples,
                         unsigned bit_width, const char* board, const char* serial,
                         const char* fpga, const char* fw, double start_time,
                         unsigned char* out, unsigned out_len) {
  IqPacket p;
  std::memset(&p, 0, sizeof(p));
  p.endianness = marker;
  p.linkSpeed = link_speed;
  p.frequencyHz = freq_hz;
  p.bandwidthHz = bw_hz;
  p.sampleRateSps = fs_sps;
  p.rxGainDb = gain_db;
  p.numSamples = num_samples;
  p.bitWidth = bit_width;
  std::strncpy(p.boardName, board, sizeof(p.boardName) - 1);
  std::strncpy(p.serialNumber, serial, sizeof(p.serialNumber) - 1);
  std::strncpy(p.fpgaVersion, fpga, sizeof(p.fpgaVersion) - 1);
  std::strncpy(p.fwVersion, fw, sizeof(p.fwVersion) - 1);
  p.sampleStartTime = start_time;
  if (out_len < sizeof(p)) return 0;
  std::memcpy(out, &p, sizeof(p));
  return static_cast<unsigned>(sizeof(p));
}

// getFilenameStr (cpp/Helper.cpp:6-23) on a time given as integer milliseconds since epoch
void ref_get_filename_str(long long epoch_ms, char* out, int out_len) {
  const std::chrono::system_clock::time_point tp{
      std::chrono::duration_cast<std::chrono::system_clock::duration>(
          std::chrono::milliseconds(epoch_ms))};
  getFilenameStr(tp, out, out_len);
}

int ref_filename_length() { return FILENAME_LENGTH; }
int ref_iq_file_format() { return IQ_FILE_FORMAT; }
}
