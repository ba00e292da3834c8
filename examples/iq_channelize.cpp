// iq_channelize.cpp -- C++ host loop over the C ABI, shaped like the recorders' loop body
// (/root/reference/cpp/blade_record_iq_12bit.cpp:287-325: one dwell buffer per iteration, blocking,
// status ints, caller-owned buffers).  Reads .iq records (header + interleaved int payload), channelizes
// each with libpfb_channelizer.so and writes <name>.chan (raw complex64, frame-major F x M), then runs the rest of
// create_pdws_channelized.m's loop body in one call (pfb_pdw_from_iq_file: record -> channels -> PDWs, the matrix
// never leaving the GPU) and writes <name>.pdw (pfb_pdw structs).
//
//   hipcc -O2 -Iinclude examples/iq_channelize.cpp -Lsdr_channelizer_amd -lpfb_channelizer \
//         -Wl,-rpath,$PWD/sdr_channelizer_amd -o examples/iq_channelize
//   ./examples/iq_channelize 64 12 capture1.iq capture2.iq
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "pfb_channelizer.h"

int main(int argc, char** argv) {
  if (argc < 4) {
    std::fprintf(stderr, "usage: %s <numBands> <tapsPerBand> <file.iq> [...]\n", argv[0]);
    return 1;
  }
  const uint32_t M = (uint32_t)std::atoi(argv[1]), P = (uint32_t)std::atoi(argv[2]);
  std::vector<float> taps((size_t)M * P);
  int status = pfb_design_prototype(M, P, 80.0, taps.data());
  pfb_handle* ch = nullptr;
  uint32_t open_format = ~0u, open_width = 0;

  for (int a = 3; a < argc; ++a) {
    std::ifstream fin(argv[a], std::ifstream::binary);
    std::vector<char> head(PFB_IQ_HEADER_BYTES);
    fin.read(head.data(), (std::streamsize)head.size());
    pfb_iq_info info;
    status = pfb_iq_parse_header(head.data(), (size_t)fin.gcount(), &info);
    if (status != PFB_OK) {
      std::printf("%s: %s\n", argv[a], pfb_strerror(status));
      continue;
    }
    const uint32_t n = info.packet.numSamples;
    std::vector<char> iq((size_t)n * info.bytes_per_sample);  // std::complex<int16_t>[n] or <int8_t>[n]
    fin.seekg(info.header_bytes);
    fin.read(iq.data(), (std::streamsize)iq.size());

    if (!ch || open_format != info.sample_format || open_width != info.packet.bitWidth) {
      if (ch) pfb_destroy(ch);
      pfb_config cfg{};
      cfg.struct_size = sizeof(cfg);
      cfg.num_channels = M;           // numBands = fs*1e-6 in the scripts
      cfg.taps_per_channel = P;
      cfg.decimation = 0;             // = M
      cfg.taps = taps.data();
      cfg.sample_format = info.sample_format;
      cfg.bit_width = info.packet.bitWidth;
      cfg.output_layout = PFB_LAYOUT_FRAME_MAJOR;
      cfg.flags = PFB_FLAG_FFTSHIFT;  // create_pdws_channelized.m:60
      cfg.input_offset = -1;
      cfg.device_id = -1;
      status = pfb_create(&cfg, &ch);
      if (status != PFB_OK) {
        std::printf("Failed to create channelizer: %s (%s)\n", pfb_strerror(status), pfb_last_error_detail());
        return __LINE__;
      }
      open_format = info.sample_format;
      open_width = info.packet.bitWidth;
    } else {
      pfb_reset(ch);  // a fresh dsp.Channelizer per file, create_pdws_channelized.m:33
    }

    uint64_t frames = 0;
    pfb_frames_for(ch, n, &frames);
    std::vector<std::complex<float>> out((size_t)frames * M);
    status = pfb_process(ch, iq.data(), n, out.data(), frames, &frames, PFB_MEM_HOST);
    if (status != PFB_OK) {
      std::printf("Channelizer failed: %s (%s)\n", pfb_strerror(status), pfb_last_error_detail());
      continue;
    }
    const double fs_out = (double)info.packet.sampleRateSps / M;  // create_pdws_channelized.m:62
    std::printf("%s: %u samples, %u-bit -> %llu frames x %u channels at %.1f Hz (%s)\n", argv[a], n,
                info.packet.bitWidth, (unsigned long long)frames, M, fs_out, pfb_last_kernel(ch));
    std::ofstream fout(std::string(argv[a]) + ".chan", std::ofstream::binary);
    fout.write((const char*)out.data(), (std::streamsize)(out.size() * sizeof(out[0])));

    // create_pdws_channelized.m:64-143 on the same record (snrThreshold 15 dB, the script's indexing quirks kept)
    std::vector<pfb_pdw> pdws(1 << 16);
    uint64_t count = 0;
    pfb_reset(ch);
    status = pfb_pdw_from_iq_file(ch, argv[a], 15.0, PFB_PDW_MATLAB_QUIRKS, pdws.data(), pdws.size(), &count, nullptr, nullptr);
    if (status != PFB_OK) {
      std::printf("PDW extraction failed: %s (%s)\n", pfb_strerror(status), pfb_pdw_last_error_detail());
      continue;
    }
    if (count > pdws.size()) count = pdws.size();
    std::printf("%s: %llu PDWs\n", argv[a], (unsigned long long)count);
    std::ofstream fpdw(std::string(argv[a]) + ".pdw", std::ofstream::binary);
    fpdw.write((const char*)pdws.data(), (std::streamsize)(count * sizeof(pfb_pdw)));
  }
  if (ch) pfb_destroy(ch);
  return status;
}
