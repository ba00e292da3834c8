// sharded_rccl.cpp -- a C++ host time-sharding ONE stream over every GPU of the node through the C ABI, with the halo
// moved by RCCL (ncclSend / ncclRecv over xGMI) from the library's exchange callback.
//
// The reference channelizes a whole record in one call (matlab/create_pdws_channelized.m:57) from a single-threaded
// recorder loop (cpp/blade_record_iq_12bit.cpp:287-325); this is what that loop becomes on an 8-GPU node: one host
// thread and one handle per device, one contiguous segment each, one neighbour exchange of (P-1)*M raw samples per
// call, hidden under the frames that do not need it (pfb_process_shard_async).  The output, gathered in rank
// order, is bit-identical to one device channelizing the whole stream -- checked below against device 0.
//
// Build (examples/Makefile-free):  hipcc -O2 -std=c++17 -Iinclude examples/sharded_rccl.cpp -o examples/sharded_rccl \
//                                        -Lsdr_channelizer_amd -lpfb_channelizer -lrccl -Wl,-rpath,$PWD/sdr_channelizer_amd
// Run:    examples/sharded_rccl [frames_per_gpu]      (uses every visible GPU; with one GPU the exchange is skipped)
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <complex>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "pfb_channelizer.h"

#define CK(x) do { if ((x) != 0) { std::fprintf(stderr, "%s failed (line %d)\n", #x, __LINE__); std::exit(1); } } while (0)

struct Peer { ncclComm_t comm; };

// the library's exchange callback: enqueue both transfers on the side stream it hands over, return at once
static int exchange(void* user, const void* d_send, void* d_recv, size_t bytes, int send_to, int recv_from, void* stream) {
  Peer* p = static_cast<Peer*>(user);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (ncclGroupStart() != ncclSuccess) return 1;
  int rc = 0;  // the group is ALWAYS closed: an open group would swallow every later RCCL call of this thread
  if (send_to >= 0 && ncclSend(d_send, bytes, ncclUint8, send_to, p->comm, s) != ncclSuccess) rc = 2;
  if (rc == 0 && recv_from >= 0 && ncclRecv(d_recv, bytes, ncclUint8, recv_from, p->comm, s) != ncclSuccess) rc = 3;
  if (ncclGroupEnd() != ncclSuccess && rc == 0) rc = 4;
  return rc;  // the first failure; non-zero makes pfb_process_shard_async return PFB_ERR_COMM
}

int main(int argc, char** argv) {
  const uint32_t M = 64, P = 12;
  const uint64_t frames = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 1u << 16;
  int world = pfb_device_count();
  if (world < 1) { std::fprintf(stderr, "no HIP device\n"); return 1; }
  const uint64_t n = frames * M;  // samples per GPU
  std::vector<float> taps((size_t)M * P);
  CK(pfb_design_prototype(M, P, 80.0, taps.data()));
  // one stream, cut into `world` segments: a deterministic 12-bit pattern (the recorders' int16 I,Q layout)
  std::vector<int16_t> iq((size_t)world * n * 2);
  uint32_t lcg = 12345;
  for (auto& v : iq) { lcg = lcg * 1664525u + 1013904223u; v = (int16_t)((int)(lcg >> 20) - 2048); }

  std::vector<ncclComm_t> comms((size_t)world);
  if (world > 1) CK(ncclCommInitAll(comms.data(), world, nullptr));
  std::vector<std::vector<std::complex<float>>> out((size_t)world, std::vector<std::complex<float>>((size_t)frames * M));
  std::vector<std::thread> threads;
  for (int r = 0; r < world; ++r) {
    threads.emplace_back([&, r] {  // one host thread per device, like one recorder loop per radio
      CK(hipSetDevice(r));
      pfb_config cfg{};
      cfg.struct_size = sizeof(cfg); cfg.num_channels = M; cfg.taps_per_channel = P; cfg.taps = taps.data();
      cfg.sample_format = PFB_FMT_INT16_IQ; cfg.bit_width = 12; cfg.input_offset = -1; cfg.device_id = r;
      pfb_handle* h = nullptr;
      CK(pfb_create(&cfg, &h));
      Peer peer{world > 1 ? comms[(size_t)r] : nullptr};
      pfb_shard_config sc{};
      sc.struct_size = sizeof(sc); sc.rank = r; sc.world = world; sc.ring = 0;
      sc.exchange = world > 1 ? exchange : nullptr; sc.user = &peer;
      CK(pfb_shard_attach(h, &sc));
      void *d_iq = nullptr, *d_out = nullptr;
      CK(hipMalloc(&d_iq, n * 4));
      CK(hipMalloc(&d_out, frames * M * 8));
      CK(hipMemcpy(d_iq, iq.data() + (size_t)r * n * 2, n * 4, hipMemcpyHostToDevice));
      CK(pfb_set_frame_index(h, (uint64_t)r * frames));
      uint64_t got = 0;
      CK(pfb_process_shard_async(h, d_iq, n, d_out, frames, &got));  // exchange || interior frames, then the head frames
      CK(pfb_sync(h));
      CK(hipMemcpy(out[(size_t)r].data(), d_out, frames * M * 8, hipMemcpyDeviceToHost));
      CK(hipFree(d_iq)); CK(hipFree(d_out));
      CK(pfb_destroy(h));
    });
  }
  for (auto& t : threads) t.join();
  if (world > 1) for (auto c : comms) ncclCommDestroy(c);

  // reference: device 0 channelizes the whole stream in one call (host pointers: the library stages them)
  CK(hipSetDevice(0));
  pfb_config cfg{};
  cfg.struct_size = sizeof(cfg); cfg.num_channels = M; cfg.taps_per_channel = P; cfg.taps = taps.data();
  cfg.sample_format = PFB_FMT_INT16_IQ; cfg.bit_width = 12; cfg.input_offset = -1; cfg.device_id = 0;
  pfb_handle* h = nullptr;
  CK(pfb_create(&cfg, &h));
  std::vector<std::complex<float>> one((size_t)world * frames * M);
  uint64_t got = 0;
  CK(pfb_process(h, iq.data(), (uint64_t)world * n, one.data(), (uint64_t)world * frames, &got, PFB_MEM_HOST));
  CK(pfb_destroy(h));
  for (int r = 0; r < world; ++r)
    if (std::memcmp(out[(size_t)r].data(), one.data() + (size_t)r * frames * M, frames * M * 8) != 0) {
      std::fprintf(stderr, "shard %d differs from the single-device result\n", r);
      return 2;
    }
  std::printf("sharded_rccl: %d GPU(s) x %llu frames, halo %u samples: bit-identical to one device\n", world,
              (unsigned long long)frames, (P - 1) * M);
  return 0;
}
