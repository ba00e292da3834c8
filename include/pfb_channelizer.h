/*
 * pfb_channelizer.h -- C ABI of the MI355X polyphase-filterbank channelizer.
 *
 * This is the drop-in boundary for the reference's channelizer path.  The
 * reference has no FFI for it: the path is MATLAB script code around MathWorks'
 * dsp.Channelizer.  Each entry point below names the reference lines it
 * replaces (paths relative to /root/reference):
 *
 *   pfb_create            dsp.Channelizer(M) construction
 *                         matlab/channelizer_example.m:29-31
 *                         matlab/create_pdws_channelized.m:31-33
 *                         matlab/generate_channelized_training_iq.m:95-98
 *   pfb_process*          int->complex normalise + (conj) transpose + truncate +
 *                         channelizer(x) + fftshift(.,2)
 *                         matlab/channelizer_example.m:18-23,56,58
 *                         matlab/create_pdws_channelized.m:35-60
 *                         input buffer = what the recorders hold:
 *                         cpp/blade_record_iq_12bit.cpp:268,322 (&iq[FILTER_DELAY])
 *                         cpp/usrp_record_iq_08bit.cpp:179,226
 *   pfb_reset             a fresh dsp.Channelizer per file
 *                         matlab/create_pdws_channelized.m:33
 *   pfb_center_frequencies  centerFrequencies(channelizer, fs)
 *                         matlab/channelizer_example.m:60
 *                         matlab/create_pdws_channelized.m:42
 *   pfb_strerror          bladerf_strerror(status) convention
 *                         cpp/blade_record_iq_12bit.cpp:56-60
 *
 * Conventions follow the recorders: every call returns an int status, 0 = OK,
 * negative = failure (never throws, never aborts); the caller owns sample and
 * output buffers; a handle is used by one thread at a time.
 *
 * Arithmetic:
 *   x[n]   = (I[n] + jQ[n]) / 2^(bit_width-1)
 *   y_k[m] = sum_{n=0}^{MP-1} taps[n] e^{+j2 pi k n/M} x[m D + input_offset - n]
 * computed in fp32 on the GPU as M polyphase branches + an M-point FFT.
 * The handle is stateful like the MATLAB System object: samples from earlier
 * calls are the x[n<0] of later ones, and a call whose length is not a multiple
 * of D carries its tail to the next call.
 *
 * There is NO CPU fallback: without a HIP device every compute entry point
 * returns PFB_ERR_NO_DEVICE.
 */
#ifndef PFB_CHANNELIZER_H
#define PFB_CHANNELIZER_H

#include <stddef.h>
#include <stdint.h>

#include "pfb_iq_packet.h"

#ifdef __cplusplus
extern "C" {
#endif

#define PFB_ABI_VERSION 2

typedef enum pfb_status {
  PFB_OK = 0,
  PFB_ERR_BAD_ARG = -1,      /* null pointer, size mismatch, value out of range      */
  PFB_ERR_BAD_FORMAT = -2,   /* unknown .iq marker / sample format / bit width       */
  PFB_ERR_UNSUPPORTED = -3,  /* configuration the library has no kernel for          */
  PFB_ERR_NO_DEVICE = -4,    /* no HIP device / HIP runtime unavailable              */
  PFB_ERR_HIP = -5,          /* a HIP call failed (pfb_last_error_detail has more)   */
  PFB_ERR_NO_MEMORY = -6,
  PFB_ERR_CAPACITY = -7,     /* output buffer smaller than the frames produced       */
  PFB_ERR_INTERNAL = -8,     /* a C++ exception other than bad_alloc was stopped at the ABI   */
  PFB_ERR_COMM = -9          /* the caller's halo-exchange callback (RCCL) reported a failure */
} pfb_status;

typedef enum pfb_sample_format {
  PFB_FMT_INT8_IQ = 0,   /* std::complex<int8_t>  (blade/usrp_record_iq_08bit.cpp)   */
  PFB_FMT_INT16_IQ = 1,  /* std::complex<int16_t> (blade/usrp_record_iq_12bit.cpp)   */
  PFB_FMT_CF32 = 2       /* interleaved float I,Q, already normalised                */
} pfb_sample_format;

typedef enum pfb_output_layout {
  PFB_LAYOUT_FRAME_MAJOR = 0,   /* out[m*M + k]      (row-major F x M)               */
  PFB_LAYOUT_CHANNEL_MAJOR = 1  /* out[k*F + m]      (MATLAB's column-major F x M)   */
} pfb_output_layout;

enum {
  PFB_FLAG_FFTSHIFT = 1u << 0,        /* fftshift(out,2): create_pdws_channelized.m:60 */
  PFB_FLAG_CONJUGATE_INPUT = 1u << 1, /* iq' quirk: channelizer_example.m:23           */
  PFB_FLAG_DEROTATE = 1u << 2,        /* y_k[m] *= e^{-j2 pi k m D/M} (identity if D=M) */
  PFB_FLAG_MAGNITUDE = 1u << 3,       /* fused abs(): `out` receives float32 |y_k[m]| instead of  */
                                      /* complex64 (abs(channelizer(x)), channelizer_example.m:56; */
                                      /* mag = abs(iq), create_pdws_channelized.m:67): half the    */
                                      /* output bytes                                              */
  PFB_FLAG_POWER = 1u << 4            /* with PFB_FLAG_MAGNITUDE: |y_k[m]|^2 instead of |y_k[m]|   */
                                      /* (no square root; what a detector that thresholds power    */
                                      /* wants).  Without PFB_FLAG_MAGNITUDE: PFB_ERR_BAD_ARG      */
};

enum {
  PFB_MEM_HOST = 0,   /* iq/out are host pointers: the library stages them            */
  PFB_MEM_DEVICE = 1  /* iq/out are device pointers on the handle's GPU               */
};

typedef struct pfb_config {
  uint32_t struct_size;        /* = sizeof(pfb_config), for ABI growth               */
  uint32_t num_channels;       /* M  (reference: fs*1e-6, channelizer_example.m:29)  */
  uint32_t taps_per_channel;   /* P  (dsp.Channelizer default 12).  Fewer taps than a    */
                               /* fused shape has run on that shape with zeros appended  */
                               /* (pfb_history_samples reports the padded length)        */
  uint32_t decimation;         /* D: 0 or M = maximally decimated, M/2 = 2x oversampled */
  const float* taps;           /* M*P prototype taps h[n], copied at create          */
  uint32_t sample_format;      /* pfb_sample_format                                  */
  uint32_t bit_width;          /* 8, 12 or 16: scale 2^-(bit_width-1); ignored for CF32 */
  uint32_t output_layout;      /* pfb_output_layout                                  */
  uint32_t flags;              /* PFB_FLAG_*                                         */
  int32_t  input_offset;       /* 0..D-1, or -1 for the default D-1                  */
  int32_t  device_id;          /* HIP device ordinal, -1 = current device            */
} pfb_config;

typedef struct pfb_handle pfb_handle;

/* ---- lifecycle ------------------------------------------------------------ */
int pfb_create(const pfb_config* cfg, pfb_handle** out);
int pfb_destroy(pfb_handle* h);
/* zero the filter state, the carried tail and the frame counter */
int pfb_reset(pfb_handle* h);
/* launch on this hipStream_t from now on (default: the null stream).  Work already queued on the
 * previous stream is ordered in front of whatever the new stream runs next for this handle. */
int pfb_set_stream(pfb_handle* h, void* hip_stream);

/* ---- process -------------------------------------------------------------- */
/* Channelize num_samples complex samples.  `iq` is the recorder's buffer
 * (interleaved I,Q of cfg.sample_format).  Frames produced =
 * floor((carried + num_samples)/D) is written to *frames_out; `out` receives
 * frames*M complex64 values and must hold at least out_capacity_frames frames
 * (PFB_ERR_CAPACITY otherwise, with *frames_out = frames needed and no state
 * change).  With PFB_LAYOUT_CHANNEL_MAJOR the column stride is the number of
 * frames of this call.  Synchronous: results are complete on return. */
int pfb_process(pfb_handle* h, const void* iq, uint64_t num_samples, void* out,
                uint64_t out_capacity_frames, uint64_t* frames_out, uint32_t mem);
/* Same, device pointers only, enqueued on the handle's stream without a host
 * sync.  The input buffer must stay valid until pfb_sync (the state update
 * reads its tail on the stream). */
int pfb_process_async(pfb_handle* h, const void* d_iq, uint64_t num_samples, void* d_out,
                      uint64_t out_capacity_frames, uint64_t* frames_out);
int pfb_sync(pfb_handle* h);
/* Frames a call with num_samples would produce right now. */
int pfb_frames_for(const pfb_handle* h, uint64_t num_samples, uint64_t* frames_out);

/* Channelize one .iq record straight from disk: parse the header (pfb_iq_parse_header, all three
 * formats), check that the payload matches it (the reference asserts length(iq)==numSamples,
 * matlab/convert_my_iq_to_mat.m:102) and that the handle was created for its sample format and
 * bit width, then stream the payload through the GPU in chunks without holding the file in
 * memory.  Replaces convert_my_iq_to_mat.m:40-118 + the load/normalise lines of the channelizer
 * scripts for the common case.  `out` is host memory; a channel-major handle fills one M x frames
 * matrix for the whole record (column stride = the record's frame count).  The handle's state
 * carries over exactly as with pfb_process (call pfb_reset first for a fresh channelizer per file,
 * create_pdws_channelized.m:33). */
int pfb_process_iq_file(pfb_handle* h, const char* path, void* out, uint64_t out_capacity_frames,
                        uint64_t* frames_out, pfb_iq_info* info_out);

/* ---- state (checkpoint/resume, and the multi-GPU halo) --------------------- */
/* History is the raw input samples (cfg.sample_format) that precede the next
 * call: pfb_history_samples() of them.  A time shard on GPU g>0 is primed with
 * the last pfb_history_samples() samples of shard g-1. */
uint64_t pfb_history_samples(const pfb_handle* h);
/* Feed samples into the history without producing output (n may be any size;
 * only the trailing pfb_history_samples() matter).  Carried-tail phase is
 * advanced by n exactly as pfb_process would. */
int pfb_prime(pfb_handle* h, const void* iq, uint64_t num_samples, uint32_t mem);
/* Opaque state blob: history + counters.  Query size with buf == NULL. */
int pfb_get_state(pfb_handle* h, void* buf, size_t* bytes);
int pfb_set_state(pfb_handle* h, const void* buf, size_t bytes);
/* Global index of the next frame (used by PFB_FLAG_DEROTATE); settable so a
 * time shard can start mid-stream. */
int pfb_set_frame_index(pfb_handle* h, uint64_t next_frame);
int pfb_get_frame_index(const pfb_handle* h, uint64_t* next_frame);

/* ---- time-sharded streams: one segment per GPU ------------------------
 * The reference channelizes a whole record in one call (matlab/create_pdws_channelized.m:57); a host
 * that cuts the stream into G contiguous segments, one per GPU, needs exactly one exchange: the last
 * pfb_halo_samples() RAW samples of segment g are the filter state of segment g+1.  The library
 * orchestrates the step and leaves the transport to the caller, so it links no communication library
 * itself: the callback is where a C++ host puts
 *     ncclGroupStart(); ncclSend(d_send, bytes, ncclUint8, send_to, comm, stream);
 *                       ncclRecv(d_recv, bytes, ncclUint8, recv_from, comm, stream); ncclGroupEnd();
 * (examples/sharded_rccl.cpp; sdr_channelizer_amd/sharded.py does the same through torch.distributed).
 * It must ENQUEUE both transfers on `hip_stream` and return without waiting; either side may be absent
 * (rank < 0 and pointer NULL: the first shard of a non-ring stream receives nothing, the last sends
 * nothing).  Return 0 on success; anything else makes the call fail with PFB_ERR_COMM. */
typedef int (*pfb_halo_exchange_fn)(void* user, const void* d_send, void* d_recv, size_t bytes,
                                    int send_to_rank, int recv_from_rank, void* hip_stream);
typedef struct pfb_shard_config {
  uint32_t struct_size;          /* = sizeof(pfb_shard_config)                                    */
  int32_t rank, world;           /* this handle owns segment `rank` of `world`                    */
  uint32_t ring;                 /* 1: rank 0 receives from rank world-1 (the segments of call i+1 */
                                 /* follow those of call i: an endless stream, what bench.py times).  */
                                 /* The last rank then sends its CARRIED state -- the tail of its    */
                                 /* previous segment, zeros after pfb_reset -- not the tail of the   */
                                 /* segment in hand: its send of call i is what rank 0's receive of  */
                                 /* call i is matched with, and rank 0 is one call ahead of it.      */
                                 /* 0: rank 0 continues from the handle's own state                */
  pfb_halo_exchange_fn exchange; /* may be NULL when world == 1                                   */
  void* user;
} pfb_shard_config;
/* Make the handle one shard of a time-sharded stream (world == 1 detaches). */
int pfb_shard_attach(pfb_handle* h, const pfb_shard_config* cfg);
/* Raw samples a shard needs from its predecessor: M*P - 1 - input_offset, i.e. (P-1)*M with the
 * default input_offset D-1 at D = M -- the (taps_per_branch-1)*M overlap and nothing more. */
uint64_t pfb_halo_samples(const pfb_handle* h);
/* The first frames of a segment whose windows reach into the halo: ceil(pfb_history_samples / D). */
uint64_t pfb_shard_head_frames(const pfb_handle* h);
/* Where the predecessor's tail lands (device memory owned by the handle, pfb_halo_samples() samples
 * of cfg.sample_format): what the library passes to the callback as d_recv. */
void* pfb_halo_recv_buffer(pfb_handle* h);
/* Channelize this rank's segment (device pointers; num_samples a multiple of D and at least
 * pfb_history_samples(), carried tail empty).  Enqueued, no host sync:
 *   side stream : the halo exchange (the callback), ordered behind what the handle's stream has queued
 *   main stream : frames [head, F) -- every frame whose window lies inside the segment -- start at
 *                 once; frames [0, head) follow when the halo has landed (one event wait on the GPU).
 * The result is bit-identical to channelizing the whole stream on one device.  The handle's state
 * afterwards is the segment's tail, as after pfb_process.  pfb_sync() also covers the transfers. */
int pfb_process_shard_async(pfb_handle* h, const void* d_segment, uint64_t num_samples, void* d_out,
                            uint64_t out_capacity_frames, uint64_t* frames_out);

/* ---- helpers -------------------------------------------------------------- */
/* centerFrequencies(channelizer, fs).  What order MathWorks' function returns is NOT pinned here
 * (closed toolbox).  matlab/channelizer_example.m:58-66 plots
 * fftshift(out,2) against fc - centerFrequencies(channelizer,fs), which only reads sensibly if the
 * list is already centred and ascending; matlab/create_pdws_channelized.m:42,80 indexes it with the
 * fftshift-ed column either way.  Both orders are offered:
 *   PFB_FREQ_ORDER_FFT       out[k] = [0,1,..,ceil(M/2)-1,-floor(M/2),..,-1]*fs/M  (column k of `out`)
 *   PFB_FREQ_ORDER_CENTERED  out[c] = (c - floor(M/2))*fs/M                        (column c of fftshift(out,2))
 * pfb_center_frequencies is the FFT order (the frequency of UNSHIFTED output column k). */
enum { PFB_FREQ_ORDER_FFT = 0, PFB_FREQ_ORDER_CENTERED = 1 };
int pfb_center_frequencies(uint32_t num_channels, double fs, double* out);
int pfb_center_frequencies_ordered(uint32_t num_channels, double fs, uint32_t order, double* out);
/* Convenience prototype: Kaiser-windowed sinc, M*P taps, cutoff fs/(2M).
 * NOT verified against MathWorks' internal design -- pass your own taps for
 * parity work. */
int pfb_design_prototype(uint32_t num_channels, uint32_t taps_per_channel,
                         double stopband_atten_db, float* taps_out);
const char* pfb_strerror(int status);
/* Text of the most recent HIP failure on this thread ("" if none). */
const char* pfb_last_error_detail(void);
int pfb_abi_version(void);
int pfb_device_count(void);

/* ---- tuning / introspection (stable names, values may grow) ----------------
 * Production knobs only.  The measurement entry points (copy-kernel yardsticks, the exception-guard self test) and the
 * kernel-study options (PFB_OPT_GRID / TILE_WAVES / EXPERIMENT / VARIANT) live in pfb_channelizer_dev.h: same library,
 * same pfb_set_option, not part of what an integrator binds. */
typedef enum pfb_option {
  PFB_OPT_KERNEL = 0,           /* 0 = auto, 1 = force generic kernel, 2 = require fast kernel */
  PFB_OPT_FRAMES_PER_BLOCK = 1, /* run length per workgroup for the fast kernels (0 = default) */
  PFB_OPT_HOST_CHUNK_SAMPLES = 2, /* staging chunk for PFB_MEM_HOST (0 = default)              */
  PFB_OPT_NONTEMPORAL = 3,      /* 1 = nontemporal output stores                               */
  PFB_OPT_PROFILE = 4,          /* 1 = bracket every channelizer kernel launch with HIP events  */
  PFB_OPT_XCD_REMAP = 5,        /* -1 (default) = per schedule, 1 = consecutive runs stay on one XCD, */
                                /* G > 1 = in groups of G workgroups (schedule 3), 0 = off       */
  PFB_OPT_SCHEDULE = 6,         /* fast kernels: -1 (default) = best measured for the kernel; every schedule of a shape gives */
                                /* the same bits.  Frame-major: 0 = one sliding-window run per workgroup, 2 = one chunk per   */
                                /* wave with adjacent chunks grouped into workgroups, 3 = short sliding runs whose halo rows  */
                                /* are shared through LDS, 4 = 3 with the FIR and the FFT on different waves (M = 64), 6 = a  */
                                /* FIR team + an FFT team in one workgroup (M = 1024 / 560), 7 = a FIR wave + an FFT wave per */
                                /* long sliding run, 11 = 0 software-pipelined inside the wave (M = 128 D = 64), 13 = several */
                                /* independent few-wave workgroups per CU (a variant of M = 1024).  Channel-major handles:    */
                                /* 0 / 2 as above, 8 = short runs whose output is transposed in LDS, 9 = frame-major slabs +  */
                                /* a transpose kernel (the default of the M = 1024 / 560 plans).  1, 5, 10, 12: removed.      */
  /* 7 ... 10: pfb_channelizer_dev.h */
  PFB_OPT_SLAB_FRAMES = 11      /* channel-major by slabs (schedule 9): frames per slab, 0 = one run per CU */
} pfb_option;
int pfb_set_option(pfb_handle* h, int option, int64_t value);
/* With PFB_OPT_PROFILE on: durations (ms) of the channelizer kernel launches
 * since the option was set or since the last call of this function, measured
 * by hipEvents recorded on the handle's stream immediately around each launch
 * (oldest first, at most 4096 kept).  Synchronises the stream.  *count receives
 * the number written (<= capacity). */
int pfb_get_kernel_times(pfb_handle* h, float* ms_out, int capacity, int* count);
/* Name of the kernel the last process call launched ("" before the first). */
const char* pfb_last_kernel(const pfb_handle* h);
/* The device the handle lives on (pfb_config.device_id resolved: -1 became the device current at pfb_create). */
int pfb_get_device(const pfb_handle* h, int* device_id);

/* Page-locked host memory for the sample and output buffers of PFB_MEM_HOST calls -- what the
 * recorders would use in place of `new std::complex<std::int16_t>[n]`
 * (cpp/blade_record_iq_12bit.cpp:268).  The host path copies chunk i+1 in while chunk i is
 * transformed and chunk i-1 is copied out; only page-locked buffers let the two PCIe directions
 * really overlap (pageable memory works, at the rate of the runtime's own staging).  NULL on failure. */
void* pfb_host_alloc(size_t bytes);
void pfb_host_free(void* p);

/* ---- channelized PDW extraction ------------------------------------------------
 * Replaces the second half of matlab/create_pdws_channelized.m (lines 64-143): per-channel
 * noise floor = median magnitude (:73), threshold NF*10^(SNR_THRESHOLD/10) (:74-75), the
 * leading/trailing-edge state machine over frames (:85-135), and per pulse the median
 * magnitude (:101), SNR (:105), pulse width (:110), median wrapped phase step -> frequency
 * (:114-122) and the saturation flag (:130-132).  Input is the F x M matrix the channelizer
 * produced (frame-major complex64, already fftshift-ed as in :60), on the GPU or the host.
 * PDWs come back in the reference's order: channels outermost, time within a channel. */
typedef struct pfb_pdw {
  double toa;   /* UTC seconds: (1-based frame index)/fs + sample_start_time (:98)        */
  double freq;  /* Hz (:122)                                                             */
  double pw;    /* seconds (:110)                                                        */
  double snr;   /* dB (:105)                                                             */
  int32_t sat;  /* 0/1 (:130-132)                                                        */
  int32_t bin;  /* 0-based (shifted) column the pulse was found in                       */
  double mag;   /* median magnitude over the pulse: thisAmp (:101; the channelized script
                   does not store it) / pdw.mag of matlab/create_pdws.m:70,96            */
} pfb_pdw;

enum {
  /* reproduce the reference's indexing quirk at :114: phase(toa:jj) linear-indexes COLUMN 1 of the
   * phase matrix whatever channel the pulse is in.  Without the flag the pulse's own column is used. */
  PFB_PDW_MATLAB_QUIRKS = 1u << 0,
  /* `y` is channel-major, y[k*frames + m]: MATLAB's own column-major F x M matrix (what pfb_process
   * writes for a PFB_LAYOUT_CHANNEL_MAJOR handle).  Transposed once into device scratch. */
  PFB_PDW_CHANNEL_MAJOR = 1u << 1,
  /* fc_chan = fc + binFreqs(bin) (:80) with binFreqs in FFT order (PFB_FREQ_ORDER_FFT) indexed by the
   * fftshift-ed column -- what the script computes IF MathWorks' centerFrequencies returns the unshifted
   * list; every pulse's freq is then half the band away from its channel.  Unpinned (see
   * pfb_center_frequencies): the default is the column's true centre frequency, which is also what the
   * script computes if centerFrequencies returns the centred list. */
  PFB_PDW_BINFREQ_UNSHIFTED = 1u << 2
};

/* fs_in: sample rate BEFORE decimation; the frame rate used is fs_in/decimation (:62).
 * noise_floor_out (optional, M doubles): the per-channel medians.  *count receives the number
 * of pulses found even when it exceeds capacity (only `capacity` are written).
 * mem: PFB_MEM_HOST / PFB_MEM_DEVICE for `y`; `out` and `noise_floor_out` are host memory. */
int pfb_pdw_extract(const void* y, uint64_t frames, uint32_t num_channels, uint32_t decimation,
                    double fs_in, double fc, double sample_start_time, double snr_threshold_db,
                    uint32_t flags, pfb_pdw* out, uint64_t capacity, uint64_t* count,
                    double* noise_floor_out, uint32_t mem, int32_t device_id, void* hip_stream);
/* ---- raw (un-channelized) PDW extraction -----------------------------------------
 * Replaces matlab/create_pdws.m:30-105: the same extraction on the recorder stream itself.
 * x = (I + jQ) / 2^(bit_width-1) (:30-33); ONE noise floor = median |x| (:44); a pulse starts at
 * |x| >= NF*10^(snr_threshold_db/10) (:45-46,57) and ends at |x| <= NF*10^(trailing_threshold_db/10)
 * (:47,63) -- the script uses 18 and 3 dB; trailing must not exceed leading.  Per pulse: toa (:67),
 * mag (:70), snr (:74), pw (:79), freq = fc + fs/(360/median wrapped phase step) (:83-91), sat
 * (:100-102); bin is 0.  `iq` is interleaved I,Q in `sample_format` (PFB_FMT_*), host or device per
 * `mem`; fs is the sample rate; noise_floor_out (optional) receives the one median.  *count receives
 * the number of pulses found even when it exceeds capacity. */
int pfb_pdw_extract_raw(const void* iq, uint64_t num_samples, uint32_t sample_format, uint32_t bit_width,
                        double fs, double fc, double sample_start_time, double snr_threshold_db,
                        double trailing_threshold_db, pfb_pdw* out, uint64_t capacity, uint64_t* count,
                        double* noise_floor_out, uint32_t mem, int32_t device_id, void* hip_stream);
/* One iteration of the loop in matlab/create_pdws_channelized.m:22-143 -- load a record, channelize
 * it, fftshift (the handle's PFB_FLAG_FFTSHIFT), extract its PDWs -- in one call: the payload is
 * streamed to the GPU as in pfb_process_iq_file, the F x M channel matrix stays in device memory
 * (scratch kept on the handle, freed with it) and only the PDWs come back.  fs, fc and the sample
 * start time are the record's (sampleRateSps, frequencyHz, sampleStartTime: the values
 * convert_my_iq_to_mat.m stores in the .mat the script loads).  The handle must be frame-major with
 * complex output and match the record's format and bit width; its state carries over as with
 * pfb_process (pfb_reset first for a fresh channelizer per file, :33).  Remaining arguments as in
 * pfb_pdw_extract. */
int pfb_pdw_from_iq_file(pfb_handle* h, const char* path, double snr_threshold_db, uint32_t flags,
                         pfb_pdw* out, uint64_t capacity, uint64_t* count, double* noise_floor_out,
                         pfb_iq_info* info_out);
/* One iteration of the loop in matlab/create_pdws.m (load a record, :30-105 on its raw stream):
 * the payload goes to the device through page-locked chunks, pfb_pdw_extract_raw runs there, the PDWs
 * come back.  fs, fc, bit width and start time are the record's. */
int pfb_pdw_raw_from_iq_file(const char* path, double snr_threshold_db, double trailing_threshold_db,
                             pfb_pdw* out, uint64_t capacity, uint64_t* count, double* noise_floor_out,
                             pfb_iq_info* info_out, int32_t device_id);
/* Text of the most recent HIP failure inside pfb_pdw_extract on this thread. */
const char* pfb_pdw_last_error_detail(void);
/* How the last pfb_pdw_extract on this thread found the noise floors: 1 = sampled bracket + one
 * pass that also produced the edge masks, 4 = the same but the masks needed their own pass (too many
 * samples near the threshold, or a median at the very edge of its bracket), 2 = full radix select
 * (short input), 3 = the bracket check failed (heavily tied data) and the full radix select ran
 * after it.  All give the exact medians and edges; diagnostic only. */
int pfb_pdw_last_noise_floor_path(void);
/* pfb_pdw_extract keeps its device scratch between calls (grow-only, per device; calls are
 * serialised on it).  This frees it: device_id >= 0 for one device, < 0 for all. */
int pfb_pdw_release_workspace(int32_t device_id);

#ifdef __cplusplus
}
#endif
#endif
