/*
 * specscan.h — C ABI of the MI355X spectral-scan engine (libspecscan.so).
 *
 * This is the drop-in boundary for ONE path of shajen/rtl-sdr-scanner-cpp: the chain of GNU Radio
 * blocks that `SdrDevice::setupChains` wires right after the Blocker
 * (reference sources/radio/sdr_device.cpp:161-168):
 *
 *   Decimator<gr_complex>   sources/radio/blocks/decimator.h:11-22    keep first N of each N*D item
 *   fft_v<gr_complex,true>  sources/radio/sdr_device.cpp:164          Hamming window, forward FFT, shift
 *   PSD                     sources/radio/blocks/psd.cpp:11-22        10*log10(|X|^2 / fs)
 *   NoiseLearner            sources/radio/blocks/noise_learner.cpp:11-67   learn max, then subtract
 *   Transmission (front)    sources/radio/blocks/transmission.cpp:57-61,88-96
 *       Averager::push      sources/radio/averager.cpp:14-25,52-61    mean of the last 21 frames
 *       average()           sources/utils/utils.cpp:31-53             centred 21-bin mean
 *       threshold           sources/radio/blocks/transmission.cpp:90-94   startLevel, range, ignored
 *
 * Every block above is a gr::sync_block whose only entry point is
 *   int work(int noutput_items, gr_vector_const_void_star& in, gr_vector_void_star& out)
 * (sources/radio/blocks/psd.h:11). ss_process() is that call for the fused chain: `nframes` input items
 * of N*D complex samples in, per-bin planes and per-frame candidate lists out. The control entry
 * points mirror the calls the Scanner thread makes into the blocks.
 *
 * No exceptions cross this ABI. Every function returns 0 (SS_OK) or a negative ss_status;
 * ss_last_error() gives the message for the last failure on that context.
 *
 * The CPU oracle (oracle/specscan_oracle.h) exports the same set with the prefix orc_.
 */
#ifndef SPECSCAN_H
#define SPECSCAN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SS_ABI_VERSION 3 /* 3: ss_get_stats, ss_input_wait, SS_FLAG_REFERENCE_NAN; 2: ss_flush, SS_FLAG_NO_CULL / SS_FLAG_STREAM_ORDERED, buffer lifetime of ss_process_device */

typedef enum ss_status {
  SS_OK = 0,
  SS_ERR_INVALID = -1,       /* bad argument / bad config field */
  SS_ERR_NO_DEVICE = -2,     /* no HIP device, or device_id out of range */
  SS_ERR_HIP = -3,           /* a HIP runtime call failed (message has the HIP error string) */
  SS_ERR_BATCH = -4,         /* nframes > max_batch */
  SS_ERR_CAND_OVERFLOW = -5, /* more candidates than cand_cap; cand_off is still exact */
  SS_ERR_NOMEM = -6
} ss_status;

/* Sample format of the IQ stream handed to ss_process. The reference always asks SoapySDR for CF32
 * (sources/radio/blocks/sdr_source.cpp:52,75); CS8/CU8 are what RTL-SDR/HackRF produce natively, and
 * the engine converts them in the FFT kernel's load stage: cf32 = (int8 - offset) * int_scale. */
typedef enum ss_format {
  SS_FMT_CF32 = 0, /* interleaved float re,im  (gr_complex)            8 B/sample */
  SS_FMT_CS8 = 1,  /* interleaved int8  re,im  (HackRF)                2 B/sample */
  SS_FMT_CU8 = 2   /* interleaved uint8 re,im, offset 127.5 (RTL-SDR)  2 B/sample */
} ss_format;

/* Which per-bin plane ss_read_window returns. */
typedef enum ss_plane {
  SS_PLANE_PSD = 0, /* PSD::work output, dB                                      */
  SS_PLANE_REL = 1, /* NoiseLearner::work output (rawPower in Transmission)      */
  SS_PLANE_AVG = 2  /* average(Averager.average()) (avgPower in Transmission)    */
} ss_plane;

/* ss_config.flags: keep the full avg plane of every batch on the device so that ss_read_window can serve
 * SS_PLANE_AVG (the host-side signal tracker needs it); costs 4 B/sample of extra HBM writes. */
#define SS_FLAG_KEEP_PLANES 1u
/* Also run the Spectrogram side branch (sources/radio/blocks/spectrogram.cpp): accumulate the bin-decimated raw
 * PSD per centre frequency; read it back with ss_spectrogram_read. */
#define SS_FLAG_SPECTROGRAM 2u
/* 8192-, 65536-, 131072- (int8), 262144- and 2^20-point frames: evaluate every averaging tile, also those whose per-frame maxima show that no window
 * mean of the tile can reach start_level (csrc/detect_fused.h, tile culling). Results are identical either way; the flag
 * exists so that the data-independent cost of the chain can be measured (bench.py reports both). */
#define SS_FLAG_NO_CULL 4u
/* ss_process_device keeps to the context's stream: every stage of a call — FFT + dB, averaging / threshold, candidate lists — is
 * enqueued on ss_stream, in order, before the call returns, and work the caller enqueues there afterwards (a producer refilling
 * d_iq, a consumer of the planes or the lists) is ordered behind it: the classic stream contract, no ss_flush needed, at about
 * half the throughput of the default, in which consecutive calls overlap on queues of the library's own and every buffer of a
 * call must stay untouched until ss_sync (see ss_process_device). tests/test_gpu_stream_ordered.py. */
#define SS_FLAG_STREAM_ORDERED 8u
/* Reproduce what the reference does after a degenerate frame instead of recovering from it. A dB value that is not finite
 * (-inf: a bin of exact zeros, psd.cpp:19; NaN: a NaN sample) poisons the reference's running sums for good: the Averager's per-bin
 * sum turns NaN when the row leaves its 21-frame window (inf - inf, averager.cpp:40-50) — at once for a NaN — and average()'s one
 * running sum per row carries the first such bin along the rest of the row (utils.cpp:39-48), so from then on every bin from ten
 * below the first poisoned one upwards is NaN and detects nothing until Transmission::resetBuffers (transmission.cpp:42-55, here
 * ss_reset). With this flag the library does the same: identical candidate lists and the same NaN / -inf pattern in the avg plane
 * to the end of the stream (tests/test_gpu_degenerate_input.py). Without it (the default) the library is blind exactly while the
 * degenerate row is inside the window and detects again at most 15 frames later. Needs the 21 x 21 grouping; calls run their
 * stages in order on ss_stream (as with SS_FLAG_STREAM_ORDERED). */
#define SS_FLAG_REFERENCE_NAN 16u

#define SS_NO_DATA (-100.0f) /* setNoData sentinel, sources/utils/radio_utils.cpp:72-76 */

typedef struct ss_config {
  int32_t abi_version; /* SS_ABI_VERSION */
  int32_t fft_size;    /* N, power of two, 64 .. 2^20. Reference rule: getFft(fs, 250), radio_utils.cpp:98-104 */
  int32_t sample_rate; /* fs in Hz (Frequency = int32_t, help_structures.h:13)                 */
  int32_t decim;       /* D >= 1: an input item is N*D samples and only the first N are used
                          (sdr_device.cpp:152, decimator.h:15-22)                              */
  int32_t in_format;   /* ss_format */
  float int_scale;     /* CS8/CU8 -> float scale; 0 selects 1/128 (CS8) or 1/127.5 (CU8)       */
  const float* window; /* N taps, or NULL for gr::fft::window::hamming(N) (sdr_device.cpp:164)  */
  int32_t grouping_x;  /* bins averaged in frequency, odd; GROUPING_X = 21 (config.h:28)        */
  int32_t grouping_y;  /* frames averaged in time;       GROUPING_Y = 21 (config.h:29)          */
  float start_level;   /* Device::m_startLevel, dB over the learned ceiling (config.h:30)       */
  int32_t range_lo;    /* scanned range in Hz; centre = (lo+hi)/2 (sdr_device.cpp:66,146)       */
  int32_t range_hi;
  int32_t n_ignored;       /* Config::ignoredRanges(): n pairs lo,hi in Hz                      */
  const int32_t* ignored;  /* 2*n_ignored values, copied at create                              */
  int32_t learn_frames;    /* frames absorbed by the noise ceiling per centre frequency when no
                              timestamps are given (2 s * 50 fps = 100 in the reference's regime) */
  int32_t learn_ms;        /* NOISE_LEARNING_TIME (config.h:24) used when timestamps are given  */
  int32_t max_batch;       /* largest nframes a single ss_process call may carry                */
  int32_t device_id;       /* HIP device ordinal                                                */
  uint32_t flags;          /* SS_FLAG_* bits                                                    */
} ss_config;

typedef struct ss_ctx ss_ctx;

/* Fill `cfg` with the reference's compile-time constants (config.h:24-33) for sample rate `fs`:
 * N = getFft(fs, 250), D = max(1, int(fs/N/50)), grouping 21x21, start level 8 dB, learn 2000 ms /
 * 100 frames, Hamming window, CF32 input, range = centre +- fs/2. */
void ss_default_config(ss_config* cfg, int32_t sample_rate, int32_t center_hz);

int ss_device_count(void);

/* Replaces the construction of Decimator/fft_v/PSD/NoiseLearner/Transmission in
 * SdrDevice::setupChains (sdr_device.cpp:148-168). */
int ss_create(const ss_config* cfg, ss_ctx** out);
void ss_destroy(ss_ctx* ctx);
const char* ss_last_error(const ss_ctx* ctx); /* ctx may be NULL: message of the last failed ss_create */

/* One work() call of the fused chain on HOST buffers (copied in and out before returning; buffers
 * stay owned by the caller, as GNU Radio's scheduler owns them in the reference).
 *   iq        nframes items of N*D samples in cfg.in_format
 *   t_ms      per-frame wall-clock in ms (what getTime() returned in the reference, utils.cpp:14), or
 *             NULL for frame-count noise learning
 *   psd_db, rel_db, avg_db   nframes*N floats each, any may be NULL
 *   cand_off  nframes+1 offsets into cand_idx (CSR); may be NULL when cand_cap == 0
 *   cand_idx  bins i with start_level <= avg[i] && inRange(i) && !ignored(i), ascending per frame
 *             (transmission.cpp:90-94, before the sort at :95)
 *   cand_avg  avg[i] for each candidate (the sort key of transmission.cpp:95), may be NULL
 * Returns SS_OK or a negative status. */
int ss_process(ss_ctx* ctx, const void* iq, int32_t nframes, const int64_t* t_ms,
               float* psd_db, float* rel_db, float* avg_db,
               int32_t* cand_off, int32_t* cand_idx, float* cand_avg, int32_t cand_cap);

/* Degenerate input. A frame of exact zeros gives -inf in every PSD bin, as PSD::work does (log10f(0), psd.cpp:19). The
 * reference then never detects again until the next resetBuffers: -inf - (-inf) = NaN when the row leaves the Averager's
 * window (averager.cpp:40-50) and average() carries NaN along each row (utils.cpp:39-48). This library is blind exactly
 * while the -inf row is inside the 21-frame window — like the reference — and detects again at most 15 frames later (its
 * sliding sums restart every 16 frames and every 16 bins). Pinned, with the frames where the two part, by
 * tests/test_gpu_degenerate_input.py. */

/* Same call on DEVICE buffers (hipMalloc'd on cfg.device_id), enqueued on the context's stream and
 * NOT synchronised: call ss_sync before reading results. n_learn = how many leading frames of this
 * batch belong to the noise-learning phase is decided on the host from learn_frames.
 *
 * Stage pipelining (8192-point frames and 16384 points upwards, 21 x 21 grouping): like the reference's flowgraph, whose
 * blocks each work on a different frame at any moment (sdr_device.cpp:161-171), consecutive calls overlap on the device —
 * a launch carries the FFT + dB stage of one call and the averaging / threshold and candidate-list stages of earlier calls
 * (csrc/scan_step.h); for 8192-point frames up to five calls are in flight, on two hardware queues of the library's own
 * (ss_ctx::deep in csrc/specscan.hip). 65536-point frames (and, since ABI 3 / round 6, 262144-point frames — the size getFft picks at
 * 61.44 MS/s —, which take the same pipeline): the launch of call k carries the plan of call k - 1 (which of its
 * averaging tiles can hold a candidate), the averaging / threshold stage of call k - 2 and the candidate lists of call k - 3 — the
 * lists of a call exist three calls later, or after ss_flush / ss_sync —, mask / counter / list sets rotate over four to six, and
 * the averager ring's buffer holds four batches (csrc/ring_place.h). 2^20-point frames: two launches per call, the stages of
 * the two calls before riding on them. The results of a call are therefore complete only after ss_sync, or after ss_flush
 * followed by any synchronisation of ss_stream; every buffer passed to a call (d_iq included) must stay valid and
 * untouched until then — d_iq in particular: the next call's launch reads the call's last frames once more. A streaming
 * producer that cannot afford ss_sync learns when an input buffer is dead from ss_input_wait (below). (As a courtesy, a caller
 * seen handing in frames where those of one of the last five calls lie is taken off the overlapped path until ss_reset: its
 * calls then run their stages in order on ss_stream, ss_get_stats says so; SS_FLAG_STREAM_ORDERED asks for that from the
 * start. The check sees addresses, not contents, and only the latest calls: it is no substitute for ss_input_wait.) Work the
 * caller has enqueued on ss_stream before a call (a producer of d_iq) is waited for.
 * Handing a plane or candidate buffer to a later call again without ss_sync in between is safe — the library orders the
 * stages that touch it, draining its pipeline first where it has to — and costs nothing when the output sets rotate with
 * an even period of at least six calls (four for the PSD / rel planes alone); of course only the newest contents can be
 * read afterwards. The call itself only enqueues; with SS_FLAG_SPECTROGRAM it waits once the host is 64 calls ahead of the
 * device. Results are bit-identical to running the three stages back to back. The host-buffer entry points
 * (ss_process, ss_feed_*) and every call that reads or changes state (ss_set_frequency_range, ss_reset, ss_reset_noise,
 * ss_read_window, ss_read_noise, ss_spectrogram_read) drain the deferred stages themselves. */
int ss_process_device(ss_ctx* ctx, const void* d_iq, int32_t nframes,
                      float* d_psd_db, float* d_rel_db, float* d_avg_db,
                      int32_t* d_cand_off, int32_t* d_cand_idx, float* d_cand_avg, int32_t cand_cap);
int ss_flush(ss_ctx* ctx); /* enqueue the deferred stages of earlier ss_process_device calls (asynchronous) */
/* Input lifetime without ss_sync, for a streaming producer that rotates m >= 2 input buffers: makes `stream` (a hipStream_t; NULL =
 * ss_stream) wait until the input frames of every ss_process_device call up to and including the one `calls_back` calls before the
 * latest are dead — read for the last time. calls_back >= 1: the LATEST call's last frames are read once more by the launch of the
 * call that follows it, so they cannot be released before that call exists. A producer refilling buffer k mod m for call k calls
 * ss_input_wait(ctx, its_stream, m - 1) first, then enqueues the refill on its_stream, then (its_stream == ss_stream, or after
 * making ss_stream wait for the refill) calls ss_process_device. From the first use on the library records one event per launch
 * (about 1 us per call). On contexts whose calls run in order on ss_stream (SS_FLAG_STREAM_ORDERED, transforms other than 8192
 * points) the wait is on ss_stream's latest work. */
int ss_input_wait(ss_ctx* ctx, void* stream, int32_t calls_back);
int ss_sync(ss_ctx* ctx);  /* ss_flush + wait for the context's stream */
void* ss_stream(ss_ctx* ctx); /* the hipStream_t ss_process_device enqueues on */

/* What the library did, for callers and benchmarks that want to say so: every counter runs from ss_create. The device-side counters
 * (tiles_*, wait_fallbacks) cover the stages that have finished on the device; call ss_sync first for exact figures. `size` must
 * hold sizeof(ss_stats) as the caller knows it (fields are only ever appended). */
typedef struct ss_stats {
  uint32_t size;
  uint32_t state;             /* SS_STATE_* bits, as of now */
  uint64_t calls;             /* batches processed (ss_process, ss_process_device, ss_feed_submit) */
  uint64_t calls_overlapped;  /* ... whose launch went to the library's own queues, overlapping its neighbours (8192 points) */
  uint64_t calls_in_order;    /* ... whose stages ran in order on ss_stream */
  uint64_t drains;            /* times the deferred stages were drained (ss_sync, ss_flush, reads, retunes, resets, buffer clashes) */
  uint64_t demotions;         /* times a caller refilling an input buffer in flight took the context off the overlapped path */
  uint64_t tiles_total;       /* 16-frame x 256-bin averaging tiles of the batches processed (21 x 21 grouping) */
  uint64_t tiles_tested;      /* ... that went through the culling test (tile culling: 8192, 65536, 131072 (int8), 262144 and 2^20 points, not with SS_FLAG_NO_CULL) */
  uint64_t tiles_culled;      /* ... that the test proved empty and nobody evaluated (tiles evaluated = tiles_total - tiles_culled) */
  uint64_t wait_fallbacks;    /* workgroups that stopped waiting for a launch's tile plan and made it themselves (csrc/detect_fused.h) */
} ss_stats;
#define SS_STATE_CULLING 1u        /* tile culling is on for this context */
#define SS_STATE_OVERLAP 2u        /* consecutive ss_process_device calls may overlap on the library's queues */
#define SS_STATE_DEMOTED 4u        /* ... but this caller was seen refilling an input buffer in flight: in order until ss_reset */
#define SS_STATE_EAGER 8u          /* ... but this caller waits after every call: in order until it stops doing so */
int ss_get_stats(ss_ctx* ctx, ss_stats* out);

/* Measurement aid (bench.py): when enabled (enable = 1: every launch, enable = k > 1: every k-th launch from the k/2-th on, to keep
 * the ~4 us the two event packets cost out of most steps), a launch of the dominant kernel (fused load + window +
 * FFT + dB) carries its own start/stop events on the context's stream; ss_kernel_timing_read
 * synchronises the stream, returns the summed device time in ms and the number of timed launches,
 * and clears the tally. Replaces the reference's PerformanceLogger::kick (sources/performance_logger.cpp:9-22,
 * called from PSD::work, psd.cpp:15-17). */
int ss_kernel_timing(ss_ctx* ctx, int enable);
int ss_kernel_timing_read(ss_ctx* ctx, double* total_ms, int32_t* launches);
/* The same tally kernel by kernel, for the chains that take several launches per call (16384 points and more: the column half of
 * the transform with the deferred detect / emit stages riding on it, the radix-A step of 2^19 / 2^20-point rows, the row half +
 * dB, the tile-culling plan): a sampled call attaches events to every one of its launches. ms_by_slot and launches_by_slot
 * hold SS_KSLOT_COUNT entries each; ss_kernel_timing_read is entry SS_KSLOT_STEP of this. Either call clears the tally. */
enum { SS_KSLOT_STEP = 0, SS_KSLOT_ROWS = 1, SS_KSLOT_SUB = 2, SS_KSLOT_PLAN = 3, SS_KSLOT_COUNT = 4 };
/* (2^20-point frames take two passes since round 4: SS_KSLOT_STEP is the column half there — a launch of its own, the plan of the call
 * before in its first workgroups —, SS_KSLOT_ROWS the row half + dB with the deferred detect / emit stages of earlier calls riding on
 * it. 65536-point frames: SS_KSLOT_STEP the column half with the plan, detect and emit stages of earlier calls, SS_KSLOT_ROWS the row
 * half. SS_KSLOT_PLAN: the plan as a launch of its own — drains only.) */
int ss_kernel_timing_read_slots(ss_ctx* ctx, double* ms_by_slot, int32_t* launches_by_slot);
/* ... and with the frames the timed launches covered, slot by slot (frames_by_slot: SS_KSLOT_COUNT entries): a call the library
 * takes through in chunks — 2^20 points beyond 16 frames, the four-step form of 65536 points beyond 256 — has several launches per
 * slot, each over its chunk's frames; bytes per launch follow from frames / launches, not from the call's size. */
int ss_kernel_timing_read_frames(ss_ctx* ctx, double* ms_by_slot, int32_t* launches_by_slot, int64_t* frames_by_slot);

/* Device self-test of arithmetic shortcuts used by the kernels (which = 0: the 3-instruction division by
 * 21 equals the IEEE division for every float). Returns the number of mismatches (0 = pass) or < 0. */
long long ss_selftest(int device_id, int which);

/* SdrDevice::setFrequencyRange's effect on the chain (sdr_device.cpp:66,77,146): new scanned range,
 * centre = (lo+hi)/2. Noise ceilings are kept per centre frequency (noise_learner.h:33). */
int ss_set_frequency_range(ss_ctx* ctx, int32_t lo_hz, int32_t hi_hz);
/* Transmission::resetBuffers -> Averager::reset (transmission.cpp:42-55, averager.cpp:27-34). */
int ss_reset(ss_ctx* ctx);
/* NoiseLearner::resetBuffers (noise_learner.cpp:69-72): forget every learned ceiling. */
int ss_reset_noise(ss_ctx* ctx);

/* Fetch bins [lo,hi) of one plane for frame `frame` of the LAST processed batch; frame may be
 * negative down to -(grouping_y-1) for SS_PLANE_REL, addressing the averager ring rows that
 * Transmission::getBestIndex walks (transmission.cpp:132-154). SS_PLANE_REL is always there. SS_PLANE_AVG needs
 * SS_FLAG_KEEP_PLANES. SS_PLANE_PSD is there after ss_process and after ss_process_device calls that were given d_psd_db; a
 * 65536-, 262144- or 2^20-point ss_process_device call in detect mode (no plane handed out, no SS_FLAG_KEEP_PLANES) writes no dB
 * plane at all — its rows go straight to the averager ring's buffer (65536 / 131072 points, int8 IQ: as dB values, in the blocked
 * order the fold leaves them in, csrc/fft65536_dif8.h; this call hands noise-relative values back in bin order all the
 * same) — and SS_PLANE_PSD / SS_PLANE_AVG then fail with SS_ERR_INVALID: pass d_psd_db, or SS_FLAG_KEEP_PLANES at ss_create,
 * to a caller that wants them. */
int ss_read_window(ss_ctx* ctx, int32_t plane, int32_t frame, int32_t lo, int32_t hi, float* out);

/* Spectrogram side branch (needs SS_FLAG_SPECTROGRAM). ss_spectrogram_size: number of output bins,
 * min(16384, getFft(fs, 1000)) (spectrogram.cpp:14), 0 when disabled. ss_spectrogram_read: what Spectrogram::send
 * publishes (spectrogram.cpp:62-75, minus its 1000 ms gate and the MQTT framing of data_controller.cpp:44-57):
 * out[j] = int8(sum[j]/count) for the current centre frequency, container cleared; mean_out (nullable) gets the
 * float before the conversion. Returns the number of frames accumulated (0 = nothing to send) or < 0. */
int ss_spectrogram_size(const ss_ctx* ctx);
int ss_spectrogram_read(ss_ctx* ctx, int8_t* out, float* mean_out);

/* DataController::pushSpectrogram payload (sources/network/data_controller.cpp:44-57): uint64 time ms, int32 start
 * (= frequency - sample_rate / 2), int32 stop, int32 step (= sample_rate / size), uint32 size, then the int8 row as it
 * is. Host-side helper for the row ss_spectrogram_read returned; gives the byte count (out may be NULL to query it),
 * or < 0 when size < 1 or cap is too small. */
int ss_spectrogram_payload(uint64_t time_ms, int32_t frequency, int32_t sample_rate, const int8_t* row, int32_t size, uint8_t* out,
                           int32_t cap);

/* Learned ceiling for the current centre frequency: N floats, -FLT_MAX where nothing was learned.
 * Returns 1 if learning is complete, 0 if still learning, <0 on error. */
int ss_read_noise(ss_ctx* ctx, float* thr);

/* ---- pipelined host feeding (replay front end) ------------------------------------------------
 * ss_process is synchronous, as a GNU Radio work() has to be (the scheduler owns the buffers only during the
 * call). A source that owns its buffers — a file being replayed (the reference's own raw dumps,
 * sources/utils/radio_utils.cpp:78-84, sources/radio/sdr_device.cpp:173-181), or a SoapySDR readStream loop
 * (sources/radio/blocks/sdr_source.cpp:63-93) writing straight into pinned memory — can do better: the feed owns
 * `depth` slots of pinned staging; while batch k runs, batch k+1 crosses PCIe on a copy stream and the caller
 * fills k+2. Batches go through the chain strictly in submission order, with the same state as ss_process.
 *
 *   ss_feed_acquire  -> pinned buffer for up to max_batch frames of N samples (in_format), already decimated
 *                       (first N samples of each N*decim item, sources/radio/blocks/decimator.h:15-22)
 *   ss_feed_submit   -> async H2D + the chain + async D2H of the results; returns at once
 *   ss_feed_collect  -> oldest submitted batch; blocks until it is done. Pointers are pinned host memory owned
 *                       by the feed, valid until that slot is handed out again by ss_feed_acquire.
 * Returns SS_ERR_INVALID from acquire when every slot is submitted-and-uncollected (collect first), and from
 * collect when nothing is pending. Not to be mixed with ss_process on the same context while batches are pending. */
typedef struct ss_feed ss_feed;
typedef struct ss_feed_result {
  int32_t nframes;
  int32_t status;          /* SS_OK, or SS_ERR_CAND_OVERFLOW (cand_off exact, lists truncated to cand_cap) */
  int64_t user_tag;        /* the tag given to ss_feed_submit */
  const int32_t* cand_off; /* nframes + 1 */
  const int32_t* cand_idx; /* min(cand_off[nframes], cand_cap) */
  const float* cand_avg;
  const float* psd_db;     /* nframes * N when the feed was created with want_psd, else NULL */
} ss_feed_result;

int ss_feed_create(ss_ctx* ctx, int32_t depth, int32_t cand_cap, int32_t want_psd, ss_feed** out);
void ss_feed_destroy(ss_feed* feed);
int ss_feed_acquire(ss_feed* feed, void** frames);
int ss_feed_submit(ss_feed* feed, int32_t nframes, const int64_t* t_ms, int64_t user_tag);
int ss_feed_collect(ss_feed* feed, ss_feed_result* out);
int ss_feed_pending(const ss_feed* feed);

#ifdef __cplusplus
}
#endif
#endif /* SPECSCAN_H */
