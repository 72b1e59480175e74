/*
 * specscan_channelizer.h — C ABI of the recorder channeliser, part of libspecscan.so.
 *
 * The reference records every detected transmission with a per-slot GNU Radio chain
 * (Recorder, sources/radio/recorder.cpp:14-46; `recordersCount` slots per device, sources/radio/sdr_device.cpp:39-41):
 *
 *   source -> Blocker(drop while idle) -> rotator_cc(-2*pi*shift/fs) -> rational_resampler<cc>(f1, f2) x stages
 *          -> complex_to_interleaved_char(vector, 127.0) -> stream_to_vector -> Buffer -> DataController::pushTransmission
 *
 * with the stage factors from getResamplersFactors(fs, recording bandwidth, RESAMPLER_THRESHOLD = 125)
 * (sources/utils/radio_utils.cpp:128-152, sources/config.h:20) and GNU Radio's default resampler taps.
 * One sc_ctx replaces the rotator -> resamplers -> int8 conversion of ALL slots of one device: every call takes
 * the device's IQ stream once and produces, per recording slot, the int8 samples the reference hands to its Buffer.
 * Slot bookkeeping (which shift is recorded where, flush intervals, MQTT framing) stays on the host
 * (sdr_device.cpp:82-144, recorder.cpp:58-98, network/data_controller.cpp:27-42).
 *
 * State carried between calls, exactly as the reference's blocks carry it: per slot the rotator phase and every
 * resampler's history and polyphase counter; an idle slot sees no samples (Blocker drops them) and keeps its state.
 * No CPU fallback: without a GPU sc_create fails.
 */
#ifndef SPECSCAN_CHANNELIZER_H
#define SPECSCAN_CHANNELIZER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SC_ABI_VERSION 1u
#define SC_MAX_CHANNELS 16
#define SC_MAX_STAGES 8

typedef struct sc_ctx sc_ctx;

typedef struct sc_config {
  uint32_t abi_version; /* SC_ABI_VERSION */
  int32_t sample_rate;  /* Recorder::m_sampleRate (recorder.cpp:16) */
  int32_t bandwidth;    /* Config::recordingBandwidth(): recording.min_sample_rate (config.cpp:79,140) */
  int32_t threshold;    /* RESAMPLER_THRESHOLD (config.h:20) */
  int32_t channels;     /* recordersCount (sdr_device.cpp:39): 1..SC_MAX_CHANNELS */
  int32_t max_samples;  /* largest nsamples of one call */
  float pack_scale;     /* complex_to_interleaved_char scale, 127.0 (recorder.cpp:36) */
  int32_t device_id;
} sc_config;

/* status codes are the ss_status values of specscan.h (0 = ok, < 0 = error) */
void sc_default_config(sc_config* cfg, int32_t sample_rate, int32_t bandwidth);
int sc_create(const sc_config* cfg, sc_ctx** out);
void sc_destroy(sc_ctx* ctx);
const char* sc_last_error(const sc_ctx* ctx); /* ctx may be NULL: last sc_create failure of this thread */

/* The resampler cascade this context runs (getResamplersFactors + rational_resampler's default taps). */
int sc_stage_count(const sc_ctx* ctx);
int sc_stage_info(const sc_ctx* ctx, int32_t stage, int32_t* interpolation, int32_t* decimation, int32_t* ntaps);
int sc_stage_taps(const sc_ctx* ctx, int32_t stage, float* taps /* ntaps */);
/* Upper bound of output samples per channel for a call of nsamples input samples. */
int32_t sc_output_capacity(const sc_ctx* ctx, int32_t nsamples);

/* Recorder::startRecording (recorder.cpp:58-73): rotator increment 2*pi*(-shift/fs), slot unblocked. Phase and filter
 * histories are NOT reset, as in the reference. Recorder::stopRecording (:75-87): slot blocked. */
int sc_start(sc_ctx* ctx, int32_t channel, int32_t shift_hz);
int sc_stop(sc_ctx* ctx, int32_t channel);
int sc_is_recording(const sc_ctx* ctx, int32_t channel);

/* One work() pass over nsamples CF32 samples (interleaved re,im) of the device stream.
 *   out_i8   [channels][cap][2] int8 (re,im) — what complex_to_interleaved_char emits;   nullable
 *   out_cf32 [channels][cap][2] float — the last resampler's output (DEBUG_SAVE_RECORDING_RAW_IQ tap, recorder.cpp:42-45); nullable
 *   counts   [channels] samples produced per channel (0 for idle slots)
 * sc_process takes host pointers and is synchronous; sc_process_device takes device pointers, is asynchronous on the
 * context's stream (sc_sync), and returns counts (host array) immediately — they do not depend on the data. */
int sc_process(sc_ctx* ctx, const void* iq, int32_t nsamples, int8_t* out_i8, float* out_cf32, int32_t* counts, int32_t cap);
int sc_process_device(sc_ctx* ctx, const void* d_iq, int32_t nsamples, int8_t* d_out_i8, float* d_out_cf32, int32_t* counts, int32_t cap);
int sc_sync(sc_ctx* ctx);

/* DataController::pushTransmission payload (data_controller.cpp:27-42): uint64 time ms, int32 start, int32 stop,
 * uint32 sample rate, then the samples as offset-binary bytes (int8 ^ 0x80). Host-side helper; returns the byte count
 * (out may be NULL to query it), or < 0 if cap is too small. */
int sc_transmission_payload(uint64_t time_ms, int32_t frequency, int32_t sample_rate, const int8_t* iq_i8, int32_t nsamples, uint8_t* out,
                            int32_t cap);

#ifdef __cplusplus
}
#endif
#endif /* SPECSCAN_CHANNELIZER_H */
