/*
 * channelizer_oracle.h — CPU oracle for the recorder's channeliser.
 *
 * TEST INFRASTRUCTURE ONLY (same rule as specscan_oracle.h): only tests/ may load it.
 *
 * The reference's Recorder (sources/radio/recorder.cpp:14-46) is a GNU Radio chain per recording slot:
 *   source -> Blocker(drop) -> rotator_cc -> rational_resampler<cc,cc,cc>(f1, f2) [x stages] ->
 *   complex_to_interleaved_char(vector=true, 127.0) -> stream_to_vector -> Buffer -> DataController::pushTransmission
 * What is the reference's OWN code and pinned by its own tests:
 *   getResamplersFactors / split / getPrimeFactors  (sources/utils/radio_utils.cpp:9-35,105-152;
 *   known-answer vectors tests/test_radio_utils.cpp:28-69, and oracle/_ref compiles radio_utils.cpp in place),
 *   the phase increment -2*pi*shift/fs (recorder.cpp:64), the wire format (network/data_controller.cpp:27-42).
 * What lives in GNU Radio / VOLK (un-vendored, version unpinned) and is RESTATED here from the published 3.10
 * sources — "PARITY UNPINNED" by any reference test, anchored in tests/ against fp64 scipy:
 *   gr::blocks::rotator (phase recurrence, renormalised every 512 samples and at the end of every call),
 *   rational_resampler's default taps (design_resampler_filter -> firdes::low_pass, Kaiser beta 7, fractional_bw 0.4),
 *   its polyphase general_work, and volk_32f_s32f_convert_8i (scale, saturate, rintf).
 */
#ifndef CHANNELIZER_ORACLE_H
#define CHANNELIZER_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CHO_MAX_STAGES 8

/* getResamplersFactors(sampleRate, bandwidth, threshold): pairs (interpolation, decimation), returns the count */
int cho_resampler_factors(int32_t sample_rate, int32_t bandwidth, int threshold, int* interp, int* decim, int cap);

/* rational_resampler::make(interp, decim) with no taps: gcd-reduced factors, design_resampler_filter(.., 0.4).
 * Returns the tap count (taps may be NULL to query it). */
int cho_design_taps(int interp, int decim, float* taps, int cap);

typedef struct cho_chain cho_chain;

/* one recording slot: rotator + the resampler cascade for (sample_rate -> bandwidth) + int8 conversion */
cho_chain* cho_create(int32_t sample_rate, int32_t bandwidth, int threshold);
void cho_destroy(cho_chain* c);
int cho_stage_count(const cho_chain* c);
void cho_stage_info(const cho_chain* c, int stage, int* interp, int* decim, int* ntaps);
/* Recorder::startRecording: set_phase_inc(2*pi*(-shift/fs)); phase and filter histories carry over, as in the reference */
void cho_set_shift(cho_chain* c, int32_t shift_hz);
/* One work() call on n input samples (interleaved re,im). Writes at most cap output samples:
 * out_cf32 (nullable): the last resampler's output; out_i8 (nullable): after complex_to_interleaved_char(127).
 * Returns the number of output samples produced. */
int cho_process(cho_chain* c, const float* iq, int n, float* out_cf32, int8_t* out_i8, int cap);

/* DataController::pushTransmission payload (data_controller.cpp:27-42): header + samples with ^0x80. Returns bytes. */
int cho_transmission_payload(uint64_t time_ms, int32_t frequency, int32_t sample_rate, const int8_t* iq_i8, int nsamples, uint8_t* out, int cap);

#ifdef __cplusplus
}
#endif
#endif
