/*
 * specscan_oracle.h — CPU oracle for the spectral-scan hot path.
 *
 * TEST INFRASTRUCTURE ONLY. This is a plain-C, single-threaded restatement of the reference's
 * algorithm (shajen/rtl-sdr-scanner-cpp @ 2025-10-31), written to be read next to the reference
 * sources: every function cites the file:line it follows and keeps the reference's order of
 * floating-point operations (running sums included). Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; the product (libspecscan.so) never does.
 *
 * Parity pinning:
 *   - back end (Averager, average(), getMaxIndex, containsWithMargin, mostFrequentValue, getFft,
 *     getTunedFrequency): pinned by the reference's own gtest known-answer vectors, ported to
 *     tests/test_oracle_reference_vectors.py, and bit-compared against oracle/_ref (the reference's
 *     own .cpp files compiled in place by oracle/Makefile).
 *   - front end (Hamming window, forward FFT, half rotation): lives in GNU Radio / VOLK / FFTW, which
 *     are not vendored in the reference and not installed here. "PARITY UNPINNED" by any reference
 *     test; restated from the call site sources/radio/sdr_device.cpp:164 and anchored against fp64
 *     numpy.fft and MKL's FFTW3 interface (tests/test_oracle_fft.py).
 *
 * The exported set mirrors include/specscan.h with the prefix orc_.
 */
#ifndef SPECSCAN_ORACLE_H
#define SPECSCAN_ORACLE_H

#include <stdint.h>

#include "../include/specscan.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_ctx orc_ctx;

/* ---- stand-alone restated helpers (each cites its reference lines in the .c file) ---- */
void orc_hamming(int n, float* taps);
int orc_get_fft(int32_t sample_rate, int32_t max_step);
int32_t orc_get_tuned_frequency(int32_t frequency, int32_t step);
int32_t orc_index_to_shift(int32_t sample_rate, int fft_size, int index);
void orc_average(const float* input, float* output, int size, int group_size);
int orc_get_max_index(const float* data, int size, int index, int group_size);
/* keys: ascending ints (the std::map keys). Returns 1 and *found if a key is within the margin. */
int orc_contains_with_margin(const int* keys, int nkeys, int index, int margin, int* found);
int orc_most_frequent_value(const int* data, int n);
void orc_psd(const float* x_interleaved, float* out_db, int n, int32_t sample_rate);

/* FFT back ends for the restated fft_v. 0 = built-in fp32 radix-2, 1 = built-in fp64 (rounded to
 * fp32 at the end), 2 = MKL's FFTW3 single-precision interface (dlopen libmkl_rt.so; returns <0 from
 * orc_set_fft_backend when it cannot be loaded). Default 0. */
int orc_set_fft_backend(int which);
/* unnormalised forward c2c, interleaved re,im */
void orc_fft_forward(int n, const float* in, float* out);
/* gr::fft::fft_v<gr_complex,true>(n, window, shift=true): multiply, forward FFT, half rotation */
void orc_fft_v(int n, const float* window, const float* in, float* out);

/* Averager as an object, for the known-answer tests (tests/test_averager.cpp in the reference). */
typedef struct orc_averager orc_averager;
orc_averager* orc_averager_create(int size, int group_size);
void orc_averager_destroy(orc_averager* a);
void orc_averager_push(orc_averager* a, const float* data);
void orc_averager_reset(orc_averager* a);
const float* orc_averager_average(const orc_averager* a);
/* row 0 = oldest ... group_size-1 = newest (the deque order of Averager::data()) */
const float* orc_averager_row(const orc_averager* a, int row);

/* Spectrogram side branch (sources/radio/blocks/spectrogram.cpp), fed with raw PSD rows. */
typedef struct orc_spectrogram orc_spectrogram;
orc_spectrogram* orc_spectrogram_create(int in_size, int32_t sample_rate);
void orc_spectrogram_destroy(orc_spectrogram* g);
int orc_spectrogram_size(const orc_spectrogram* g);
void orc_spectrogram_process(orc_spectrogram* g, const float* psd_row);
/* returns the number of accumulated frames; out = int8(sum/count), mean_out (nullable) = the float before conversion */
int orc_spectrogram_send(orc_spectrogram* g, int8_t* out, float* mean_out);
/* DataController::pushSpectrogram's payload (sources/network/data_controller.cpp:44-57); byte count, or -1 if cap is short */
int orc_spectrogram_payload(uint64_t time_ms, int32_t frequency, int32_t sample_rate, const int8_t* row, int32_t size, uint8_t* out, int32_t cap);

/* ---- the chain behind the same boundary as ss_* ---- */
void orc_default_config(ss_config* cfg, int32_t sample_rate, int32_t center_hz);
int orc_create(const ss_config* cfg, orc_ctx** out);
void orc_destroy(orc_ctx* ctx);
const char* orc_last_error(const orc_ctx* ctx);
int orc_process(orc_ctx* ctx, const void* iq, int32_t nframes, const int64_t* t_ms,
                float* psd_db, float* rel_db, float* avg_db,
                int32_t* cand_off, int32_t* cand_idx, float* cand_avg, int32_t cand_cap);
int orc_set_frequency_range(orc_ctx* ctx, int32_t lo_hz, int32_t hi_hz);
int orc_reset(orc_ctx* ctx);
int orc_reset_noise(orc_ctx* ctx);
int orc_read_window(orc_ctx* ctx, int32_t plane, int32_t frame, int32_t lo, int32_t hi, float* out);
int orc_read_noise(orc_ctx* ctx, float* thr);

/* Per-stage wall time in seconds accumulated by orc_process since the last call to this function:
 * [0] convert+window+FFT+shift, [1] PSD, [2] noise, [3] averager push, [4] average(), [5] threshold */
void orc_stage_seconds(orc_ctx* ctx, double out[6]);

#ifdef __cplusplus
}
#endif
#endif
