// gpu_spectrum_block.h — the reference-side adapter: one GNU Radio block that replaces the chain
//   Decimator -> fft_v -> PSD -> NoiseLearner -> Transmission(front)
// wired at sources/radio/sdr_device.cpp:161-168 of shajen/rtl-sdr-scanner-cpp, by calling libspecscan.so
// through the C ABI of include/specscan.h. Header-only, C++17, depends on GNU Radio's sync_block only
// (tests compile it against oracle/stubs/gnuradio/sync_block.h).
//
//   input  0: items of N*D gr_complex  (what stream_to_vector + Blocker deliver, sdr_device.cpp:161-162)
//   output 0: items of N float, the raw PSD in dB (what PSD::work produced, psd.cpp:18-20) — feeds the
//             Spectrogram side branch (sdr_device.cpp:170-171) unchanged
//   candidates: per frame, the bins passing transmission.cpp:91 and their avg power, handed to a callback
//             on the work() thread, in frame order — the input of Transmission::addSignals' sort (:95)
//
// Optionally the block also runs the host-side remainder of Transmission (enableTracker): every frame's avg / rel rows and
// candidates go through specscan::SignalTracker (host/signal_tracker.h) and the resulting std::vector<FrequencyFlush> —
// what Transmission::process hands to m_notification.notify (transmission.cpp:67) — goes to a callback, so that the
// Scanner thread (sources/scanner.cpp:36-64) sees exactly the interface it sees today.
//
// With SS_FLAG_SPECTROGRAM in the config the Spectrogram side branch runs on the GPU as well (enableSpectrogram): the
// block keeps Spectrogram::send's 1000 ms gate per centre frequency (spectrogram.cpp:62-75) and hands the int8 row to a
// callback with DataController::pushSpectrogram's arguments; ss_spectrogram_payload frames it for the wire. The gate is
// looked at once per work() call (the reference looks after every frame), so a row closes at a batch boundary.
//
// Control calls mirror what SdrDevice does to the blocks it owns: setFrequencyRange (sdr_device.cpp:54-80)
// -> set_frequency_range() + reset_buffers(). They come from the Scanner's thread while work() runs on the flowgraph's:
// the library serialises its own entry points, and m_mutex does for the host-side tracker what Transmission::m_mutex does
// in the reference (taken in work() and in resetBuffers(), transmission.cpp:35,43).
#pragma once
#include <gnuradio/sync_block.h>
#include <specscan.h>

#include "signal_tracker.h"

#include <atomic>
#include <chrono>
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

class GpuSpectrum : virtual public gr::sync_block {
 public:
  // frame index inside this work() call, candidate bins (ascending), their avg power in dB
  using CandidateCallback = std::function<void(int frame, const int32_t* bins, const float* avg_db, int count)>;

  GpuSpectrum(const ss_config& config, CandidateCallback on_candidates)
      : gr::sync_block("GpuSpectrum", gr::io_signature::make(1, 1, sizeof(gr_complex) * config.fft_size * config.decim),
                       gr::io_signature::make(1, 1, sizeof(float) * config.fft_size)),
        m_config(config),
        m_onCandidates(std::move(on_candidates)) {
    // construction errors throw, like the reference's blocks (caught per device at sources/main.cpp:60-62)
    if (ss_create(&m_config, &m_ctx) != SS_OK) {
      throw std::runtime_error(std::string("GpuSpectrum: ") + ss_last_error(nullptr));
    }
    m_offsets.resize(static_cast<size_t>(m_config.max_batch) + 1);
    m_bins.resize(static_cast<size_t>(m_config.max_batch) * 1024);
    m_avg.resize(m_bins.size());
    m_times.resize(static_cast<size_t>(m_config.max_batch));
    m_center = (m_config.range_lo + m_config.range_hi) / 2;
  }

  ~GpuSpectrum() override { ss_destroy(m_ctx); }

  GpuSpectrum(const GpuSpectrum&) = delete;
  GpuSpectrum& operator=(const GpuSpectrum&) = delete;

  using TransmissionCallback = std::function<void(const std::vector<specscan::FrequencyFlush>&)>;
  using Clock = std::function<int64_t()>;  // milliseconds; the reference's getTime() (utils/utils.cpp:14)

  // Run Transmission's signal bookkeeping on every frame and report what Notification::notify would receive.
  void enableTracker(const specscan::TrackerConfig& config, TransmissionCallback on_transmissions) {
    m_tracker = std::make_unique<specscan::SignalTracker>(config);
    m_onTransmissions = std::move(on_transmissions);
    m_rel.resize(static_cast<size_t>(m_config.max_batch) * static_cast<size_t>(m_config.fft_size));
    m_avgPlane.resize(m_rel.size());
  }
  void setClock(Clock clock) { m_clock = std::move(clock); }

  // time (ms), centre frequency, sample rate, int8 row, bins — DataController::pushSpectrogram(time, frequency, sampleRate, data, size)
  using SpectrogramCallback = std::function<void(int64_t time_ms, int32_t frequency, int32_t sample_rate, const int8_t* row, int size)>;
  void enableSpectrogram(SpectrogramCallback on_row) {
    const int size = ss_spectrogram_size(m_ctx);
    if (size <= 0) throw std::runtime_error("GpuSpectrum: the context was created without SS_FLAG_SPECTROGRAM");
    m_spectrogramRow.resize(static_cast<size_t>(size));
    m_onSpectrogram = std::move(on_row);
  }

  int work(int noutput_items, gr_vector_const_void_star& input_items, gr_vector_void_star& output_items) override {
    const int nframes = noutput_items < m_config.max_batch ? noutput_items : m_config.max_batch;
    // the reference's blocks read the wall clock per frame (noise_learner.cpp:18, transmission.cpp:62)
    const int64_t now = m_clock ? m_clock()
                                : std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now().time_since_epoch()).count();
    for (int f = 0; f < nframes; ++f) m_times[static_cast<size_t>(f)] = now;
    const int status = ss_process(m_ctx, input_items[0], nframes, m_times.data(), static_cast<float*>(output_items[0]),
                                  m_tracker ? m_rel.data() : nullptr, m_tracker ? m_avgPlane.data() : nullptr, m_offsets.data(), m_bins.data(),
                                  m_avg.data(), static_cast<int32_t>(m_bins.size()));
    size_t grow_to = 0;
    if (status == SS_ERR_CAND_OVERFLOW) {
      // More candidates than the lists hold (the reference never drops one): the offsets are exact, so once this call's
      // own lists — truncated, later frames incomplete — have been handed on, the lists are grown to fit; the call is
      // flagged (lastError, candidateOverflows) and the next one fits.
      const size_t need = static_cast<size_t>(m_offsets[static_cast<size_t>(nframes)]);
      const size_t per_frame = (need + static_cast<size_t>(nframes) - 1) / static_cast<size_t>(nframes);
      grow_to = static_cast<size_t>(m_config.max_batch) * (per_frame + 64);
      m_lastError = "candidate lists overflowed; capacity raised to " + std::to_string(grow_to);
      ++m_overflows;
    } else if (status != SS_OK) {
      // work() has no error channel in the reference either (sdr_source.cpp:37-41 logs and exits): produce nothing
      m_lastError = ss_last_error(m_ctx);
      return 0;
    }
    if (m_onCandidates) {
      const int32_t cap = static_cast<int32_t>(m_bins.size());
      for (int f = 0; f < nframes; ++f) {
        const int32_t begin = m_offsets[static_cast<size_t>(f)] < cap ? m_offsets[static_cast<size_t>(f)] : cap;
        const int32_t end = m_offsets[static_cast<size_t>(f) + 1] < cap ? m_offsets[static_cast<size_t>(f) + 1] : cap;
        m_onCandidates(f, m_bins.data() + begin, m_avg.data() + begin, end - begin);
      }
    }
    if (m_tracker) {
      std::lock_guard<std::mutex> lock(m_mutex);
      const int32_t cap = static_cast<int32_t>(m_bins.size());
      const size_t n = static_cast<size_t>(m_config.fft_size);
      for (int f = 0; f < nframes; ++f) {
        const int32_t begin = m_offsets[static_cast<size_t>(f)] < cap ? m_offsets[static_cast<size_t>(f)] : cap;
        const int32_t end = m_offsets[static_cast<size_t>(f) + 1] < cap ? m_offsets[static_cast<size_t>(f) + 1] : cap;
        const auto& tx = m_tracker->processFrame(now, m_avgPlane.data() + static_cast<size_t>(f) * n, m_rel.data() + static_cast<size_t>(f) * n,
                                                 m_bins.data() + begin, end - begin);
        if (m_onTransmissions) m_onTransmissions(tx);
      }
    }
    if (m_onSpectrogram) {
      // Container's ctor stamps m_lastDataSendTime with the time of the centre frequency's first frame (spectrogram.cpp:9,36)
      const int32_t center = m_center.load();
      const auto it = m_spectrogramSent.try_emplace(center, now).first;
      if (it->second + 1000 < now) {  // SPECTROGRAM_SEND_INTERVAL, config.h:38; spectrogram.cpp:65
        if (ss_spectrogram_read(m_ctx, m_spectrogramRow.data(), nullptr) > 0) {
          m_onSpectrogram(now, center, m_config.sample_rate, m_spectrogramRow.data(), static_cast<int>(m_spectrogramRow.size()));
        }
        it->second = now;
      }
    }
    if (grow_to > m_bins.size()) {
      m_bins.resize(grow_to);
      m_avg.resize(grow_to);
    }
    return nframes;
  }

  // SdrDevice::setFrequencyRange's effect on the chain (sdr_device.cpp:66,74,77)
  void setFrequencyRange(int32_t lo_hz, int32_t hi_hz) {
    ss_set_frequency_range(m_ctx, lo_hz, hi_hz);
    m_center = (lo_hz + hi_hz) / 2;
  }
  void resetBuffers() {  // Transmission::resetBuffers, transmission.cpp:42-55: signals cleared, averager reset
    ss_reset(m_ctx);
    std::lock_guard<std::mutex> lock(m_mutex);
    if (m_tracker) m_tracker->reset();
  }
  const std::string& lastError() const { return m_lastError; }
  int candidateOverflows() const { return m_overflows; }  // work() calls whose lists were cut (capacity grows each time)

 private:
  ss_config m_config;
  ss_ctx* m_ctx = nullptr;
  CandidateCallback m_onCandidates;
  std::vector<int32_t> m_offsets, m_bins;
  std::vector<float> m_avg;
  std::vector<int64_t> m_times;
  std::string m_lastError;
  std::unique_ptr<specscan::SignalTracker> m_tracker;
  TransmissionCallback m_onTransmissions;
  Clock m_clock;
  std::vector<float> m_rel, m_avgPlane;
  SpectrogramCallback m_onSpectrogram;
  std::vector<int8_t> m_spectrogramRow;
  std::map<int32_t, int64_t> m_spectrogramSent;  // Container::m_lastDataSendTime per centre frequency
  std::atomic<int32_t> m_center{0};  // written by the retuning thread (setFrequencyRange), read by work()
  std::mutex m_mutex;                // the tracker: work() on the flowgraph thread, resetBuffers() on the Scanner's
  int m_overflows = 0;
};
