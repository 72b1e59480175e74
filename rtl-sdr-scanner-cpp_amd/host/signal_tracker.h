// signal_tracker.h — host-side signal bookkeeping on top of the engine's per-frame planes and candidates:
// the part of the reference's Transmission block that is stateful, tiny and wall-clock driven, and therefore
// stays on the host. It turns per-frame candidates into the list the Scanner consumes,
//   std::vector<FrequencyFlush> = {tuned frequency shift in Hz, flush flag}, strongest first
// (sources/radio/help_structures.h:15-16, consumed at sources/scanner.cpp:43), exactly as the reference does.
//
// Mirrors, with the same order of operations (reference file:line):
//   Transmission::process        sources/radio/blocks/transmission.cpp:57-68   add -> update -> clear -> notify
//   Transmission::addSignals     :88-111   candidates sorted by avg power (std::sort, descending), margin test,
//                                          getBestIndex, std::map::insert (an existing key is kept)
//   Transmission::updateSignals  :113-130  window arg-max of avg and raw power around every tracked signal
//   Transmission::clearSignals   :70-86    timeout / maximal time
//   Transmission::getBestIndex   :132-154  mode of per-row window arg-maxes over the newest ceil(21/2) ring rows
//   Transmission::getSortedTransmissions :166-176
//   Signal                       sources/radio/signal.cpp:16-38
//   getMaxIndex / containsWithMargin / mostFrequentValue   sources/utils/collection_utils.h:8-50
//   getTunedFrequency            sources/utils/radio_utils.cpp:86-96
//   indexToShift                 sources/radio/sdr_device.cpp:154
// C++17, no GPU, no GNU Radio. The C entry points at the bottom are what tests and other hosts bind.
#pragma once
#include <cstdint>
#include <map>
#include <utility>
#include <vector>

namespace specscan {

struct TrackerConfig {
  int fft_size = 0;              // N
  int32_t sample_rate = 0;       // fs
  float start_level = 8.0f;      // Device::m_startLevel  (config.h:30)
  float stop_level = 5.0f;       // Device::m_stopLevel   (config.h:31)
  int group_size = 0;            // indexStep = ceil(recordingBandwidth / (fs/N))  (sdr_device.cpp:151)
  int grouping_y = 21;           // GROUPING_Y: rows in the averager ring (config.h:29)
  int64_t min_time_ms = 2000;    // Config::recordingMinTime
  int64_t timeout_ms = 2000;     // Config::recordingTimeout
  int64_t max_time_ms = 600000;  // TRANSMISSION_MAX_TIME, 10 minutes (config.h:21)
  int32_t tuning_step = 2500;    // Config::recordingTuningStep
};

struct FrequencyFlush {  // help_structures.h:15
  int32_t shift_hz;
  bool flush;
};

class SignalTracker {
 public:
  explicit SignalTracker(const TrackerConfig& config);

  // One frame. avg = average(Averager.average()) row, raw = NoiseLearner output row (both N floats);
  // candidates = bins passing transmission.cpp:91 in ascending order (what ss_process reports for the frame).
  // Returns the vector the reference hands to Notification::notify (transmission.cpp:67).
  const std::vector<FrequencyFlush>& processFrame(int64_t now_ms, const float* avg, const float* raw, const int32_t* candidates, int ncand);

  void reset();  // Transmission::resetBuffers (transmission.cpp:42-55): signals cleared, ring zeroed
  std::vector<int> signalKeys() const;

 private:
  struct Signal {  // sources/radio/signal.h
    int64_t first_ms, last_ms;
    float power;
    std::vector<int> indexes;
  };
  int getBestIndex(int index) const;
  int32_t indexToShift(int index) const;

  TrackerConfig m_config;
  std::map<int, Signal> m_signals;
  std::vector<float> m_ring;  // grouping_y rows of N floats, m_head = oldest (the Averager's deque)
  int m_head = 0;
  std::vector<FrequencyFlush> m_out;
};

}  // namespace specscan

extern "C" {
// C binding (libspecscan_host.so). Handles are opaque; arrays are caller-owned.
void* sst_create(int fft_size, int32_t sample_rate, float start_level, float stop_level, int group_size, int grouping_y,
                 int64_t min_time_ms, int64_t timeout_ms, int32_t tuning_step);
void sst_destroy(void* tracker);
void sst_reset(void* tracker);
// tx_out: pairs (shift_hz, flush) strongest first, capacity tx_cap pairs; sig_out: tracked keys ascending, capacity sig_cap.
// Returns the number of transmissions; *nsig receives the number of tracked signals.
int sst_process_frame(void* tracker, int64_t now_ms, const float* avg, const float* raw, const int32_t* candidates, int ncand,
                      int32_t* tx_out, int tx_cap, int32_t* sig_out, int sig_cap, int* nsig);
}
