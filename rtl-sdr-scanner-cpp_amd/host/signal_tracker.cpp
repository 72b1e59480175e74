// signal_tracker.cpp — see signal_tracker.h for what this mirrors and why it lives on the host.
#include "signal_tracker.h"

#include <algorithm>
#include <cstring>

namespace specscan {
namespace {

// getMaxIndex — sources/utils/collection_utils.h:8-14 (first maximum of the clipped window)
int getMaxIndex(const float* data, int size, int index, int groupSize) {
  const int lo = std::max(0, index - groupSize / 2);
  const int hi = std::min(size, index + groupSize / 2 + 1);
  return static_cast<int>(std::max_element(data + lo, data + hi) - data);
}

// containsWithMargin — collection_utils.h:16-27
template <typename T>
bool containsWithMargin(const std::map<int, T>& indexes, int index, int margin) {
  const int submargin = margin % 2 == 0 ? margin / 2 : margin / 2 + 1;
  const auto it = indexes.lower_bound(index - submargin);
  return it != indexes.end() && it->first <= index + submargin;
}

// mostFrequentValue — collection_utils.h:29-50: among the values with the highest count, ascending, the one at
// position size/2. The reference is undefined for an empty input (transmission.cpp:151); callers guard it.
int mostFrequentValue(std::vector<int> data) {
  std::sort(data.begin(), data.end());
  std::vector<std::pair<int, int>> runs;  // value, count
  for (size_t i = 0; i < data.size();) {
    size_t j = i;
    while (j < data.size() && data[j] == data[i]) ++j;
    runs.emplace_back(data[i], static_cast<int>(j - i));
    i = j;
  }
  int best = 0;
  for (const auto& r : runs) best = std::max(best, r.second);
  std::vector<int> tied;
  for (const auto& r : runs)
    if (r.second == best) tied.push_back(r.first);
  return tied[tied.size() / 2];
}

// getTunedFrequency — sources/utils/radio_utils.cpp:86-96
int32_t getTunedFrequency(int32_t frequency, int32_t step) {
  const int32_t rest = frequency < 0 ? frequency % step + step : frequency % step;
  const int32_t down = frequency - rest;
  const int32_t up = down + step;
  return rest < step - rest ? down : up;
}

}  // namespace

SignalTracker::SignalTracker(const TrackerConfig& config)
    : m_config(config), m_ring(static_cast<size_t>(config.fft_size) * static_cast<size_t>(config.grouping_y), 0.0f) {}

void SignalTracker::reset() {
  m_signals.clear();
  std::fill(m_ring.begin(), m_ring.end(), 0.0f);  // Averager::reset, averager.cpp:27-34
  m_head = 0;
}

std::vector<int> SignalTracker::signalKeys() const {
  std::vector<int> keys;
  for (const auto& kv : m_signals) keys.push_back(kv.first);
  return keys;
}

// indexToShift — sources/radio/sdr_device.cpp:150,154 (double step, truncation to int32, integer fs/2)
int32_t SignalTracker::indexToShift(int index) const {
  const double step = static_cast<double>(m_config.sample_rate) / m_config.fft_size;
  return static_cast<int32_t>(step * (index + 0.5)) - m_config.sample_rate / 2;
}

// Transmission::getBestIndex — transmission.cpp:132-154
int SignalTracker::getBestIndex(int index) const {
  const int n = m_config.fft_size;
  const int rows = m_config.grouping_y;
  std::vector<int> buffer;
  for (int i = rows / 2; i < rows; ++i) {  // deque positions size/2 .. size-1: the newest ceil(rows/2) rows
    const float* row = &m_ring[static_cast<size_t>((m_head + i) % rows) * n];
    const int best = getMaxIndex(row, n, index, m_config.group_size);
    if (m_config.start_level <= row[best]) buffer.push_back(best);
  }
  if (buffer.empty()) return index;  // undefined behaviour in the reference; keep the candidate itself
  return mostFrequentValue(std::move(buffer));
}

const std::vector<FrequencyFlush>& SignalTracker::processFrame(int64_t now, const float* avg, const float* raw, const int32_t* candidates,
                                                               int ncand) {
  const int n = m_config.fft_size;
  // Averager::push (averager.cpp:14-25): the oldest row is replaced, the ring advances
  std::memcpy(&m_ring[static_cast<size_t>(m_head) * n], raw, sizeof(float) * static_cast<size_t>(n));
  m_head = (m_head + 1) % m_config.grouping_y;

  // addSignals — transmission.cpp:88-111
  std::vector<int> indexes(candidates, candidates + ncand);
  std::sort(indexes.begin(), indexes.end(), [avg](const int& i1, const int& i2) { return avg[i1] > avg[i2]; });
  for (const int index : indexes) {
    if (!containsWithMargin(m_signals, index, m_config.group_size)) {
      const int best = getBestIndex(index);
      m_signals.insert({best, Signal{now, now, 0.0f, {}}});  // std::map::insert keeps an existing entry
    }
  }
  // updateSignals — transmission.cpp:113-130, Signal::newData — signal.cpp:16-24
  for (auto& kv : m_signals) {
    Signal& s = kv.second;
    const int bestAvg = getMaxIndex(avg, n, kv.first, m_config.group_size);
    s.power = avg[bestAvg];
    if (m_config.stop_level <= avg[bestAvg]) s.last_ms = now;
    if (m_config.start_level <= avg[bestAvg]) s.indexes.push_back(bestAvg);
  }
  // clearSignals — transmission.cpp:70-86, Signal::isTimeout / isMaximalTime — signal.cpp:28-30
  for (auto it = m_signals.begin(); it != m_signals.end();) {
    const Signal& s = it->second;
    if (s.last_ms + m_config.timeout_ms <= now || s.first_ms + m_config.max_time_ms <= now) {
      it = m_signals.erase(it);
    } else {
      ++it;
    }
  }
  // getSortedTransmissions — transmission.cpp:166-176, Signal::needFlush — signal.cpp:32
  std::vector<int> keys;
  for (const auto& kv : m_signals) keys.push_back(kv.first);
  std::sort(keys.begin(), keys.end(), [this](const int& i1, const int& i2) { return m_signals.at(i1).power > m_signals.at(i2).power; });
  m_out.clear();
  for (const int key : keys) {
    const Signal& s = m_signals.at(key);
    const bool flush = s.last_ms == now && s.first_ms + m_config.min_time_ms <= now;
    m_out.push_back({getTunedFrequency(indexToShift(key), m_config.tuning_step), flush});
  }
  return m_out;
}

}  // namespace specscan

extern "C" {

void* sst_create(int fft_size, int32_t sample_rate, float start_level, float stop_level, int group_size, int grouping_y, int64_t min_time_ms,
                 int64_t timeout_ms, int32_t tuning_step) {
  if (fft_size <= 0 || sample_rate <= 0 || grouping_y <= 0 || tuning_step <= 0 || group_size < 0) return nullptr;
  specscan::TrackerConfig cfg;
  cfg.fft_size = fft_size;
  cfg.sample_rate = sample_rate;
  cfg.start_level = start_level;
  cfg.stop_level = stop_level;
  cfg.group_size = group_size;
  cfg.grouping_y = grouping_y;
  cfg.min_time_ms = min_time_ms;
  cfg.timeout_ms = timeout_ms;
  cfg.tuning_step = tuning_step;
  return new specscan::SignalTracker(cfg);
}

void sst_destroy(void* tracker) { delete static_cast<specscan::SignalTracker*>(tracker); }

void sst_reset(void* tracker) { static_cast<specscan::SignalTracker*>(tracker)->reset(); }

int sst_process_frame(void* tracker, int64_t now_ms, const float* avg, const float* raw, const int32_t* candidates, int ncand, int32_t* tx_out,
                      int tx_cap, int32_t* sig_out, int sig_cap, int* nsig) {
  auto* t = static_cast<specscan::SignalTracker*>(tracker);
  const auto& tx = t->processFrame(now_ms, avg, raw, candidates, ncand);
  int k = 0;
  for (const auto& x : tx) {
    if (k >= tx_cap) break;
    tx_out[2 * k] = x.shift_hz;
    tx_out[2 * k + 1] = x.flush ? 1 : 0;
    ++k;
  }
  if (nsig) {
    const auto keys = t->signalKeys();
    int m = 0;
    for (const int key : keys) {
      if (m >= sig_cap) break;
      sig_out[m++] = key;
    }
    *nsig = static_cast<int>(keys.size());
  }
  return static_cast<int>(tx.size());
}

}  // extern "C"
