// recorder_bank.h — the reference's recorders of one device on top of the GPU channeliser:
//   Recorder x recordersCount            sources/radio/recorder.cpp:14-98, sources/radio/sdr_device.cpp:39-41
//   SdrDevice::updateRecordings          sources/radio/sdr_device.cpp:82-144
//   stream_to_vector + Buffer            recorder.cpp:35-39, sources/radio/blocks/buffer.h:16-58
// The per-slot GNU Radio chains (rotator -> resamplers -> int8) are ONE sc_ctx (include/specscan_channelizer.h); what is
// left on the host is exactly what the reference keeps on the host: which shift is recorded by which slot, the 100 ms
// items (`samplesSize` int8 samples, recorder.cpp:35) with their arrival times, and flush -> publish in the argument
// order of DataController::pushTransmission (sources/network/data_controller.cpp:27). Header-only C++17; links libspecscan.so.
#pragma once
#include <specscan_channelizer.h>

#include <algorithm>
#include <cstdint>
#include <functional>
#include <limits>
#include <set>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace specscan {

class RecorderBank {
 public:
  // time of the item (ms), centre frequency of the recording (device centre + shift), sample rate of the recording,
  // int8 (re,im) samples, their count — DataController::pushTransmission(time, frequency, sampleRate, data, size)
  using Publish = std::function<void(int64_t time_ms, int32_t frequency, int32_t sample_rate, const int8_t* iq, int nsamples)>;
  using ShiftFlush = std::pair<int32_t, bool>;  // FrequencyFlush, sources/radio/help_structures.h:15

  RecorderBank(int32_t sample_rate, int32_t bandwidth, int recorders, int max_samples, Publish publish, int device_id = 0)
      : m_bandwidth(bandwidth), m_publish(std::move(publish)), m_slots((size_t)recorders) {
    sc_config cfg;
    sc_default_config(&cfg, sample_rate, bandwidth);
    cfg.channels = recorders;
    cfg.max_samples = max_samples;
    cfg.device_id = device_id;
    if (sc_create(&cfg, &m_ctx) != 0) throw std::runtime_error(std::string("RecorderBank: ") + sc_last_error(nullptr));
    // roundUp(recordingBandwidth * RECORDER_FLUSH_INTERVAL / 1000, 4096), recorder.cpp:35, config.h:19
    const int raw = bandwidth * 100 / 1000;
    m_itemSamples = raw % 4096 == 0 ? raw : (raw / 4096 + 1) * 4096;
    m_cap = sc_output_capacity(m_ctx, max_samples);
    m_out.resize((size_t)recorders * (size_t)m_cap * 2);
    m_counts.resize((size_t)recorders);
  }
  ~RecorderBank() { sc_destroy(m_ctx); }
  RecorderBank(const RecorderBank&) = delete;
  RecorderBank& operator=(const RecorderBank&) = delete;

  int itemSamples() const { return m_itemSamples; }
  bool isRecording(int slot) const { return m_slots[(size_t)slot].recording; }
  int32_t shift(int slot) const { return m_slots[(size_t)slot].shift; }  // Recorder::getShift: max() while idle
  int64_t durationMs(int slot) const { return m_slots[(size_t)slot].last_ms - m_slots[(size_t)slot].first_ms; }

  // The device stream: what every work() of the source delivers. Slots that record get their samples.
  void work(const void* iq_cf32, int nsamples, int64_t now_ms) {
    if (sc_process(m_ctx, iq_cf32, nsamples, m_out.data(), nullptr, m_counts.data(), m_cap) != 0) throw std::runtime_error(sc_last_error(m_ctx));
    for (size_t k = 0; k < m_slots.size(); ++k) {
      Slot& s = m_slots[k];
      if (!s.recording) continue;
      const int8_t* p = m_out.data() + k * (size_t)m_cap * 2;
      s.pending.insert(s.pending.end(), p, p + (size_t)m_counts[k] * 2);
      // stream_to_vector: complete items of itemSamples go to the Buffer, stamped with their arrival time (buffer.h:35-37)
      size_t done = 0;
      while (s.pending.size() - done >= (size_t)m_itemSamples * 2) {
        s.items.insert(s.items.end(), s.pending.begin() + (long)done, s.pending.begin() + (long)(done + (size_t)m_itemSamples * 2));
        s.item_times.push_back(now_ms);
        done += (size_t)m_itemSamples * 2;
      }
      s.pending.erase(s.pending.begin(), s.pending.begin() + (long)done);
    }
  }

  // SdrDevice::updateRecordings(sortedShifts), sdr_device.cpp:82-144
  void updateRecordings(const std::vector<ShiftFlush>& sortedShifts, int32_t center_frequency, int64_t now_ms) {
    const auto waiting = [&](int32_t shift) {
      return std::find_if(sortedShifts.begin(), sortedShifts.end(), [shift](const ShiftFlush& sf) { return sf.first == shift; }) != sortedShifts.end();
    };
    for (size_t k = 0; k < m_slots.size(); ++k) {
      if (m_slots[k].recording && !waiting(m_slots[k].shift)) stopRecording((int)k);
    }
    for (const auto& sf : sortedShifts) {
      const auto it = std::find_if(m_slots.begin(), m_slots.end(), [&](const Slot& s) { return s.shift == sf.first; });
      if (it != m_slots.end()) {
        if (sf.second) flush((int)(it - m_slots.begin()), now_ms);
      } else {
        const auto free_it = std::find_if(m_slots.begin(), m_slots.end(), [](const Slot& s) { return !s.recording; });
        if (free_it != m_slots.end()) startRecording((int)(free_it - m_slots.begin()), center_frequency, sf.first, now_ms);
        else m_ignored.insert(sf.first);  // "no recorders available" (logged once per shift in the reference)
      }
    }
    for (auto it = m_ignored.begin(); it != m_ignored.end();) it = waiting(*it) ? std::next(it) : m_ignored.erase(it);
  }

  // Recorder::startRecording / stopRecording / flush (recorder.cpp:58-98)
  void startRecording(int slot, int32_t frequency, int32_t shift, int64_t now_ms) {
    Slot& s = m_slots[(size_t)slot];
    if (s.recording) return;  // "can not start recording, recorder already recording"
    s.first_ms = s.last_ms = now_ms;
    s.frequency = frequency;
    s.shift = shift;
    if (sc_start(m_ctx, slot, shift) != 0) throw std::runtime_error(sc_last_error(m_ctx));
    s.recording = true;
    s.items.clear();  // m_buffer->clear(); samples still inside stream_to_vector (s.pending) stay, as in the reference
    s.item_times.clear();
  }
  void stopRecording(int slot) {
    Slot& s = m_slots[(size_t)slot];
    if (!s.recording) return;
    s.frequency = s.shift = std::numeric_limits<int32_t>::max();
    sc_stop(m_ctx, slot);
    s.recording = false;
    s.items.clear();
    s.item_times.clear();
  }
  void flush(int slot, int64_t now_ms) {
    Slot& s = m_slots[(size_t)slot];
    s.last_ms = now_ms;
    for (size_t i = 0; i < s.item_times.size(); ++i) {  // Buffer::popSingleSample
      if (m_publish) m_publish(s.item_times[i], s.frequency + s.shift, m_bandwidth, s.items.data() + i * (size_t)m_itemSamples * 2, m_itemSamples);
    }
    s.items.clear();
    s.item_times.clear();
  }

 private:
  struct Slot {
    bool recording = false;
    int32_t frequency = std::numeric_limits<int32_t>::max(), shift = std::numeric_limits<int32_t>::max();
    int64_t first_ms = 0, last_ms = 0;
    std::vector<int8_t> pending;  // inside stream_to_vector: less than one item
    std::vector<int8_t> items;    // the Buffer: whole items
    std::vector<int64_t> item_times;
  };
  const int32_t m_bandwidth;
  Publish m_publish;
  sc_ctx* m_ctx = nullptr;
  std::vector<Slot> m_slots;
  std::set<int32_t> m_ignored;
  int m_itemSamples = 0;
  int32_t m_cap = 0;
  std::vector<int8_t> m_out;
  std::vector<int32_t> m_counts;
};

}  // namespace specscan
