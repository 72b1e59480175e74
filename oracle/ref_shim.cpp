// ref_shim.cpp — C driver around the REFERENCE'S OWN source files. TEST INFRASTRUCTURE ONLY.
//
// oracle/Makefile compiles, in place from /root/reference/sources and unmodified:
//   radio/blocks/psd.cpp  radio/blocks/noise_learner.cpp  radio/blocks/transmission.cpp
//   radio/averager.cpp    radio/signal.cpp   utils/utils.cpp   utils/radio_utils.cpp
//   performance_logger.cpp            (+ header-only utils/collection_utils.h, notification.h)
//   radio/blocks/spectrogram.cpp      network/data_controller.cpp   (against stubs/network/mqtt.h, which keeps what
//                                     DataController publishes)
// against oracle/stubs/ (stand-ins for the GNU Radio / spdlog / nlohmann / boost headers) and links them
// with this file into oracle/_ref/libref_specscan.so. Nothing of the reference is copied into the repo;
// the .so is git-ignored.
//
// This file supplies only what those sources leave undefined and cannot be compiled here:
//   * Config's private constructor and the four getters the path reads (sources/config.cpp:72-86,
//     :134,:141-143 need nlohmann-json and SoapySDR),
//   * getTime() (sources/utils/utils.cpp:14 is renamed away with -DgetTime=... so the clock can be
//     injected; every other line of utils.cpp is the reference's),
//   * the index<->frequency lambdas of SdrDevice::setupChains (sources/radio/sdr_device.cpp:149-158;
//     sdr_device.cpp needs GNU Radio proper),
//   * gr::fft::fft_v (GNU Radio, un-vendored): restated in specscan_oracle.c (orc_fft_v).
//
// It then drives PSD::work -> NoiseLearner::work -> Transmission::work one frame at a time, exactly
// the order of the flowgraph (sdr_device.cpp:168), and reads the blocks' private members to report
// what they computed.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <complex>
#include <condition_variable>
#include <deque>
#include <cstdint>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <optional>
#include <set>
#include <sstream>
#include <string>
#include <utility>
#include <unordered_map>
#include <vector>

#define private public
#define protected public
#include <config.h>
#include <logger.h>
#include <radio/averager.h>
#include <radio/blocks/noise_learner.h>
#include <radio/blocks/psd.h>
#include <radio/blocks/spectrogram.h>
#include <radio/blocks/transmission.h>
#include <radio/signal.h>
#include <utils/collection_utils.h>
#include <utils/radio_utils.h>
#include <utils/utils.h>
#undef private
#undef protected

#include "specscan_oracle.h"

// ---- injected clock (replaces sources/utils/utils.cpp:14) ----
static int64_t g_now_ms = 0;
std::chrono::milliseconds getTime() { return std::chrono::milliseconds(g_now_ms); }

// ---- Config pieces (sources/config.cpp:72-86 ctor, :134,:141-143 getters) ----
namespace {
std::vector<FrequencyRange> g_ignored;
int g_min_time_ms = 2000, g_timeout_ms = 2000, g_tuning_step = 2500, g_bandwidth = 32000;  // config.example.json:9-14
}  // namespace

Config::Config(const nlohmann::json& json)
    : m_json(json),
      m_devices(),
      m_isColorLogEnabled(false),
      m_consoleLogLevel(spdlog::level::off),
      m_fileLogLevel(spdlog::level::off),
      m_ignoredRanges(g_ignored),
      m_recordingBandwidth(g_bandwidth),
      m_recordingMinTime(g_min_time_ms),
      m_recordingTimeout(g_timeout_ms),
      m_recordingTuningStep(g_tuning_step),
      m_workers(0),
      m_mqttUrl(),
      m_mqttUsername(),
      m_mqttPassword() {}
std::vector<FrequencyRange> Config::ignoredRanges() const { return m_ignoredRanges; }
std::chrono::milliseconds Config::recordingMinTime() const { return m_recordingMinTime; }
std::chrono::milliseconds Config::recordingTimeout() const { return m_recordingTimeout; }
Frequency Config::recordingTuningStep() const { return m_recordingTuningStep; }

namespace {
struct RefChain {
  int fftSize;
  Frequency sampleRate;
  FrequencyRange range;
  std::unique_ptr<Config> config;
  Device device;
  TransmissionNotification notification;
  std::shared_ptr<PSD> psd;
  std::shared_ptr<NoiseLearner> noise;
  std::shared_ptr<Transmission> transmission;
  std::vector<float> window;
  std::vector<gr_complex> spectrum;
  std::vector<float> psdRow, relRow;

  Frequency getFrequency() const { return (range.first + range.second) / 2; }  // sdr_device.cpp:146
};

void ensureLogger() {
  if (!Logger::_logger) {
    Logger::_logger = std::make_shared<spdlog::logger>();
    Logger::_isColorLogEnabled = false;
  }
}
}  // namespace

extern "C" {

void ref_set_time(int64_t ms) { g_now_ms = ms; }

void* ref_create(int fft_size, int sample_rate, float start_level, float stop_level, int range_lo, int range_hi, int n_ignored,
                 const int32_t* ignored, int group_size, int min_time_ms, int timeout_ms, int tuning_step) {
  ensureLogger();
  g_ignored.clear();
  for (int i = 0; i < n_ignored; ++i) g_ignored.emplace_back(ignored[2 * i], ignored[2 * i + 1]);
  g_min_time_ms = min_time_ms;
  g_timeout_ms = timeout_ms;
  g_tuning_step = tuning_step;

  auto* c = new RefChain();
  c->fftSize = fft_size;
  c->sampleRate = sample_rate;
  c->range = {range_lo, range_hi};
  c->config.reset(new Config(nlohmann::json()));
  c->device.m_enabled = true;
  c->device.m_sampleRate = sample_rate;
  c->device.m_startLevel = start_level;
  c->device.m_stopLevel = stop_level;
  c->device.m_ranges = {c->range};

  // the lambdas of SdrDevice::setupChains, sources/radio/sdr_device.cpp:150-158
  const auto step = static_cast<double>(c->sampleRate) / fft_size;
  const auto indexToFrequency = [c, step](const int index) { return c->getFrequency() + static_cast<Frequency>(step * (index + 0.5)) - c->sampleRate / 2; };
  const auto indexToShift = [c, step](const int index) { return static_cast<Frequency>(step * (index + 0.5)) - c->sampleRate / 2; };
  const auto isIndexInRange = [c, indexToFrequency](const int index) {
    const auto f = indexToFrequency(index);
    return c->range.first <= f && f <= c->range.second;
  };
  // construction as at sdr_device.cpp:165-167
  c->psd = std::make_shared<PSD>(fft_size, c->sampleRate);
  c->noise = std::make_shared<NoiseLearner>(fft_size, [c]() { return c->getFrequency(); }, indexToFrequency);
  c->transmission = std::make_shared<Transmission>(*c->config, c->device, fft_size, group_size, c->notification, indexToFrequency, indexToShift, isIndexInRange);

  c->window.resize(fft_size);
  orc_hamming(fft_size, c->window.data());
  c->spectrum.resize(fft_size);
  c->psdRow.resize(fft_size);
  c->relRow.resize(fft_size);
  return c;
}

void ref_destroy(void* h) { delete static_cast<RefChain*>(h); }

void ref_set_window(void* h, const float* taps) {
  auto* c = static_cast<RefChain*>(h);
  std::memcpy(c->window.data(), taps, sizeof(float) * c->fftSize);
}

void ref_set_range(void* h, int lo, int hi) { static_cast<RefChain*>(h)->range = {lo, hi}; }

void ref_reset(void* h) { static_cast<RefChain*>(h)->transmission->resetBuffers(); }       // sdr_device.cpp:74
void ref_reset_noise(void* h) { static_cast<RefChain*>(h)->noise->resetBuffers(); }

// One frame through PSD -> NoiseLearner -> Transmission. `spectrum` = fft_v output (N gr_complex).
// Outputs (any may be null): psd/rel/avg rows; cand = indexes passing transmission.cpp:91 in ascending
// order (capacity N); tx = the vector handed to Notification::notify (transmission.cpp:67) as
// (shift Hz, flush) pairs (capacity 2*N ints); sig = keys of m_signals after the frame (capacity N).
int ref_process_spectrum(void* h, const float* spectrum, float* psd, float* rel, float* avg, int32_t* cand, int32_t* ncand,
                         int32_t* tx, int32_t* ntx, int32_t* sig, int32_t* nsig) {
  auto* c = static_cast<RefChain*>(h);
  const int n = c->fftSize;
  {
    gr_vector_const_void_star in{spectrum};
    gr_vector_void_star out{c->psdRow.data()};
    c->psd->work(1, in, out);
  }
  {
    gr_vector_const_void_star in{c->psdRow.data()};
    gr_vector_void_star out{c->relRow.data()};
    c->noise->work(1, in, out);
  }
  {
    gr_vector_const_void_star in{c->relRow.data()};
    gr_vector_void_star out{};
    c->transmission->work(1, in, out);
  }
  if (psd) std::memcpy(psd, c->psdRow.data(), sizeof(float) * n);
  if (rel) std::memcpy(rel, c->relRow.data(), sizeof(float) * n);

  // what Transmission::process computed for this frame (transmission.cpp:58-61), from its own members
  const auto& bufferPower = c->transmission->m_averager.average();
  std::vector<float> avgPower(bufferPower.size(), 0.0);
  average(bufferPower.data(), avgPower.data(), bufferPower.size(), GROUPING_X);
  if (avg) std::memcpy(avg, avgPower.data(), sizeof(float) * n);
  if (cand && ncand) {
    int k = 0;
    for (int i = 0; i < n; ++i) {  // the predicate of transmission.cpp:91
      if (c->device.m_startLevel <= avgPower[i] && c->transmission->m_isIndexInRange(i) && !c->transmission->isIndexIgnored(i)) {
        cand[k++] = i;
      }
    }
    *ncand = k;
  }
  const auto transmissions = c->notification.wait();  // value stored by notify() at transmission.cpp:67
  if (tx && ntx) {
    int k = 0;
    for (const auto& [shift, flush] : transmissions) {
      tx[2 * k] = shift;
      tx[2 * k + 1] = flush ? 1 : 0;
      ++k;
    }
    *ntx = k;
  }
  if (sig && nsig) {
    int k = 0;
    for (const auto& kv : c->transmission->m_signals) sig[k++] = kv.first;
    *nsig = k;
  }
  return 0;
}

// One frame of CF32 IQ: restated fft_v (window, FFT back end chosen with orc_set_fft_backend, shift),
// then the reference blocks.
int ref_process_iq(void* h, const float* iq, float* psd, float* rel, float* avg, int32_t* cand, int32_t* ncand, int32_t* tx,
                   int32_t* ntx, int32_t* sig, int32_t* nsig) {
  auto* c = static_cast<RefChain*>(h);
  orc_fft_v(c->fftSize, c->window.data(), iq, reinterpret_cast<float*>(c->spectrum.data()));
  return ref_process_spectrum(h, reinterpret_cast<const float*>(c->spectrum.data()), psd, rel, avg, cand, ncand, tx, ntx, sig, nsig);
}

// averager ring row r (0 = oldest) of the Transmission block, and the learned ceiling
void ref_ring_row(void* h, int r, float* out) {
  auto* c = static_cast<RefChain*>(h);
  const auto& row = c->transmission->m_averager.data().at(r);
  std::memcpy(out, row.data(), sizeof(float) * row.size());
}
int ref_noise(void* h, float* thr) {
  auto* c = static_cast<RefChain*>(h);
  auto it = c->noise->m_noise.find(c->getFrequency());
  if (it == c->noise->m_noise.end() || it->second.m_threshold.empty()) return -1;
  std::memcpy(thr, it->second.m_threshold.data(), sizeof(float) * c->fftSize);
  return it->second.m_isReady ? 1 : 0;
}

// ---- thin wrappers over the reference's pure helpers, for the known-answer tests ----
void ref_average(const float* in, float* out, int size, int group) { average(in, out, size, group); }
int ref_get_max_index(const float* data, int size, int index, int group) { return getMaxIndex(data, size, index, group); }
int ref_get_fft(int sample_rate, int max_step) { return getFft(sample_rate, max_step); }
int ref_get_tuned_frequency(int f, int step) { return getTunedFrequency(f, step); }
// getResamplersFactors (utils/radio_utils.cpp:128-152): pairs (interpolation, decimation)
int ref_get_resamplers_factors(int sample_rate, int bandwidth, int threshold, int* interp, int* decim, int cap) {
  const auto r = getResamplersFactors(sample_rate, bandwidth, threshold);
  for (size_t i = 0; i < r.size() && (int)i < cap; ++i) {
    interp[i] = r[i].first;
    decim[i] = r[i].second;
  }
  return (int)r.size();
}
// getRawFileName (utils/radio_utils.cpp:78-84) reads the wall clock itself: the caller compares within one second
int ref_get_raw_file_name(const char* label, const char* extension, int frequency, int sample_rate, char* out, int cap) {
  const std::string s = getRawFileName(label, extension, frequency, sample_rate);
  if ((int)s.size() + 1 > cap) return -1;
  memcpy(out, s.c_str(), s.size() + 1);
  return (int)s.size();
}
int ref_contains_with_margin(const int* keys, int nkeys, int index, int margin, int* found) {
  std::map<int, bool> m;
  for (int i = 0; i < nkeys; ++i) m[keys[i]] = false;
  const auto r = containsWithMargin(m, index, margin);
  if (r && found) *found = *r;
  return r ? 1 : 0;
}
int ref_most_frequent_value(const int* data, int n) {
  std::vector<int> v(data, data + n);
  return mostFrequentValue(v);
}
void ref_psd(const float* spectrum, float* out, int n, int sample_rate) {
  ensureLogger();
  PSD psd(n, sample_rate);
  gr_vector_const_void_star in{spectrum};
  gr_vector_void_star o{out};
  psd.work(1, in, o);
}

// ---- the Spectrogram block and DataController's framing, on the reference's own code ----
// Spectrogram::work (spectrogram.cpp:29-43) per frame: accumulate into the container of the current centre frequency,
// then send() — which publishes through DataController::pushSpectrogram (data_controller.cpp:44-57) once more than
// SPECTROGRAM_SEND_INTERVAL has passed on the injected clock (ref_set_time) since the container's last send.
namespace {
struct RefSpectrogram {
  Mqtt mqtt;
  DataController controller;
  Frequency frequency;
  Spectrogram block;
  RefSpectrogram(int item_size, Frequency sample_rate, Frequency f)
      : mqtt(), controller(mqtt, "ref"), frequency(f), block(item_size, sample_rate, controller, [this]() { return frequency; }) {}
};
}  // namespace

void* ref_spectrogram_create(int item_size, int sample_rate, int frequency) {
  ensureLogger();
  auto* g = new RefSpectrogram(item_size, sample_rate, frequency);
  return g;
}
void ref_spectrogram_destroy(void* h) { delete static_cast<RefSpectrogram*>(h); }
int ref_spectrogram_size(void* h) { return static_cast<RefSpectrogram*>(h)->block.m_outputSize; }
// Retune. The reference's Container constructor leaves m_counter uninitialised (spectrogram.cpp:9; whatever the heap
// holds): the shim creates the container of a new centre frequency itself — exactly as work() would, spectrogram.cpp:34-37 —
// and zeroes that one field, so that the comparison is deterministic.
void ref_spectrogram_set_frequency(void* h, int frequency) {
  auto* g = static_cast<RefSpectrogram*>(h);
  g->frequency = frequency;
  auto it = g->block.m_containers.find(frequency);
  if (it == g->block.m_containers.end()) {
    it = g->block.m_containers.emplace(frequency, g->block.m_outputSize).first;
    it->second.m_counter = 0;
  }
}
// nframes PSD rows through Spectrogram::work at the current injected time; returns how many payloads are waiting
int ref_spectrogram_work(void* h, const float* psd_rows, int nframes) {
  auto* g = static_cast<RefSpectrogram*>(h);
  ref_spectrogram_set_frequency(h, g->frequency);
  gr_vector_const_void_star in{psd_rows};
  gr_vector_void_star out{};
  g->block.work(nframes, in, out);
  return (int)g->mqtt.m_sent.size();
}
// container of the current centre frequency: m_sum (m_outputSize floats) and m_counter
int ref_spectrogram_container(void* h, float* sum_out) {
  auto* g = static_cast<RefSpectrogram*>(h);
  const auto it = g->block.m_containers.find(g->frequency);
  if (it == g->block.m_containers.end()) return -1;
  if (sum_out) std::memcpy(sum_out, it->second.m_sum.data(), sizeof(float) * it->second.m_sum.size());
  return it->second.m_counter;
}
// oldest published payload (bytes as handed to Mqtt::publish); returns its size, 0 when none, -1 when cap is short
int ref_spectrogram_pop(void* h, uint8_t* out, int cap) {
  auto* g = static_cast<RefSpectrogram*>(h);
  if (g->mqtt.m_sent.empty()) return 0;
  const auto& p = g->mqtt.m_sent.front().second;
  if ((int)p.size() > cap) return -1;
  std::memcpy(out, p.data(), p.size());
  const int n = (int)p.size();
  g->mqtt.m_sent.pop_front();
  return n;
}
// DataController::pushTransmission (data_controller.cpp:27-42) for `size` complex int8 samples
int ref_transmission_payload(uint64_t time_ms, int frequency, int sample_rate, const int8_t* iq_pairs, int size, uint8_t* out, int cap) {
  ensureLogger();
  Mqtt mqtt;
  DataController controller(mqtt, "ref");
  controller.pushTransmission(std::chrono::milliseconds(time_ms), frequency, sample_rate, reinterpret_cast<const SimpleComplex*>(iq_pairs), size);
  const auto& p = mqtt.m_sent.front().second;
  if ((int)p.size() > cap) return -1;
  std::memcpy(out, p.data(), p.size());
  return (int)p.size();
}

void* ref_averager_create(int size, int group) { return new Averager(size, group); }
void ref_averager_destroy(void* a) { delete static_cast<Averager*>(a); }
void ref_averager_push(void* a, const float* data) { static_cast<Averager*>(a)->push(data); }
void ref_averager_reset(void* a) { static_cast<Averager*>(a)->reset(); }
void ref_averager_average(void* a, float* out) {
  const auto& v = static_cast<Averager*>(a)->average();
  std::memcpy(out, v.data(), sizeof(float) * v.size());
}
void ref_averager_row(void* a, int r, float* out) {
  const auto& row = static_cast<Averager*>(a)->data().at(r);
  std::memcpy(out, row.data(), sizeof(float) * row.size());
}

}  // extern "C"
