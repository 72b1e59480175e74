// raw_file.cpp — C entry points of raw_file.h (libspecscan_host.so). No GPU, no GNU Radio.
#include "raw_file.h"

using specscan::RawFileInfo;
using specscan::RawFileSink;
using specscan::RawIqReader;
using specscan::RawKind;

extern "C" {

int srf_make_name(const char* label, const char* extension, int32_t frequency, int32_t sample_rate, int year, int month, int day, int hour,
                  int minute, int second, char* out, int out_cap) {
  if (!label || !extension || !out || out_cap <= 0) return -1;
  std::tm tm{};
  tm.tm_year = year - 1900;
  tm.tm_mon = month - 1;
  tm.tm_mday = day;
  tm.tm_hour = hour;
  tm.tm_min = minute;
  tm.tm_sec = second;
  const std::string name = specscan::makeRawFileName(label, extension, frequency, sample_rate, tm);
  if ((int)name.size() + 1 > out_cap) return -2;
  memcpy(out, name.c_str(), name.size() + 1);
  return (int)name.size();
}

int srf_parse_name(const char* path, int32_t* frequency, int32_t* sample_rate, int* kind, int* ymdhms, char* label, char* extension) {
  if (!path) return -1;
  RawFileInfo info;
  try {
    if (!specscan::parseRawFileName(path, &info)) return -2;
  } catch (...) {
    return -2;
  }
  if (frequency) *frequency = info.frequency;
  if (sample_rate) *sample_rate = info.sample_rate;
  if (kind) *kind = (int)info.kind;
  if (ymdhms) {
    const int v[6] = {info.year, info.month, info.day, info.hour, info.minute, info.second};
    memcpy(ymdhms, v, sizeof(v));
  }
  if (label) snprintf(label, 64, "%s", info.label.c_str());
  if (extension) snprintf(extension, 64, "%s", info.extension.c_str());
  return 0;
}

void* srf_sink_create(int64_t item_bytes) { return item_bytes > 0 ? new RawFileSink((size_t)item_bytes) : nullptr; }
void srf_sink_destroy(void* sink) { delete static_cast<RawFileSink*>(sink); }
int srf_sink_start(void* sink, const char* filename) {
  if (!sink || !filename) return -1;
  static_cast<RawFileSink*>(sink)->startRecording(filename);
  return 0;
}
int srf_sink_stop(void* sink) {
  if (!sink) return -1;
  static_cast<RawFileSink*>(sink)->stopRecording();
  return 0;
}
int srf_sink_work(void* sink, const void* items, int nitems) {
  if (!sink || (!items && nitems > 0)) return -1;
  try {
    return static_cast<RawFileSink*>(sink)->work(items, nitems);
  } catch (const std::exception&) {
    return -3;  // FileSink::save throws std::runtime_error on open/write failure (file_sink.h:65-66,75-76)
  }
}

void* srf_reader_open(const char* path, int kind, int fft_size, int decim) {
  if (!path || fft_size <= 0 || kind < 0 || kind > 2) return nullptr;
  try {
    return new RawIqReader(path, (RawKind)kind, fft_size, decim);
  } catch (const std::exception&) {
    return nullptr;
  }
}
void srf_reader_close(void* reader) { delete static_cast<RawIqReader*>(reader); }
int64_t srf_reader_items(void* reader) { return reader ? static_cast<RawIqReader*>(reader)->items() : -1; }
int srf_reader_read_frames(void* reader, void* out, int max_frames) {
  if (!reader || !out || max_frames < 0) return -1;
  try {
    return static_cast<RawIqReader*>(reader)->readFrames(out, max_frames);
  } catch (const std::exception&) {
    return -3;
  }
}
int srf_reader_read_frames_parallel(void* reader, void* out, int max_frames, int threads) {
  if (!reader || !out || max_frames < 0) return -1;
  try {
    return static_cast<RawIqReader*>(reader)->readFramesParallel(out, max_frames, threads);
  } catch (const std::exception&) {
    return -3;
  }
}
}
