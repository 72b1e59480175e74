// raw_file.h — raw dump files of the reference, as a replay front end for the engine.
//
// The reference can dump what its SDR source produced and what its PSD block computed
// (DEBUG_SAVE_FULL_RAW_IQ / DEBUG_SAVE_FULL_POWER, sources/config.h:11-12):
//   ./full_YYYYMMDD_HHMMSS_<centre Hz>_<sample rate>_fc.raw      interleaved float re,im   (FileSink<gr_complex>(1))
//   ./full_YYYYMMDD_HHMMSS_<centre Hz>_<sample rate>_power.raw   N floats per frame, dB    (FileSink<float>(fftSize))
// wired at sources/radio/sdr_device.cpp:173-181, named by getRawFileName (sources/utils/radio_utils.cpp:78-84),
// restarted on every retune (sdr_device.cpp:64-65,75-76), written by FileSink::save (sources/radio/blocks/file_sink.h:61-84).
// Its own tooling reads them back with scripts/converter.py:30-53, which also accepts `.cs8` files (interleaved int8,
// value / 127.5) and takes centre frequency and sample rate from fields 3 and 4 of the name split at [._]
// (converter.py:58-59).
//
// This header gives a C++ host the same three things with no GNU Radio: the name (build and parse), a sink with the
// FileSink start/stop/save behaviour, and a reader that re-frames a dump into the items the engine's boundary takes
// (N*D samples per item, of which the Decimator keeps the first N: sources/radio/blocks/decimator.h:11-22).
// Header-only; raw_file.cpp adds the C entry points tests and other hosts bind.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include <unistd.h>

namespace specscan {

enum class RawKind { CF32 = 0, CS8 = 1, CU8 = 2, F32 = 3 };

struct RawFileInfo {
  std::string label;      // "full" or "recording"
  std::string extension;  // "fc" / "power" (the token before ".raw"), or the file suffix for foreign files ("cs8")
  int32_t frequency = 0;  // centre frequency, Hz
  int32_t sample_rate = 0;
  int year = 0, month = 0, day = 0, hour = 0, minute = 0, second = 0;
  RawKind kind = RawKind::CF32;
};

// getRawFileName (radio_utils.cpp:78-84) with the clock handed in: the reference takes time(nullptr) + localtime.
inline std::string makeRawFileName(const char* label, const char* extension, int32_t frequency, int32_t sample_rate, const std::tm& tm) {
  char buf[1024];
  snprintf(buf, sizeof(buf), "./%s_%04d%02d%02d_%02d%02d%02d_%d_%d_%s.raw", label, tm.tm_year + 1900, tm.tm_mon + 1, tm.tm_mday, tm.tm_hour,
           tm.tm_min, tm.tm_sec, frequency, sample_rate, extension);
  return buf;
}

inline std::string makeRawFileName(const char* label, const char* extension, int32_t frequency, int32_t sample_rate) {
  const time_t now = time(nullptr);
  std::tm tm{};
  localtime_r(&now, &tm);
  return makeRawFileName(label, extension, frequency, sample_rate, tm);
}

// The inverse. Accepts what getRawFileName writes and what converter.py accepts: the base name is split at '_' and
// '.', fields are <label> <date> <time> <frequency> <rate> <extension...>; a `cs8` / `cu8` suffix or name ending
// selects int8 IQ (converter.py:32-35), `power` selects float rows (converter.py:42-51 default), anything else cf32.
inline bool parseRawFileName(const std::string& path, RawFileInfo* out) {
  const size_t slash = path.find_last_of('/');
  const std::string base = slash == std::string::npos ? path : path.substr(slash + 1);
  std::vector<std::string> f;
  size_t start = 0;
  for (size_t i = 0; i <= base.size(); ++i) {
    if (i == base.size() || base[i] == '_' || base[i] == '.') {
      f.push_back(base.substr(start, i - start));
      start = i + 1;
    }
  }
  if (f.size() < 6) return false;
  const auto all_digits = [](const std::string& s, size_t len) {
    if (s.size() != len) return false;
    for (char ch : s)
      if (ch < '0' || ch > '9') return false;
    return true;
  };
  if (!all_digits(f[1], 8) || !all_digits(f[2], 6)) return false;
  RawFileInfo info;
  info.label = f[0];
  info.year = std::stoi(f[1].substr(0, 4));
  info.month = std::stoi(f[1].substr(4, 2));
  info.day = std::stoi(f[1].substr(6, 2));
  info.hour = std::stoi(f[2].substr(0, 2));
  info.minute = std::stoi(f[2].substr(2, 2));
  info.second = std::stoi(f[2].substr(4, 2));
  char* end = nullptr;
  const long freq = strtol(f[3].c_str(), &end, 10);
  if (*end != '\0' || f[3].empty()) return false;
  const long rate = strtol(f[4].c_str(), &end, 10);
  if (*end != '\0' || f[4].empty() || rate <= 0) return false;
  info.frequency = (int32_t)freq;
  info.sample_rate = (int32_t)rate;
  info.extension = f[5];
  const std::string& last = f.back();
  const auto has = [&](const char* tok) {
    for (size_t i = 5; i < f.size(); ++i)
      if (f[i] == tok) return true;
    return false;
  };
  if (last == "cs8" || has("cs8")) info.kind = RawKind::CS8;
  else if (last == "cu8" || has("cu8")) info.kind = RawKind::CU8;
  else if (has("power")) info.kind = RawKind::F32;
  else info.kind = RawKind::CF32;
  *out = info;
  return true;
}

inline size_t rawBytesPerValue(RawKind kind) {
  switch (kind) {
    case RawKind::CF32: return 8;  // one complex sample
    case RawKind::CS8:
    case RawKind::CU8: return 2;
    default: return 4;  // one float
  }
}

// FileSink<T>(itemSize, flushable=false) (file_sink.h:12-96): records only between start and stop, opens the file
// lazily at the first save (a recording that never sees data leaves no file), appends items, throws on open/write errors.
class RawFileSink {
 public:
  explicit RawFileSink(size_t item_bytes) : m_itemBytes(item_bytes) {}
  ~RawFileSink() { stopRecording(); }
  RawFileSink(const RawFileSink&) = delete;
  RawFileSink& operator=(const RawFileSink&) = delete;

  bool isRecording() const { return m_isRecording; }
  void startRecording(const std::string& filename) {
    stopRecording();
    m_filename = filename;
    m_isRecording = true;
  }
  void stopRecording() {
    if (m_file) {
      fclose(m_file);
      m_file = nullptr;
    }
    m_filename.clear();
    m_isRecording = false;
  }
  // work(): items are dropped unless recording (file_sink.h:19-29)
  int work(const void* items, int nitems) {
    if (m_isRecording) save(items, nitems);
    return nitems;
  }

 private:
  // Append whole items; the file is opened on the first write of a recording.
  void save(const void* data, int nitems) {
    if (!m_file && !(m_file = fopen(m_filename.c_str(), "wb"))) throw std::runtime_error("cannot create " + m_filename);
    const auto* bytes = static_cast<const unsigned char*>(data);
    for (size_t done = 0, total = static_cast<size_t>(nitems); done < total;) {
      const size_t n = fwrite(bytes + done * m_itemBytes, m_itemBytes, total - done, m_file);
      done += n;
      if (n == 0) {  // short write: a device error is fatal, anything else ends the append quietly (as the reference's sink does)
        if (ferror(m_file)) throw std::runtime_error("cannot append to " + m_filename);
        return;
      }
    }
  }
  const size_t m_itemBytes;
  FILE* m_file = nullptr;
  bool m_isRecording = false;
  std::string m_filename;
};

// Reads a dump back as the items the chain's front takes (stream_to_vector(N*D), sdr_device.cpp:161): item i is the
// samples [i*N*D, (i+1)*N*D); a trailing partial item is dropped (converter.py:37-38 does the same for its rows).
// readFrames hands out only the first N samples of each item — the Decimator's output — so the D-1 unused parts of
// an item are skipped in the file, never read.
class RawIqReader {
 public:
  RawIqReader(const std::string& path, RawKind kind, int fft_size, int decim) : m_kind(kind), m_n(fft_size), m_decim(decim < 1 ? 1 : decim) {
    if (kind == RawKind::F32) throw std::runtime_error("not an IQ file: " + path);
    m_file = fopen(path.c_str(), "rb");
    if (!m_file) throw std::runtime_error("open file failed: " + path);
    if (fseeko(m_file, 0, SEEK_END) != 0) throw std::runtime_error("seek failed: " + path);
    const int64_t bytes = (int64_t)ftello(m_file);
    rewind(m_file);
    m_items = bytes / (int64_t)(rawBytesPerValue(kind) * (size_t)m_n * (size_t)m_decim);
  }
  ~RawIqReader() {
    if (m_file) fclose(m_file);
  }
  RawIqReader(const RawIqReader&) = delete;
  RawIqReader& operator=(const RawIqReader&) = delete;

  int64_t items() const { return m_items; }
  int64_t position() const { return m_pos; }
  size_t frameBytes() const { return rawBytesPerValue(m_kind) * (size_t)m_n; }

  // The same with the copy out of the page cache spread over `threads` readers (pread on disjoint ranges): one
  // thread moves ~20 GB/s, less than the PCIe link behind it takes. Adjacent frames (decim 1) only; falls back otherwise.
  int readFramesParallel(void* out, int max_frames, int threads) {
    const int64_t want = std::min<int64_t>(max_frames, m_items - m_pos);
    const size_t fb = frameBytes();
    if (m_decim != 1 || threads <= 1 || want * (int64_t)fb < (int64_t)(8u << 20)) return readFrames(out, max_frames);
    const int fd = fileno(m_file);
    const int64_t base = m_pos * (int64_t)fb, total = want * (int64_t)fb;
    std::vector<std::thread> pool;
    std::vector<int> ok((size_t)threads, 1);
    for (int k = 0; k < threads; ++k) {
      const int64_t lo = total * k / threads, hi = total * (k + 1) / threads;
      pool.emplace_back([=, &ok] {
        int64_t done = lo;
        while (done < hi) {
          const ssize_t got = pread(fd, static_cast<char*>(out) + done, (size_t)(hi - done), (off_t)(base + done));
          if (got <= 0) {
            ok[(size_t)k] = 0;
            return;
          }
          done += got;
        }
      });
    }
    for (auto& t : pool) t.join();
    for (int v : ok)
      if (!v) throw std::runtime_error("read failed");
    m_pos += want;
    if (fseeko(m_file, (off_t)(m_pos * (int64_t)fb), SEEK_SET) != 0) throw std::runtime_error("seek failed");
    return (int)want;
  }

  // Up to max_frames frames of N samples, contiguous in `out`; returns how many (0 at the end of the file).
  int readFrames(void* out, int max_frames) {
    int done = 0;
    char* p = static_cast<char*>(out);
    const size_t fb = frameBytes();
    while (done < max_frames && m_pos < m_items) {
      if (m_decim == 1) {  // frames are adjacent: one read for the whole run
        const int64_t want = std::min<int64_t>(max_frames - done, m_items - m_pos);
        const size_t got = fread(p, fb, (size_t)want, m_file);
        if (got == 0) throw std::runtime_error("read failed");
        done += (int)got;
        m_pos += (int64_t)got;
        p += got * fb;
      } else {
        if (fread(p, fb, 1, m_file) != 1) throw std::runtime_error("read failed");
        if (fseeko(m_file, (off_t)(fb * (size_t)(m_decim - 1)), SEEK_CUR) != 0) throw std::runtime_error("seek failed");
        ++done;
        ++m_pos;
        p += fb;
      }
    }
    return done;
  }

 private:
  const RawKind m_kind;
  const int m_n, m_decim;
  FILE* m_file = nullptr;
  int64_t m_items = 0, m_pos = 0;
};

}  // namespace specscan

extern "C" {
// C binding (libspecscan_host.so). Strings are caller-owned buffers; returns are 0 / counts on success, <0 on error.
int srf_make_name(const char* label, const char* extension, int32_t frequency, int32_t sample_rate, int year, int month, int day, int hour,
                  int minute, int second, char* out, int out_cap);
// kind: 0 cf32, 1 cs8, 2 cu8, 3 f32 rows. label/extension buffers of at least 64 bytes.
int srf_parse_name(const char* path, int32_t* frequency, int32_t* sample_rate, int* kind, int* ymdhms /*6 ints*/, char* label, char* extension);
void* srf_sink_create(int64_t item_bytes);
void srf_sink_destroy(void* sink);
int srf_sink_start(void* sink, const char* filename);
int srf_sink_stop(void* sink);
int srf_sink_work(void* sink, const void* items, int nitems);
void* srf_reader_open(const char* path, int kind, int fft_size, int decim);
void srf_reader_close(void* reader);
int64_t srf_reader_items(void* reader);
int srf_reader_read_frames(void* reader, void* out, int max_frames);
int srf_reader_read_frames_parallel(void* reader, void* out, int max_frames, int threads);
}
