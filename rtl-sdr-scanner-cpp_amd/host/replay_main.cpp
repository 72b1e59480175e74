// specscan_replay — a C++ host on the C ABI alone (include/specscan.h + host/raw_file.h), no Python, no GNU Radio:
// streams one of the reference's raw IQ dumps (`full_<date>_<time>_<centre>_<rate>_fc.raw`, or .cs8 / .cu8;
// sources/utils/radio_utils.cpp:78-84, scripts/converter.py:30-38) through the scan chain the way SdrDevice drives
// it (sources/radio/sdr_device.cpp:148-168), with the pipelined feed: a reader thread fills pinned slots while
// earlier batches cross PCIe and run. Optionally writes the raw PSD rows as the reference's DEBUG_SAVE_FULL_POWER
// dump would (`..._power.raw`, sdr_device.cpp:173-176).
//
//   specscan_replay <dump> [--fft N] [--decim D] [--batch B] [--depth K] [--learn-frames L] [--power-dir DIR]
// Prints one JSON line: frames, batches, candidates, seconds, MS/s (file + PCIe inclusive).
#include <specscan.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>

#include "raw_file.h"

namespace {

struct Options {
  std::string path, power_dir;
  int fft = 0, decim = 0, batch = 1024, depth = 3, learn_frames = -1;
};

bool parse_args(int argc, char** argv, Options* o) {
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    const auto next = [&](int* dst) {
      if (i + 1 >= argc) return false;
      *dst = atoi(argv[++i]);
      return true;
    };
    if (a == "--fft") {
      if (!next(&o->fft)) return false;
    } else if (a == "--decim") {
      if (!next(&o->decim)) return false;
    } else if (a == "--batch") {
      if (!next(&o->batch)) return false;
    } else if (a == "--depth") {
      if (!next(&o->depth)) return false;
    } else if (a == "--learn-frames") {
      if (!next(&o->learn_frames)) return false;
    } else if (a == "--power-dir") {
      if (i + 1 >= argc) return false;
      o->power_dir = argv[++i];
    } else if (!a.empty() && a[0] != '-' && o->path.empty()) {
      o->path = a;
    } else {
      return false;
    }
  }
  return !o->path.empty();
}

}  // namespace

int main(int argc, char** argv) {
  Options opt;
  if (!parse_args(argc, argv, &opt)) {
    fprintf(stderr, "usage: %s <dump> [--fft N] [--decim D] [--batch B] [--depth K] [--learn-frames L] [--power-dir DIR]\n", argv[0]);
    return 2;
  }
  specscan::RawFileInfo info;
  if (!specscan::parseRawFileName(opt.path, &info) || info.kind == specscan::RawKind::F32) {
    fprintf(stderr, "%s: not a raw IQ dump name (<label>_<date>_<time>_<centre>_<rate>_fc.raw | .cs8 | .cu8)\n", opt.path.c_str());
    return 2;
  }
  ss_config cfg;
  ss_default_config(&cfg, info.sample_rate, info.frequency);  // N = getFft(rate, 250), D = max(1, step/50): sdr_device.cpp:149-150
  if (opt.fft > 0) cfg.fft_size = opt.fft;
  if (opt.decim > 0) cfg.decim = opt.decim;
  if (opt.learn_frames >= 0) cfg.learn_frames = opt.learn_frames;
  cfg.max_batch = opt.batch;
  if (info.kind == specscan::RawKind::CS8 || info.kind == specscan::RawKind::CU8) {
    cfg.in_format = info.kind == specscan::RawKind::CS8 ? SS_FMT_CS8 : SS_FMT_CU8;
    cfg.int_scale = 1.0f / 127.5f;  // converter.py:33
  }
  ss_ctx* ctx = nullptr;
  if (ss_create(&cfg, &ctx) != SS_OK) {
    fprintf(stderr, "ss_create: %s\n", ss_last_error(nullptr));
    return 1;
  }
  const bool want_power = !opt.power_dir.empty();
  ss_feed* feed = nullptr;
  if (ss_feed_create(ctx, opt.depth, 1 << 20, want_power ? 1 : 0, &feed) != SS_OK) {
    fprintf(stderr, "ss_feed_create: %s\n", ss_last_error(ctx));
    ss_destroy(ctx);
    return 1;
  }
  int rc = 0;
  try {
    specscan::RawIqReader reader(opt.path, info.kind, cfg.fft_size, cfg.decim);
    specscan::RawFileSink power(sizeof(float) * (size_t)cfg.fft_size);  // FileSink<float>(fftSize, false)
    if (want_power) {
      std::tm tm{};
      tm.tm_year = info.year - 1900, tm.tm_mon = info.month - 1, tm.tm_mday = info.day;
      tm.tm_hour = info.hour, tm.tm_min = info.minute, tm.tm_sec = info.second;
      power.startRecording(opt.power_dir + "/" + specscan::makeRawFileName("full", "power", info.frequency, info.sample_rate, tm).substr(2));
    }
    // slot accounting between the reader thread and this one
    std::mutex mtx;
    std::condition_variable cv;
    int free_slots = opt.depth, submitted = 0;
    bool reader_done = false, failed = false;
    std::string error;

    const auto t0 = std::chrono::steady_clock::now();
    std::thread producer([&] {
      for (;;) {
        {
          std::unique_lock<std::mutex> lock(mtx);
          cv.wait(lock, [&] { return free_slots > 0 || failed; });
          if (failed) break;
          --free_slots;
        }
        void* frames = nullptr;
        int got = 0;
        try {
          if (ss_feed_acquire(feed, &frames) != SS_OK) throw std::runtime_error(ss_last_error(ctx));
          got = reader.readFramesParallel(frames, cfg.max_batch, 4);
          if (got > 0 && ss_feed_submit(feed, got, nullptr, reader.position() - got) != SS_OK) throw std::runtime_error(ss_last_error(ctx));
        } catch (const std::exception& e) {
          std::lock_guard<std::mutex> lock(mtx);
          failed = true;
          error = e.what();
        }
        std::lock_guard<std::mutex> lock(mtx);
        if (got > 0 && !failed) ++submitted;
        if (got == 0 || failed) {
          reader_done = true;
          cv.notify_all();
          break;
        }
        cv.notify_all();
      }
    });

    long long frames = 0, batches = 0, candidates = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lock(mtx);
        cv.wait(lock, [&] { return submitted > 0 || reader_done; });
        if (submitted == 0) break;
        --submitted;
      }
      ss_feed_result r;
      if (ss_feed_collect(feed, &r) != SS_OK) {
        std::lock_guard<std::mutex> lock(mtx);
        failed = true;
        error = ss_last_error(ctx);
        cv.notify_all();
        break;
      }
      if (want_power) {
        try {
          power.work(r.psd_db, r.nframes);
        } catch (const std::exception& e) {  // the producer thread is still joinable: report, wake it, fall through to the join
          std::lock_guard<std::mutex> lock(mtx);
          failed = true;
          error = e.what();
          cv.notify_all();
          break;
        }
      }
      frames += r.nframes;
      candidates += r.cand_off[r.nframes];
      ++batches;
      std::lock_guard<std::mutex> lock(mtx);
      ++free_slots;
      cv.notify_all();
    }
    producer.join();
    const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (failed) {
      fprintf(stderr, "replay failed: %s\n", error.c_str());
      rc = 1;
    } else {
      printf("{\"file\": \"%s\", \"fft_size\": %d, \"decim\": %d, \"items_in_file\": %lld, \"frames\": %lld, \"batches\": %lld, "
             "\"candidates\": %lld, \"seconds\": %.6f, \"msamples_per_sec\": %.1f}\n",
             opt.path.c_str(), cfg.fft_size, cfg.decim, (long long)reader.items(), frames, batches, candidates, seconds,
             seconds > 0 ? (double)frames * cfg.fft_size / seconds / 1e6 : 0.0);
    }
  } catch (const std::exception& e) {
    fprintf(stderr, "%s\n", e.what());
    rc = 1;
  }
  ss_feed_destroy(feed);
  ss_destroy(ctx);
  return rc;
}
