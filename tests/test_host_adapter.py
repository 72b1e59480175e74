"""The C++ reference-side adapter (host/gpu_spectrum_block.h) must compile as C++17 against the block API it
targets. GNU Radio is not installed here, so the build-only stand-in under oracle/stubs provides
gr::sync_block; linking is not attempted (the adapter only calls the C ABI)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_adapter_compiles_against_the_block_api(tmp_path):
    src = tmp_path / "use_adapter.cpp"
    src.write_text('#include <gpu_spectrum_block.h>\n'
                   'int use(const ss_config& cfg) {\n'
                   '  GpuSpectrum block(cfg, [](int, const int32_t*, const float*, int) {});\n'
                   '  gr_vector_const_void_star in{nullptr};\n'
                   '  gr_vector_void_star out{nullptr};\n'
                   '  block.setFrequencyRange(144000000, 146000000);\n'
                   '  block.resetBuffers();\n'
                   '  return block.work(0, in, out);\n'
                   '}\n')
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-Wpedantic",
           "-I" + os.path.join(ROOT, "rtl-sdr-scanner-cpp_amd", "host"), "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "oracle", "stubs"), str(src)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


ADAPTER_MAIN = r"""
// Drives the reference-side adapter the way GNU Radio's scheduler drives a sync_block: work() calls on scheduler-owned
// buffers, a retune from another control path, the tracker consuming each frame's candidates.
#include <gpu_spectrum_block.h>
#include <signal_tracker.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  const char* path = argv[1];
  const int n = atoi(argv[2]), nframes = atoi(argv[3]);
  ss_config cfg;
  ss_default_config(&cfg, n * 250, 145000000);
  cfg.fft_size = n;
  cfg.decim = 1;
  cfg.learn_frames = 20;
  cfg.learn_ms = 0;
  cfg.max_batch = 16;
  cfg.flags |= SS_FLAG_SPECTROGRAM;
  std::vector<gr_complex> iq((size_t)n * nframes);
  FILE* fp = fopen(path, "rb");
  if (!fp || fread(iq.data(), sizeof(gr_complex), iq.size(), fp) != iq.size()) return 3;
  fclose(fp);
  long long total = 0;
  std::vector<int> per_frame;
  GpuSpectrum block(cfg, [&](int, const int32_t*, const float*, int count) {
    total += count;
    per_frame.push_back(count);
  });
  // Transmission's bookkeeping on top, with an injected clock: 40 ms per frame, read once per work() call
  int clock_pos = 0;
  block.setClock([&] { return (int64_t)(1000 + 40 * clock_pos); });
  specscan::TrackerConfig tc;
  tc.fft_size = n;
  tc.sample_rate = n * 250;
  tc.group_size = 128;  // ceil(32000 / 250)
  tc.min_time_ms = 200;
  tc.timeout_ms = 400;
  std::vector<std::vector<specscan::FrequencyFlush>> tx_per_frame;
  block.enableTracker(tc, [&](const std::vector<specscan::FrequencyFlush>& tx) { tx_per_frame.push_back(tx); });
  // Spectrogram::send's gate on the same clock; each row framed the way DataController::pushSpectrogram frames it
  std::vector<std::vector<uint8_t>> spectrogram_payloads;
  block.enableSpectrogram([&](int64_t t, int32_t frequency, int32_t rate, const int8_t* row, int size) {
    std::vector<uint8_t> payload((size_t)ss_spectrogram_payload((uint64_t)t, frequency, rate, row, size, nullptr, 0));
    ss_spectrogram_payload((uint64_t)t, frequency, rate, row, size, payload.data(), (int32_t)payload.size());
    spectrogram_payloads.push_back(payload);
  });
  std::vector<float> psd((size_t)n * 16);
  int pos = 0;
  const int sizes[] = {1, 16, 7, 3, 16, 16, 5};
  int k = 0;
  while (pos < nframes) {
    int want = sizes[k++ % 7];
    if (want > nframes - pos) want = nframes - pos;
    gr_vector_const_void_star in{iq.data() + (size_t)pos * n};
    gr_vector_void_star out{psd.data()};
    clock_pos = pos;
    const int produced = block.work(want, in, out);
    if (produced != want) {
      fprintf(stderr, "work produced %d of %d: %s\n", produced, want, block.lastError().c_str());
      return 4;
    }
    pos += produced;
  }
  printf("{\"frames\": %d, \"candidates\": %lld, \"last_psd0\": %.6f, \"per_frame\": [", (int)per_frame.size(), total, psd[0]);
  for (size_t i = 0; i < per_frame.size(); ++i) printf("%s%d", i ? "," : "", per_frame[i]);
  printf("], \"tx\": [");
  for (size_t i = 0; i < tx_per_frame.size(); ++i) {
    printf("%s[", i ? "," : "");
    for (size_t k = 0; k < tx_per_frame[i].size(); ++k) printf("%s[%d,%d]", k ? "," : "", tx_per_frame[i][k].shift_hz, (int)tx_per_frame[i][k].flush);
    printf("]");
  }
  printf("], \"spectrogram\": [");
  for (size_t i = 0; i < spectrogram_payloads.size(); ++i) {
    printf("%s\"", i ? "," : "");
    for (uint8_t b : spectrogram_payloads[i]) printf("%02x", b);
    printf("\"");
  }
  printf("]}\n");
  return 0;
}
"""


@pytest.mark.gpu
def test_adapter_runs_against_the_library(tmp_path):
    """The adapter block compiled against the block API (stand-in header) and LINKED with libspecscan.so, driven like the
    scheduler drives it: the candidates it hands to the callback are the ones the boundary reports for the same frames."""
    import json
    import numpy as np
    import rtl_sdr_scanner_cpp_amd as pkg
    n, nframes = 1024, 120
    band = pkg.synth.SyntheticBand(n, seed=21, on_frame=28, off_frame=110)
    iq = band.frames_cf32(nframes)
    raw = tmp_path / "iq.cf32"
    iq.tofile(raw)
    src = tmp_path / "adapter_main.cpp"
    src.write_text(ADAPTER_MAIN)
    exe = tmp_path / "adapter_main"
    csrc = os.path.join(ROOT, "rtl-sdr-scanner-cpp_amd", "csrc")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "rtl-sdr-scanner-cpp_amd", "host"),
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "oracle", "stubs"), str(src), os.path.join(ROOT, "rtl-sdr-scanner-cpp_amd", "host", "signal_tracker.cpp"), "-o", str(exe), "-L" + csrc, "-lspecscan",
           "-Wl,-rpath," + csrc, "-Wl,-rpath-link," + os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib"), "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe), str(raw), str(n), str(nframes)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    eng = pkg.SpectrumEngine(n * 250, 145_000_000, fft_size=n, decim=1, learn_frames=20, learn_ms=0, max_batch=16, flags=pkg.abi.SS_FLAG_SPECTROGRAM)
    eng2 = pkg.SpectrumEngine(n * 250, 145_000_000, fft_size=n, decim=1, learn_frames=20, learn_ms=0, max_batch=16)
    want, last, want_tx, want_rows, last_sent = [], None, [], [], None
    pos, k = 0, 0
    sizes = [1, 16, 7, 3, 16, 16, 5]
    trk = pkg.tracker.SignalTracker(n, n * 250, group_size=128, min_time_ms=200, timeout_ms=400)
    while pos < nframes:
        s_ = min(sizes[k % 7], nframes - pos)
        k += 1
        full = eng2.process(iq[pos:pos + s_], t_ms=np.full(s_, 1_700_000_000_000, np.int64))
        for tx, _sig in trk.process_batch(np.full(s_, 1000 + 40 * pos, np.int64), full["avg"], full["rel"], full["cand_off"], full["cand_idx"]):
            want_tx.append(np.asarray(tx).reshape(-1, 2).tolist())
        # the adapter stamps every frame with the wall clock (as the reference's blocks do); with learn_ms = 0 the very
        # first frame completes the learning whatever the clock says, so constant stamps reproduce it
        o = eng.process(iq[pos:pos + s_], t_ms=np.full(s_, 1_700_000_000_000, np.int64), want=("psd",))
        want.extend(np.diff(o["cand_off"]).tolist())
        last = o["psd"]
        now = 1000 + 40 * pos  # the injected clock, read once per work() call
        last_sent = now if last_sent is None else last_sent
        if last_sent + 1000 < now:  # Spectrogram::send, spectrogram.cpp:65
            row, _mean, cnt = eng.spectrogram_read()
            assert cnt > 0
            want_rows.append(pkg.engine.spectrogram_payload(now, 145_000_000, n * 250, row).hex())
            last_sent = now
        pos += s_
    assert rep["frames"] == nframes and rep["per_frame"] == want and rep["candidates"] == sum(want) > 500
    assert abs(rep["last_psd0"] - float(last[0, 0])) < 1e-4
    # and the list Notification::notify would receive, frame by frame: tuned shifts (Hz) and flush flags
    assert rep["tx"] == want_tx and sum(len(t) for t in want_tx) > 50 and any(f for t in want_tx for _, f in t)
    # the rows Spectrogram::send would publish on the same clock, framed by DataController::pushSpectrogram: 1 s apart, at batch boundaries
    assert rep["spectrogram"] == want_rows and len(want_rows) == 3
    assert len(bytes.fromhex(want_rows[0])) == 8 + 12 + 4 + 256  # getFft(256000, 1000) = 256 bins
