"""host/recorder_bank.h — the reference's recorders of one device (Recorder x recordersCount, SdrDevice::updateRecordings,
stream_to_vector + Buffer, flush -> pushTransmission) on top of the GPU channeliser — compiled into a C++ program, driven by a
scripted session, and compared record by record with a Python model that follows the reference's code (sources/radio/
recorder.cpp:58-98, sources/radio/sdr_device.cpp:82-144, sources/radio/blocks/buffer.h:23-58) over the same library calls."""
import json
import os
import subprocess
import zlib

import numpy as np
import pytest

import rtl_sdr_scanner_cpp_amd as pkg
from rtl_sdr_scanner_cpp_amd.channelizer import Channelizer

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MAIN = r"""
#include <recorder_bank.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

static unsigned crc32_of(const int8_t* p, size_t n) {  // zlib's CRC-32, bitwise
  unsigned c = 0xffffffffu;
  for (size_t i = 0; i < n; ++i) {
    c ^= (unsigned char)p[i];
    for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0xedb88320u & (0u - (c & 1u)));
  }
  return ~c;
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const int fs = 1024000, bw = 16000, chunk = 40960, center = 145000000;
  FILE* fp = fopen(argv[1], "rb");
  const int nchunks = atoi(argv[2]);
  std::vector<float> iq((size_t)2 * chunk * nchunks);
  if (!fp || fread(iq.data(), sizeof(float), iq.size(), fp) != iq.size()) return 3;
  fclose(fp);
  printf("[");
  bool first = true;
  specscan::RecorderBank bank(fs, bw, 2, chunk, [&](int64_t t, int32_t f, int32_t rate, const int8_t* d, int n) {
    printf("%s[%lld,%d,%d,%d,%u]", first ? "" : ",", (long long)t, f, rate, n, crc32_of(d, (size_t)n * 2));
    first = false;
  });
  // the scripted session: per chunk (40 ms of stream) the list the tracker would hand over (shift, flush)
  for (int c = 0; c < nchunks; ++c) {
    const int64_t now = 1000 + 40 * c;
    std::vector<specscan::RecorderBank::ShiftFlush> want;
    if (c >= 2 && c < 40) want.push_back({100000, c % 5 == 0});
    if (c >= 10 && c < 30) want.push_back({-250000, c % 7 == 0});
    if (c >= 12 && c < 20) want.push_back({30000, true});       // no free recorder: ignored
    if (c >= 45) want.push_back({-250000, c % 3 == 0});          // a new recording on a slot that was used before
    bank.updateRecordings(want, center, now);
    bank.work(iq.data() + (size_t)2 * chunk * c, chunk, now);
  }
  printf("]\n");
  return 0;
}
"""


def _model(x, nchunks, chunk, fs, bw, center):
    """The same session in Python, following the reference's code over the same sc_* calls."""
    ch = Channelizer(fs, bw, channels=2, max_samples=chunk)
    raw = bw * 100 // 1000
    item = raw if raw % 4096 == 0 else (raw // 4096 + 1) * 4096  # roundUp(bw * 100 ms / 1000, 4096), recorder.cpp:35
    IDLE = 2**31 - 1
    slots = [dict(rec=False, freq=IDLE, shift=IDLE, pending=np.zeros((0, 2), np.int8), items=[], times=[]) for _ in range(2)]
    out = []
    for c in range(nchunks):
        now = 1000 + 40 * c
        want = []
        if 2 <= c < 40:
            want.append((100000, c % 5 == 0))
        if 10 <= c < 30:
            want.append((-250000, c % 7 == 0))
        if 12 <= c < 20:
            want.append((30000, True))
        if c >= 45:
            want.append((-250000, c % 3 == 0))
        shifts = [s for s, _ in want]
        for k, s in enumerate(slots):  # sdr_device.cpp:103-111
            if s["rec"] and s["shift"] not in shifts:
                s.update(rec=False, freq=IDLE, shift=IDLE, items=[], times=[])
                ch.stop(k)
        for shift, flush in want:  # sdr_device.cpp:113-136
            hit = [k for k, s in enumerate(slots) if s["shift"] == shift]
            if hit:
                s = slots[hit[0]]
                if flush:  # Recorder::flush -> Buffer::popSingleSample -> pushTransmission
                    for t, it in zip(s["times"], s["items"]):
                        out.append([t, s["freq"] + s["shift"], bw, item, zlib.crc32(it.tobytes())])
                    s["items"], s["times"] = [], []
            else:
                free = [k for k, s in enumerate(slots) if not s["rec"]]
                if free:
                    s = slots[free[0]]
                    s.update(rec=True, freq=center, shift=shift, items=[], times=[])
                    ch.start(free[0], shift)
        res = ch.process(x[c * chunk:(c + 1) * chunk], want_cf32=False)
        for k, s in enumerate(slots):
            if not s["rec"]:
                continue
            s["pending"] = np.concatenate([s["pending"], res[k][0]])
            while len(s["pending"]) >= item:
                s["items"].append(s["pending"][:item].copy())
                s["times"].append(now)
                s["pending"] = s["pending"][item:]
    return out, item


def test_recorder_bank_session(tmp_path):
    fs, bw, chunk, center, nchunks = 1_024_000, 16_000, 40_960, 145_000_000, 60
    rng = np.random.default_rng(8)
    n = chunk * nchunks
    t = np.arange(n) / fs
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.02
    for f0 in (100_000.0, -250_000.0, 30_000.0):
        x += 0.2 * np.exp(2j * np.pi * (f0 * t + 0.4 * np.sin(2 * np.pi * 900 * t)))
    x = x.astype(np.complex64)
    raw = tmp_path / "stream.cf32"
    x.tofile(raw)
    src = tmp_path / "bank_main.cpp"
    src.write_text(MAIN)
    exe = tmp_path / "bank_main"
    csrc = os.path.join(ROOT, "rtl-sdr-scanner-cpp_amd", "csrc")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "rtl-sdr-scanner-cpp_amd", "host"),
           "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe), "-L" + csrc, "-lspecscan", "-Wl,-rpath," + csrc,
           "-Wl,-rpath-link," + os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib"), "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe), str(raw), str(nchunks)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    got = json.loads(r.stdout.strip().splitlines()[-1])
    want, item = _model(x, nchunks, chunk, fs, bw, center)
    assert item == 4096 and len(want) > 5
    assert got == want
    freqs = {rec[1] for rec in got}
    assert freqs == {center + 100_000, center - 250_000}  # the third shift never got a recorder
