"""Latency and rate of the synchronous host-buffer entry point (ss_process: what the GNU Radio adapter's work() calls)
against the call size: pageable numpy buffers in, candidate lists (and optionally the PSD plane) out.
    python scripts/call_latency.py [--fft 8192] [--sizes 1 4 16 64 256 1024] [--psd]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtl_sdr_scanner_cpp_amd as pkg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fft", type=int, default=8192)
    ap.add_argument("--sizes", type=int, nargs="+", default=[1, 4, 16, 64, 256, 1024])
    ap.add_argument("--psd", action="store_true", help="also copy the PSD plane back (the Spectrogram block's input)")
    ap.add_argument("--calls", type=int, default=200)
    a = ap.parse_args()
    n = a.fft
    fs = 250 * n
    band = pkg.synth.SyntheticBand(n, seed=3, on_frame=130, off_frame=10**9)
    iq = band.frames_cf32(max(a.sizes) + 128)
    for size in a.sizes:
        eng = pkg.SpectrumEngine(fs, 145_000_000, fft_size=n, decim=1, learn_frames=100, max_batch=max(size, 128))
        eng.process(iq[:128], want=())  # learning + averager warm-up
        x = np.ascontiguousarray(iq[128:128 + size])
        want = ("psd",) if a.psd else ()
        for _ in range(5):
            eng.process(x, want=want)
        calls = max(20, min(a.calls, 200_000 // size))
        ts = []
        for _ in range(calls):
            t0 = time.perf_counter()
            eng.process(x, want=want)
            ts.append(time.perf_counter() - t0)
        ts = np.array(ts)
        med = float(np.median(ts))
        print(json.dumps({"fft": n, "frames_per_call": size, "psd_out": a.psd, "median_us": round(med * 1e6, 1),
                          "p95_us": round(float(np.percentile(ts, 95)) * 1e6, 1), "MS_per_s": round(size * n / med / 1e6, 1),
                          "real_time_factor_at_2.048MSps_D5": round(size * n * 5 / med / 2.048e6, 1)}))
        del eng


if __name__ == "__main__":
    main()
