"""Throughput of the recorder channeliser on HBM-resident input (sc_process_device), next to the VALU bound.
    python scripts/channelizer_rate.py [--fs 2048000] [--bw 32000] [--samples 8388608] [--slots 1 4 8]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtl_sdr_scanner_cpp_amd.channelizer import Channelizer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fs", type=int, default=2_048_000)
    ap.add_argument("--bw", type=int, default=32_000)
    ap.add_argument("--samples", type=int, default=1 << 23)
    ap.add_argument("--slots", type=int, nargs="+", default=[1, 4, 8])
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((a.samples, 2)) * 0.1).astype(np.float32)
    d_iq = torch.from_numpy(x).to(dev)
    for nslots in a.slots:
        ch = Channelizer(a.fs, a.bw, channels=nslots, max_samples=a.samples)
        for k in range(nslots):
            ch.start(k, int((k - nslots / 2) * 0.9 * a.fs / max(nslots, 2)))
        cap = ch.output_capacity(a.samples)
        d_i8 = torch.zeros((nslots, cap, 2), dtype=torch.int8, device=dev)
        for _ in range(3):
            ch.process_device(d_iq, a.samples, d_i8, None, cap)
        ch.sync()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            ch.process_device(d_iq, a.samples, d_i8, None, cap)
        ch.sync()
        dt = (time.perf_counter() - t0) / a.steps
        taps_per_in = sum(nt / d * np.prod([i2 / d2 for i2, d2, _ in ch.stages[:k]]) for k, (i, d, nt) in enumerate(ch.stages))
        flops = 4.0 * taps_per_in * a.samples * nslots  # 2 FMA per tap per input sample (real taps, complex data)
        print(json.dumps({"fs": a.fs, "bw": a.bw, "stages": ch.stages, "slots": nslots, "samples": a.samples, "ms_per_call": round(dt * 1e3, 4),
                          "input_GSps": round(a.samples / dt / 1e9, 2), "slot_GSps": round(a.samples * nslots / dt / 1e9, 2),
                          "fir_TFLOPs": round(flops / dt / 1e12, 2), "taps_per_input_sample": round(float(taps_per_in), 2)}))
        ch.close()


if __name__ == "__main__":
    main()
