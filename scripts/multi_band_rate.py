"""Several bands (contexts) on ONE GPU, each driven from its own host thread on its own stream — the reference's
"several devices in one process" (sources/main.cpp:50-59) mapped onto one MI355X. Kernels of different contexts have no
dependencies on each other, so the FFT of one band can run next to the back end of another.
    python scripts/multi_band_rate.py [--bands 1 2 3 4] [--frames 1024] [--steps 300]"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtl_sdr_scanner_cpp_amd as pkg  # noqa: E402
from rtl_sdr_scanner_cpp_amd import dist  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bands", type=int, nargs="+", default=[1, 2, 3, 4])
    ap.add_argument("--frames", type=int, default=1024)
    ap.add_argument("--fft", type=int, default=8192)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--cs8", action="store_true", help="int8 IQ (BASELINE config 3's format)")
    ap.add_argument("--detect", action="store_true", help="detect mode: no dB plane handed out")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    n, nb = a.fft, a.frames
    cfg = dict(fft_size=n, sample_rate=250 * n, decim=1, in_format=0, learn_frames=100, seed=0)
    for nbands in a.bands:
        engines, inputs, outs = [], [], []
        for b in range(nbands):
            eng = pkg.SpectrumEngine(250 * n, 140_000_000 + 2_000_000 * b, fft_size=n, decim=1, learn_frames=min(100, nb), max_batch=nb,
                                     in_format=pkg.abi.SS_FMT_CS8 if a.cs8 else pkg.abi.SS_FMT_CF32)
            iq = dist.synthetic_batch(cfg, b, nb)
            engines.append(eng)
            if a.cs8:
                v = iq.view(np.float32).reshape(nb, n, 2)
                inputs.append(torch.from_numpy(np.clip(np.rint(v * 127.0 / max(1e-9, float(np.abs(v).max()))), -127, 127).astype(np.int8)).to(dev))
            else:
                inputs.append(torch.from_numpy(iq.view(np.float32)).to(dev))
            outs.append([dict(psd=None if a.detect else torch.empty((nb, n), dtype=torch.float32, device=dev), off=torch.zeros(nb + 1, dtype=torch.int32, device=dev),
                              idx=torch.empty(nb * 1024, dtype=torch.int32, device=dev), avg=torch.empty(nb * 1024, dtype=torch.float32, device=dev))
                         for _ in range(2)])
        torch.cuda.synchronize()

        def run(b, steps):
            for k in range(steps):
                o = outs[b][k & 1]
                engines[b].process_device(inputs[b], nb, psd=o["psd"], cand_off=o["off"], cand_idx=o["idx"], cand_avg=o["avg"])
            engines[b].sync()

        for b in range(nbands):
            run(b, 20)
        start = threading.Barrier(nbands + 1)

        def worker(b):
            start.wait()
            run(b, a.steps)

        threads = [threading.Thread(target=worker, args=(b,)) for b in range(nbands)]
        for t in threads:
            t.start()
        start.wait()
        t0 = time.perf_counter()
        for t in threads:
            t.join()
        dt = time.perf_counter() - t0
        print(json.dumps({"bands_on_one_gpu": nbands, "fft": n, "frames_per_batch": nb, "steps_per_band": a.steps,
                          "us_per_batch": round(dt / (a.steps * nbands) * 1e6, 2), "GS_per_s": round(nbands * a.steps * nb * n / dt / 1e9, 1),
                          "candidates": [int(outs[b][(a.steps - 1) & 1]["off"][-1].item()) for b in range(nbands)]}))
        del engines, inputs, outs


if __name__ == "__main__":
    main()
