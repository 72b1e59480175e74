#!/usr/bin/env python
"""Dispatch orders of k_scan_step side by side IN ONE PROCESS, on the same device buffers: engines are created one after the
other from libspecscan_diag.so with SS_STEP_ORDER set, each runs `chunks` x `steps` steps of the headline workload and reports
microseconds per step per chunk — so that the spread between runs (allocation placement, clocks) is visible next to the
spread between orders. Usage: python scripts/step_order_lab.py [--rounds R] order [order ...]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("orders", nargs="+")
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--chunks", type=int, default=4)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--frames", type=int, default=1024)
    ap.add_argument("--realloc", action="store_true", help="new device buffers (behind a dummy allocation of random size) for every engine")
    ap.add_argument("--env", action="append", default=[], help="KEY=VALUE set for every engine")
    ap.add_argument("--lib", action="append", default=[], help="NAME=path/to/libspecscan_diag.so: alternate engines between these builds (A/B of two source states in one process)")
    args = ap.parse_args()
    import torch
    import rtl_sdr_scanner_cpp_amd as pkg
    from rtl_sdr_scanner_cpp_amd import dist
    pkg.engine.use_diag_library(True)
    for kv in args.env:
        k, v = kv.split("=", 1)
        os.environ[k] = v
    dev = torch.device("cuda:0")
    n, nb, fs = 8192, args.frames, 2_048_000
    cfg = dict(fft_size=n, sample_rate=fs, decim=1, in_format=0, learn_frames=100, seed=0)
    gen = dist.synthetic_stream(cfg, 0)
    host = [gen(nb) for _ in range(3)]
    rng = np.random.default_rng(1)
    keep = []

    def buffers():
        if args.realloc:
            keep.append(torch.empty(int(rng.integers(1, 64)) << 20, dtype=torch.uint8, device=dev))
        base = [torch.from_numpy(h.view(np.float32)).to(dev) for h in host]
        d_iq = [base[k % 3] if k < 3 else torch.roll(base[k % 3], shifts=37 * k, dims=0).contiguous() for k in range(7)]
        outs = [dict(psd=torch.empty((nb, n), dtype=torch.float32, device=dev), off=torch.zeros(nb + 1, dtype=torch.int32, device=dev),
                     idx=torch.empty(nb * 1024, dtype=torch.int32, device=dev), avg=torch.empty(nb * 1024, dtype=torch.float32, device=dev)) for _ in range(7)]
        return d_iq, outs

    d_iq, outs = buffers()
    print(f"# {nb} frames x {n}, {args.chunks} chunks of {args.steps} steps per engine; us per step", flush=True)
    libs = [kv.split("=", 1) for kv in args.lib] or [("", None)]
    for rnd in range(args.rounds):
      for lib_name, lib_path in libs:
        if lib_path:
            pkg.engine.LIB_DIAG = os.path.abspath(lib_path)
        for order in args.orders:
            os.environ["SS_STEP_ORDER"] = order
            if args.realloc:
                d_iq, outs = buffers()
            eng = pkg.SpectrumEngine(fs, 140_000_000, fft_size=n, decim=1, learn_frames=100, max_batch=nb)
            k = 0
            for _ in range(12):  # learning + warm-up
                o = outs[k % 7]
                eng.process_device(d_iq[k % 7], nb, psd=o["psd"], cand_off=o["off"], cand_idx=o["idx"], cand_avg=o["avg"])
                k += 1
            eng.sync()
            res = []
            for _ in range(args.chunks):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    o = outs[k % 7]
                    eng.process_device(d_iq[k % 7], nb, psd=o["psd"], cand_off=o["off"], cand_idx=o["idx"], cand_avg=o["avg"])
                    k += 1
                eng.sync()
                res.append((time.perf_counter() - t0) / args.steps * 1e6)
            ncand = int(outs[(k - 1) % 7]["off"][-1].item())
            print(f"round {rnd} {lib_name:6s} {order:30s} " + " ".join(f"{r:6.2f}" for r in res) + f"   cand {ncand}", flush=True)
            eng.close()


if __name__ == "__main__":
    main()
