#!/usr/bin/env python
"""ss_process on 16 frames of 2^20 CF32 samples, eight calls: where does a call's time go? Run under `rocprofv3 --hip-trace --stats`
(scripts/r06/s8.sh) — or alone, for the wall time per call with small and with default candidate capacities."""
import sys
import time

import numpy as np
import torch

torch.cuda.init()  # (torch first, as in bench.py: initialised after the engine's context it finds no device)
sys.path.insert(0, ".")
import rtl_sdr_scanner_cpp_amd as pkg

n, fs, nb = 1 << 20, 61_440_000, 16
band = pkg.synth.SyntheticBand(n, seed=9, on_frame=40, off_frame=10_000)
learn = band.frames_cf32(32)
batch = band.frames_cf32(nb)
eng = pkg.SpectrumEngine(fs, 145_000_000, fft_size=n, decim=1, in_format=0, learn_frames=32, max_batch=32)
eng.process(learn, want=())
for cap in (None, 1 << 20):
    for _ in range(2):
        eng.process(batch, want=(), cand_cap=cap)
    t0 = time.perf_counter()
    for _ in range(4):
        eng.process(batch, want=(), cand_cap=cap)
    dt = (time.perf_counter() - t0) / 4
    print(f"cand_cap {cap}: {dt * 1e3:.2f} ms per call = {nb * n / dt / 1e6:.0f} MS/s")
eng.close()
if "--after-working-set" in sys.argv:
    # what bench.py's run() does before its drop-in lines: a 1.2 GB working set copied from pageable numpy arrays, then released
    dev = torch.device("cuda", 0)
    band8 = pkg.synth.SyntheticBand(8192, seed=1, on_frame=10, off_frame=10_000)
    held = [torch.from_numpy(band8.frames_cf32(1024).view(np.float32)).to(dev) for _ in range(4)]
    held += [torch.roll(held[k % 4], shifts=37 * k, dims=0).contiguous() for k in range(8)]
    outs = [torch.empty((1024, 8192), dtype=torch.float32, device=dev) for _ in range(12)]
    torch.cuda.synchronize()
    eng = pkg.SpectrumEngine(fs, 145_000_000, fft_size=n, decim=1, in_format=0, learn_frames=32, max_batch=32)
    eng.process(learn, want=())
    for _ in range(2):
        eng.process(batch, want=())
    t0 = time.perf_counter()
    for _ in range(4):
        eng.process(batch, want=())
    dt = (time.perf_counter() - t0) / 4
    print(f"with a 1.2 GB torch working set alive: {dt * 1e3:.2f} ms per call = {nb * n / dt / 1e6:.0f} MS/s")
    del held, outs
    torch.cuda.empty_cache()
    t0 = time.perf_counter()
    for _ in range(4):
        eng.process(batch, want=())
    dt = (time.perf_counter() - t0) / 4
    print(f"... released: {dt * 1e3:.2f} ms per call = {nb * n / dt / 1e6:.0f} MS/s")
    eng.close()

if "--bisect" in sys.argv:
    # bench.py's drop_in_lines measures 27-31 ms per call at the end of its process: which of the things that ran before does it?
    def rate(tag):
        e = pkg.SpectrumEngine(fs, 145_000_000, fft_size=n, decim=1, in_format=0, learn_frames=32, max_batch=32)
        e.process(learn, want=())
        for _ in range(2):
            e.process(batch, want=())
        t0 = time.perf_counter()
        for _ in range(4):
            e.process(batch, want=())
        dt = (time.perf_counter() - t0) / 4
        e.close()
        print(f"{tag}: {dt * 1e3:.2f} ms per call = {nb * n / dt / 1e6:.0f} MS/s", flush=True)
    rate("fresh")
    # (1) a 65536-point int8 context with a pinned feed ring, as the drop-in entry before this one
    b65 = pkg.synth.SyntheticBand(65536, seed=9, on_frame=40, off_frame=10_000)
    e3 = pkg.SpectrumEngine(20_000_000, 145_000_000, fft_size=65536, decim=1, in_format=1, learn_frames=32, max_batch=128)
    e3.process(b65.frames_cs8(32), want=())
    x = b65.frames_cs8(128)
    e3.process(x, want=())
    rate("after a 65536-point context's ss_process (context alive)")
    feed = e3.feed(depth=3, cand_cap=1 << 20)
    for _ in range(3):
        feed.acquire()[:128] = x
        feed.submit(128)
    for _ in range(3):
        feed.collect()
    rate("... with its feed ring alive")
    feed.close()
    e3.close()
    rate("... feed and context closed")
    # (2) an 8192-point deep-pipelined context that has run a few hundred device calls
    dev = torch.device("cuda", 0)
    b8 = pkg.synth.SyntheticBand(8192, seed=1, on_frame=10, off_frame=10_000)
    e8 = pkg.SpectrumEngine(2_048_000, 145_000_000, fft_size=8192, decim=1, max_batch=1024)
    d = torch.from_numpy(b8.frames_cf32(1024).view(np.float32)).to(dev)
    o = dict(psd=torch.empty((1024, 8192), dtype=torch.float32, device=dev), off=torch.zeros(1025, dtype=torch.int32, device=dev), idx=torch.empty(1 << 20, dtype=torch.int32, device=dev))
    for _ in range(300):
        e8.process_device(d, 1024, psd=o["psd"], cand_off=o["off"], cand_idx=o["idx"])
    e8.sync()
    rate("after 300 deep-pipelined 8192-point device calls (context alive)")
    e8.close()
    rate("... closed")
