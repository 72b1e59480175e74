"""Probe: ONE band, batches alternating between two contexts on one GPU (own streams), each batch re-scanning a halo of the
previous one after a reset (frame-range sharding), all enqueued from one host thread."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtl_sdr_scanner_cpp_amd as pkg
from rtl_sdr_scanner_cpp_amd import dist

def main():
    dev = torch.device("cuda:0")
    n, nb = 8192, 1024
    cfg = dict(fft_size=n, sample_rate=250 * n, decim=1, in_format=0, learn_frames=100, seed=0)
    for halo, npipes in ((0, 1), (32, 2), (32, 3), (0, 2)):
        tot = nb + halo
        engs = [pkg.SpectrumEngine(250 * n, 140_000_000, fft_size=n, decim=1, learn_frames=100, max_batch=tot) for _ in range(npipes)]
        iq = dist.synthetic_batch(cfg, 0, tot)
        d_iq = torch.from_numpy(iq.view(np.float32)).to(dev)
        outs = [[dict(psd=torch.empty((tot, n), dtype=torch.float32, device=dev), off=torch.zeros(tot + 1, dtype=torch.int32, device=dev),
                      idx=torch.empty(tot * 1024, dtype=torch.int32, device=dev), avg=torch.empty(tot * 1024, dtype=torch.float32, device=dev))
                 for _ in range(2)] for _ in range(npipes)]
        def step(k):
            e = engs[k % npipes]; o = outs[k % npipes][(k // npipes) & 1]
            if halo: e.reset()
            e.process_device(d_iq, tot, psd=o["psd"], cand_off=o["off"], cand_idx=o["idx"], cand_avg=o["avg"])
        for k in range(20 * npipes): step(k)
        for e in engs: e.sync()
        steps = 400
        t0 = time.perf_counter()
        for k in range(steps): step(k)
        t1 = time.perf_counter()
        for e in engs: e.sync()
        dt = time.perf_counter() - t0
        print(json.dumps({"pipes": npipes, "halo": halo, "us_per_step": round(dt / steps * 1e6, 2), "GS_new": round(steps * nb * n / dt / 1e9, 1), "enqueue_us": round((t1 - t0) / steps * 1e6, 2)}))
        del engs
main()
