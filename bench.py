#!/usr/bin/env python
"""bench.py — IQ MSamples/s scanned by the spectral-scan hot path on MI355X.

One step = one pass of the whole chain (load + Hamming window + 8192-pt FFT + dB power, noise-relative,
21-frame x 21-bin averaging, threshold, candidate compaction) over one batch of 1024 synthetic frames
that is already resident in HBM: BASELINE.json configs[1] ("8192-pt FFT, 2.048 MS/s, 1024-frame batches
on 1x MI355X"). With --gpus N every rank scans its own band (weak scaling, no data-path collective; the
scan configuration is broadcast once from rank 0).

Prints ONE JSON line on rank 0. `roofline` is for the dominant kernel (fused FFT+PSD): algorithmic bytes
(8 B/sample CF32 in + 4 B/sample dB out = 12 B/sample) x samples per launch / the kernel's
mean device time, taken from start/stop events attached to each launch on the engine's own stream
during the timed region. `cpu_baseline` times the reference's own compiled sources (oracle/_ref) —
or the C oracle where _ref is absent — on the host cores, on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
ALGO_BYTES_PER_SAMPLE = 12.0  # CF32 in (8) + f32 dB out (4), SURVEY.md §8d power mode


def _cpu_worker(args):
    """One host core scanning its own band with the reference's code for `budget_s` seconds."""
    seed, n, fs, budget_s, use_ref, backend = args
    import numpy as np
    import rtl_sdr_scanner_cpp_amd as pkg
    from oracle import oracle as O
    band = pkg.synth.SyntheticBand(n, seed=seed, on_frame=130, off_frame=10_000)
    chunk = 64
    iq = band.frames_cf32(chunk)
    center = 145_000_000
    frames = 0
    if use_ref:
        O.ref().orc_set_fft_backend(backend)
        chain = O.RefChain(n, fs, center - fs // 2, center + fs // 2)
        t_ms = 0
    else:
        O.lib().orc_set_fft_backend(backend)
        chain = O.oracle_chain(fs, center, fft_size=n, decim=1, max_batch=chunk)
    t0 = time.perf_counter()
    while True:
        if use_ref:
            chain.process(iq, t_ms + 20 * np.arange(chunk))
            t_ms += 20 * chunk
        else:
            chain.process(iq, want=(), cand_cap=chunk * n)
        frames += chunk
        el = time.perf_counter() - t0
        if el >= budget_s:
            return frames, el


def usable_cores() -> int:
    """Host cores this process may really use: affinity mask, capped by the cgroup CPU quota."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(cores, 64))


def cpu_baseline(n: int, fs: int, budget_s: float = 12.0):
    os.environ.setdefault("MKL_NUM_THREADS", "1")  # one FFT thread per worker process, like fft_v's nthreads = 1
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    from oracle import oracle as O
    use_ref = O.have_ref()
    # the reference runs FFTW through GNU Radio; MKL's FFTW3 interface is the closest thing on this host
    backend = 2 if (O.ref() if use_ref else O.lib()).orc_set_fft_backend(2) == 0 else 0
    cores = usable_cores()
    ctx = mp.get_context("fork")
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(1000 + i, n, fs, budget_s, use_ref, backend) for i in range(cores)])
    frames = sum(r[0] for r in res)
    wall = max(r[1] for r in res)
    one = res[0][0] * n / res[0][1] / 1e6
    return {
        "value": round(frames * n / wall / 1e6, 3), "unit": "MS/s", "cores": cores,
        "kind": "reference" if use_ref else "port",
        "sample": (f"{frames} frames of {n} CF32 samples ({frames * n / 1e6:.0f} MS) in {wall:.1f} s: one independent band per core, "
                   f"full chain window+FFT+dB+noise+21x21 mean+threshold, "
                   f"{'reference .cpp files compiled in place (oracle/_ref)' if use_ref else 'C restatement (oracle/liboracle.so)'}, "
                   f"FFT via {'MKL FFTW3 interface' if backend == 2 else 'built-in radix-2'}; {one:.1f} MS/s per core"),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--frames", type=int, default=1024, help="frames per batch (BASELINE config 2: 1024)")
    ap.add_argument("--fft", type=int, default=8192)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fmt", default="cf32", choices=["cf32", "cs8", "cu8"], help="IQ sample format in HBM (the headline is cf32)")
    ap.add_argument("--spectrogram", action="store_true", help="also run the Spectrogram side branch (SS_FLAG_SPECTROGRAM) every batch")
    ap.add_argument("--no-psd-out", action="store_true", help="detect mode: the caller takes candidates only, no PSD plane is handed out")
    ap.add_argument("--planes", action="store_true", help="full mode: the rel and avg planes are handed out as well (20 B/sample)")
    ap.add_argument("--decim", type=int, default=1, help="frame decimation D: items of N*D samples, the first N of each are scanned (reference: 5 at 2.048 MS/s)")
    ap.add_argument("--lanes", type=int, default=1, help="ss_pipe with this many lanes (batches of the one band in flight side by side); 1 = one context")
    ap.add_argument("--single-buffer", action="store_true", help="one output set instead of two alternating ones")
    ap.add_argument("--time-every", type=int, default=8, help="attach start/stop events to every k-th launch of the FFT kernel")
    ap.add_argument("--no-kernel-timing", action="store_true", help="do not attach per-launch events to the FFT kernel (roofline omitted)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    import numpy as np
    import torch
    import rtl_sdr_scanner_cpp_amd as pkg
    from rtl_sdr_scanner_cpp_amd import dist

    # RCCL ("nccl") over xGMI in production; SS_DIST_BACKEND=gloo lets the multi-rank path be exercised on a box
    # with fewer GPUs than ranks (ranks then share devices, results are functional only)
    backend = os.environ.get("SS_DIST_BACKEND", "nccl")
    rank, local_rank, world = dist.init(backend)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    device_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")

    n, fs, nb = args.fft, 2_048_000 * (args.fft // 8192 if args.fft >= 8192 else 1), args.frames
    cfg0 = None
    if rank == 0:
        cfg0 = dict(fft_size=n, sample_rate=fs, decim=args.decim, in_format={"cf32": 0, "cs8": 1, "cu8": 2}[args.fmt], grouping_x=21, grouping_y=21,
                    start_level_mdB=8000, learn_frames=100, learn_ms=2000, max_batch=nb, band0_center=140_000_000,
                    band_spacing=2_000_000, n_bands=world, seed=0)
    cfg = dist.broadcast_config(cfg0, device=coll_dev)  # the only collective of the whole job (RCCL, < 1 KiB)
    band = dist.bands_for_rank(int(cfg["n_bands"]), rank, world)[0]

    eng_kw = dict(fft_size=int(cfg["fft_size"]), decim=int(cfg["decim"]), in_format=int(cfg["in_format"]), grouping_x=int(cfg["grouping_x"]),
                  grouping_y=int(cfg["grouping_y"]), start_level=cfg["start_level_mdB"] / 1000.0, learn_frames=int(cfg["learn_frames"]),
                  max_batch=nb, device_id=device_index, flags=pkg.abi.SS_FLAG_SPECTROGRAM if args.spectrogram else 0)
    if args.lanes > 1:
        if args.planes or args.spectrogram:
            raise SystemExit("--lanes: candidates and the PSD plane only")
        eng = pkg.engine.Pipe(int(cfg["sample_rate"]), dist.band_center(cfg, band), lanes=args.lanes, **eng_kw)
        args.no_kernel_timing = True  # per-launch events belong to one context; the lanes overlap each other's kernels
    else:
        eng = pkg.SpectrumEngine(int(cfg["sample_rate"]), dist.band_center(cfg, band), **eng_kw)
    iq = dist.synthetic_batch(cfg, band, nb)
    d_iq = torch.from_numpy(iq.view(np.float32) if iq.dtype == np.complex64 else iq).to(dev)
    # Outputs are double-buffered the way a streaming consumer would hold them: batch k writes set k & 1 while
    # the consumer still owns set (k - 1) & 1.
    cap = nb * 1024
    outs = [dict(psd=torch.empty((nb, n), dtype=torch.float32, device=dev), off=torch.zeros(nb + 1, dtype=torch.int32, device=dev),
                 idx=torch.empty(cap, dtype=torch.int32, device=dev), avg=torch.empty(cap, dtype=torch.float32, device=dev),
                 rel_plane=torch.empty((nb, n), dtype=torch.float32, device=dev) if args.planes else None,
                 avg_plane=torch.empty((nb, n), dtype=torch.float32, device=dev) if args.planes else None)
            for _ in range(1 if args.single_buffer else max(2, args.lanes + 1))]  # a lane's outputs stay its own until its batch is done
    torch.cuda.synchronize()
    counter = [0]

    def step():
        o = outs[counter[0] % len(outs)]
        counter[0] += 1
        if args.lanes > 1:
            eng.process_device(d_iq, nb, psd=None if args.no_psd_out else o["psd"], cand_off=o["off"], cand_idx=o["idx"], cand_avg=o["avg"])
        else:
            eng.process_device(d_iq, nb, psd=None if args.no_psd_out else o["psd"], rel=o["rel_plane"], avg=o["avg_plane"], cand_off=o["off"], cand_idx=o["idx"], cand_avg=o["avg"])

    for _ in range(max(args.warmup, 1)):  # first warm-up batch also absorbs the noise-learning frames
        step()
    eng.sync()
    if args.lanes == 1:
        eng.kernel_timing(0 if args.no_kernel_timing else args.time_every)
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_enq = time.perf_counter()
    torch.cuda.synchronize()
    dist.barrier()
    t1 = time.perf_counter()
    kern_ms, launches = (0.0, 0)
    if args.lanes == 1:
        kern_ms, launches = eng.kernel_timing_read()
        eng.kernel_timing(0)
    elapsed = dist.max_over_ranks(t1 - t0, device=coll_dev)
    ncand = int(outs[(counter[0] - 1) % len(outs)]["off"][-1].item())

    if rank == 0:
        samples_per_step = nb * n * world
        value = samples_per_step * args.steps / elapsed / 1e6
        kern_avg_s = kern_ms / max(launches, 1) / 1e3
        algo_bytes = ALGO_BYTES_PER_SAMPLE if args.fmt == "cf32" else 6.0  # int8 IQ: 2 B in + 4 B out
        achieved = algo_bytes * nb * n / kern_avg_s / 1e9 if launches else None
        out = {
            "metric": "iq_msamples_per_sec_scanned_8192pt_fft" if n == 8192 else f"iq_msamples_per_sec_scanned_{n}pt_fft",
            "value": round(value, 1), "unit": "MS/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{n}-pt FFT, {fs / 1e6:.3f} MS/s, {nb}-frame batches, {args.fmt.upper()} IQ resident in HBM, full chain "
                                   "(window+FFT+dB -> noise-relative -> 21x21 mean -> threshold -> candidate lists), "
                                   "one band per GPU",
                       "fft_size": n, "frames_per_batch": nb, "bands": world, "candidates_per_batch": ncand,
                       "spectrogram_branch": bool(args.spectrogram), "psd_plane_out": not args.no_psd_out, "rel_avg_planes_out": bool(args.planes), "frame_decimation": args.decim, "lanes": args.lanes, "output_sets": len(outs), "host_enqueue_ms_per_step": round((t_enq - t0) / args.steps * 1e3, 4)},
            "roofline": {"bound": "hbm", "kernel": "k_fft8192_psd (load+window+FFT+dB)",
                         "achieved": None if achieved is None else round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": None if achieved is None else round(achieved / HBM_PEAK_GBS, 4),
                         "kernel_us": round(kern_avg_s * 1e6, 2), "launches": launches, "traffic": None},
        }
        traffic_file = os.path.join(ROOT, "profiles", "traffic_per_launch.json")
        if os.path.exists(traffic_file):  # HBM bytes per launch from the PMC passes (profiles/README.md), same command
            try:
                out["roofline"]["traffic"] = json.load(open(traffic_file)).get(f"{n}x{nb}")
            except Exception:
                pass
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(n, fs, args.cpu_seconds)
        elif world == 1:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)

    if world > 1:
        import torch.distributed as td
        td.destroy_process_group()


if __name__ == "__main__":
    main()
