#!/usr/bin/env python
"""bench.py — IQ MSamples/s scanned by the spectral-scan hot path on MI355X.

One step = one pass of the whole chain (load + Hamming window + FFT + dB power, noise-relative, 21-frame x 21-bin
averaging, threshold, candidate compaction) over one batch of synthetic frames already resident in HBM. The default is
BASELINE.json configs[1]: "8192-pt FFT, 2.048 MS/s, 1024-frame batches on 1x MI355X"; `--config 1|3|4|5` select the
other BASELINE configs. Every step reads a DIFFERENT input batch and writes a DIFFERENT output set; the sets in rotation
are sized to exceed the 256 MiB Infinity Cache several times over, so the memory traffic of a step is HBM traffic.

`python bench.py --gpus N` starts its own N ranks (torch.distributed.run, one process per GPU, RCCL) when it is not already
running under a launcher. Bands shard across ranks (`--shard bands`, config 4) or one band's frame stream is cut into
contiguous ranges (`--shard frames`, config 5); there is no data-path collective either way, the scan configuration is
broadcast once from rank 0.

Prints ONE JSON line on rank 0 — a compact form of a few KB (compact_line: the contract's keys, roofline, cpu_baseline, a parity verdict
and one short record per `also` entry); everything below in full goes to bench_full.json beside this file:
  value / ms_per_step   whole-job throughput, barrier + device sync on both sides of exactly --steps steps, max over ranks (a rank's interval:
                        from behind the opening barrier + sync to behind the closing sync; the closing barrier follows it)
  roofline              the dominant kernel (k_scan_step): algorithmic bytes per launch / its mean device time from start/stop
                        events attached to launches on the engine's own streams — in 32 more steps right behind the timed region
                        (an event-timed launch costs its queue ~13 us: none rides inside the timed region; --time-every k puts
                        them there). Consecutive launches overlap on two queues, so a launch lasts ~2x the time the GPU spends
                        per launch: kernel_us is the measured mean duration (what rocprofv3 --stats reports), launches_in_flight
                        = kernel_us / wall time per launch, achieved = bytes / (kernel_us / launches_in_flight)
  roofline.traffic      (default line, one GPU) fabric bytes per launch of that kernel, counted on this box: behind the timed region the
                        process runs its own command line once more under `rocprofv3 --pmc FETCH_SIZE` and under `--pmc WRITE_SIZE`
                        (two short passes); traffic_over_algorithmic = wasted re-reads, traffic_gbs = the counted bytes per second of
                        the timed region
  roofline.kernels      every launch of the chain with its own event-timed duration, what it must move per sample given the
                        decomposition, and the fabric bytes of the committed PMC passes (one entry at 8192 points; column half,
                        radix-A step, row half and plan launch for the long transforms)
  roofline_chain        the same algorithmic bytes / the whole step's time (every kernel of the chain, launch gaps included)
  also                  (default line, one GPU) short runs in processes of their own: the default configuration with every averaging tile
                        evaluated (variant no_cull: the data-independent cost), BASELINE configs 3 and 5 (65536 x 128 and x 256 CS8, 2^20 x 16
                        and x 64), each with ms_per_step, its chain figure, its kernels, the tiles the library culled and a parity sample
  cpu_baseline          the reference's own compiled sources (oracle/_ref; the C restatement where that is absent) on the
                        host cores over a bounded sample of the same workload, one thread and all threads
"""
from __future__ import annotations

import argparse
import json
import multiprocessing as mp
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
L3_BYTES = 256 << 20   # Infinity Cache

# BASELINE.json configs -> flags (config 2 is the default command line)
CONFIGS = {
    1: dict(cpu_only=True),
    2: dict(),
    3: dict(fft=65536, frames=128, fmt="cs8", sample_rate=20_000_000, no_psd_out=True),
    4: dict(gpus=8, shard="bands"),
    5: dict(gpus=8, shard="frames", fft=1 << 20, frames=16, sample_rate=61_440_000, no_psd_out=True),
}


def algo_bytes_per_sample(fmt: str, psd_out: bool) -> float:
    """IQ in (8 B CF32 / 2 B int8) + dB row out (4 B) in power mode; detect mode hands out candidates only."""
    return (8.0 if fmt == "cf32" else 2.0) + (4.0 if psd_out else 0.0)


# ---------------------------------------------------------------------------------------------- CPU baseline
def _cpu_worker(args):
    """One host core scanning its own band with the reference's code for `budget_s` seconds."""
    seed, n, fs, budget_s, use_ref, backend, stages = args
    import numpy as np
    import rtl_sdr_scanner_cpp_amd as pkg
    from oracle import oracle as O
    band = pkg.synth.SyntheticBand(n, seed=seed, on_frame=130, off_frame=10_000)
    chunk = max(1, min(64, (1 << 19) // n))
    iq = band.frames_cf32(chunk)
    center = 145_000_000
    frames = 0
    if use_ref and not stages:
        O.ref().orc_set_fft_backend(backend)
        chain = O.RefChain(n, fs, center - fs // 2, center + fs // 2)
        t_ms = 0
    else:
        O.lib().orc_set_fft_backend(backend)
        chain = O.oracle_chain(fs, center, fft_size=n, decim=1, max_batch=chunk)
    def one_chunk():
        nonlocal t_ms
        if use_ref and not stages:
            chain.process(iq, t_ms + 20 * np.arange(chunk))
            t_ms += 20 * chunk
        else:
            chain.process(iq, want=(), cand_cap=chunk * n)

    if not (use_ref and not stages):
        t_ms = 0
    one_chunk()  # untimed: the first call of a freshly forked worker pays for MKL's start-up (seconds, once)
    chunks = 0
    t0 = time.perf_counter()
    while True:
        one_chunk()
        frames += chunk
        chunks += 1
        el = time.perf_counter() - t0
        if el >= budget_s:
            break
    split = None
    if stages:
        import ctypes as C
        buf = (C.c_double * 6)()
        O.lib().orc_stage_seconds(chain._h, buf)
        tot = sum(buf) or 1.0
        names = ("window+fft+shift", "psd_db", "noise_relative", "averager_21_frames", "average_21_bins", "threshold")
        split = {k: round(v / tot, 3) for k, v in zip(names, buf)}
    return frames, el, split, chunks


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


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(n: int, fs: int, budget_s: float = 12.0, stages: bool = False):
    os.environ.setdefault("MKL_NUM_THREADS", "1")  # one FFT thread per worker process, like fft_v's nthreads = 1
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    from oracle import oracle as O
    use_ref = O.have_ref()
    # the reference runs FFTW through GNU Radio; MKL's FFTW3 interface is the closest thing on this host
    backend = 2 if (O.ref() if use_ref else O.lib()).orc_set_fft_backend(2) == 0 else 0
    cores = usable_cores()
    ctx = mp.get_context("fork")
    one_s = max(2.0, budget_s / 3.0)
    with ctx.Pool(1) as pool:  # (i) one thread, whole chain
        one = pool.map(_cpu_worker, [(999, n, fs, one_s, use_ref, backend, False)])[0]
    split = None
    if stages:
        with ctx.Pool(1) as pool:  # per-stage split (C restatement: the reference's classes carry no stage clocks)
            split = pool.map(_cpu_worker, [(998, n, fs, one_s, use_ref, backend, True)])[0][2]
    with ctx.Pool(cores) as pool:  # (ii) frame-parallel over all hardware threads, one chain per thread
        res = pool.map(_cpu_worker, [(1000 + i, n, fs, budget_s, use_ref, backend, False) for i in range(cores)])
    frames = sum(r[0] for r in res)
    wall = max(r[1] for r in res)
    out = {
        "value": round(frames * n / wall / 1e6, 3), "unit": "MS/s", "cores": cores,
        "kind": "reference" if use_ref else "port",
        # (fewer than ten chunks inside the budget is not a rate: null)
        "one_thread": round(one[0] * n / one[1] / 1e6, 3) if one[3] >= 10 else None, "nproc": os.cpu_count(), "cpu_model": cpu_model(),
        "sample": (f"{frames} frames of {n} CF32 samples ({frames * n / 1e6:.0f} MS) in {wall:.1f} s on {cores} threads "
                   f"(one independent band per thread), {one[0]} frames in {one[1]:.1f} s on one thread; "
                   f"full chain window+FFT+dB+noise+21x21 mean+threshold, "
                   f"{'reference .cpp files compiled in place (oracle/_ref)' if use_ref else 'C restatement (oracle/liboracle.so)'}, "
                   f"FFT via {'MKL FFTW3 interface' if backend == 2 else 'built-in radix-2'}"),
    }
    if split:
        out["stage_split_one_thread"] = split
    return out


def parity_sample(n: int, fs: int, fmt: str, no_cull: bool = False, call_frames: int | None = None):
    """A small parity check beside the numbers (the checker leg, like cpu_baseline), against the reference's own code (oracle/_ref;
    the C restatement where that is absent), same contract as tests/parity.py:
      host path    frames through ss_process with every plane handed out: achieved error quantiles per plane (dB and relative on
                   linear power), the bins outside the bare tolerance measured against an fp64 chain, candidate lists
      timed path   the same frames once more the way the timed region runs them — ss_process_device calls without a synchronisation
                   in between, the planes this configuration hands out and no others (detect mode: none), tile culling as the
                   library does it at this size — candidate lists against the reference, and what ss_get_stats says was culled.
                   call_frames: the timed region's own call size (long transforms: the form of a call depends on it) — the timed path then
                   runs a stream of its own: the learning frames as one call, then calls of that many frames, enough of them for
                   steady-state launches (a launch carries stages of up to five calls)."""
    import numpy as np
    import rtl_sdr_scanner_cpp_amd as pkg
    from oracle import oracle as O
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from parity import BAND, all_bins_vs_fp64, cand_set, check_all, dont_care_limit, error_quantiles, excess_vs_fp64, excess_vs_fp64_rel, linear_power_error, strict_excess
    chunk = 48 if n <= 16384 else (32 if n <= 65536 else 16)
    nframes = 2 * chunk if n <= 65536 else 5 * chunk  # (2^20 points: 80 frames, so that whole 16-frame tiles lie behind learning + warm-up and can be culled)
    n_learn = 21
    band = pkg.synth.SyntheticBand(n, seed=77, on_frame=nframes // 2, off_frame=nframes - 4)
    center = 145_000_000
    in_format = {"cf32": 0, "cs8": 1, "cu8": 2}[fmt]
    raw = band.frames_cf32(nframes) if fmt == "cf32" else (band.frames_cs8(nframes) if fmt == "cs8" else band.frames_cu8(nframes))
    if fmt == "cf32":
        iq_c = raw
    elif fmt == "cs8":  # what the engine's load stage makes of the bytes (include/specscan.h: int8 * 1 / 128)
        iq_c = (raw[..., 0].astype(np.float32) / np.float32(128.0) + 1j * (raw[..., 1].astype(np.float32) / np.float32(128.0))).astype(np.complex64)
    else:
        iq_c = ((raw[..., 0].astype(np.float32) - np.float32(127.5)) / np.float32(127.5) + 1j * ((raw[..., 1].astype(np.float32) - np.float32(127.5)) / np.float32(127.5))).astype(np.complex64)
    t = (1_000 + 100 * np.arange(nframes)).astype(np.int64)  # learning: the first 21 frames
    flags = pkg.abi.SS_FLAG_NO_CULL if no_cull else 0
    eng = pkg.SpectrumEngine(fs, center, fft_size=n, decim=1, in_format=in_format, max_batch=chunk, flags=flags)
    outs = [eng.process(raw[a:a + chunk], t_ms=t[a:a + chunk]) for a in range(0, nframes, chunk)]
    eng.close()
    got = {k: np.concatenate([o[k] for o in outs]) for k in ("psd", "rel", "avg", "cand_idx", "cand_avg")}
    got["cand_off"] = np.concatenate([[0], np.cumsum(np.concatenate([np.diff(o["cand_off"]) for o in outs]))]).astype(np.int32)
    if O.have_ref() and fmt != "cu8":
        O.ref().orc_set_fft_backend(0)
        r = O.RefChain(n, fs, center - fs // 2, center + fs // 2).process(iq_c, t)
        off = np.zeros(nframes + 1, np.int32)
        off[1:] = np.cumsum([len(c) for c in r["cands"]])
        ref = {"psd": r["psd"], "rel": r["rel"], "avg": r["avg"], "cand_off": off, "cand_idx": np.concatenate(r["cands"]).astype(np.int32)}
        against = "reference .cpp files compiled in place (oracle/_ref)"
    else:
        O.lib().orc_set_fft_backend(0)
        ch = O.oracle_chain(fs, center, fft_size=n, decim=1, in_format=in_format, max_batch=chunk)
        routs = [ch.process(raw[a:a + chunk], t_ms=t[a:a + chunk]) for a in range(0, nframes, chunk)]
        ref = {k: np.concatenate([o[k] for o in routs]) for k in ("psd", "rel", "avg", "cand_idx")}
        ref["cand_off"] = np.concatenate([[0], np.cumsum(np.concatenate([np.diff(o["cand_off"]) for o in routs]))]).astype(np.int32)
        against = "C restatement (oracle/liboracle.so)"
    errs, ncand, ndc = check_all(got, ref)  # raises when the contract is broken
    vs64 = excess_vs_fp64(iq_c, got["psd"], ref["psd"], fs)
    vs64_rel = excess_vs_fp64_rel(iq_c, got["rel"], ref["rel"], fs, n_learn)
    fmt3 = lambda d: None if d is None else {k: (v if isinstance(v, int) else float(f"{v:.3g}")) for k, v in d.items()}  # noqa: E731
    res = {"against": against, "frames": nframes, "reference_candidates": ncand, "inside_1e-3_dB_band": ndc, "band_limit": dont_care_limit(ncand),
           "abs_err_dB": {k: {q: float(f"{v:.3g}") for q, v in d.items()} for k, d in error_quantiles(got, ref).items()},
           # north_star's wording, on linear power |X|^2 / fs: |got / ref - 1| per bin of the PSD plane
           "rel_linear": {"psd": fmt3(linear_power_error(got["psd"], ref["psd"]))},
           # bins the bare 1e-4 * max(1, |ref|) does not cover (held by the fp32-FFT floor allowance), and on those bins the distance of
           # the engine and of the reference to an fp64 chain on the same windowed frames (engine <= 1.5 x reference asserted)
           "outside_bare_1e-4": {k: {"n": v["n"], "frac": float(f"{v['frac']:.3g}"), "worst_dB": float(f"{v['worst']:.3g}")} for k, v in strict_excess(got, ref).items()},
           "outside_bins_vs_fp64_fft_dB": {"psd": fmt3(vs64), "rel": fmt3(vs64_rel)},
           # ... and over ALL ordinary bins of a sample of rows: |dB - fp64| of the engine and of the reference's fp32 FFT, and the ratio of their rms
           "all_bins_vs_fp64_fft_dB": (lambda v: None if v is None else {"rows": v["rows"], "bins": v["bins"], "engine": fmt3(v["engine"]), "reference": fmt3(v["reference"]),
                                                                         "engine_over_reference_rms": float(f"{v['engine_over_reference_rms']:.3g}"),
                                                                         "engine_over_reference_median": float(f"{v['engine_over_reference_median']:.3g}"),
                                                                         "engine_over_reference_p99": float(f"{v['engine_over_reference_p99']:.3g}")})(all_bins_vs_fp64(iq_c, got["psd"], ref["psd"], fs))}
    # ---- the timed path: device calls, nothing synchronised in between ----
    import torch
    dev = torch.device("cuda", torch.cuda.current_device())
    own_stream = bool(call_frames and call_frames != chunk and (n >= 65536 or (n == 8192 and call_frames >= 512)))
    if own_stream:
        # the timed region's call size: a stream of its own (the learning frames, then whole calls), the reference's code over all of it
        # (8192 points, round 6: the headline's own 1024-frame calls, two of them behind the learning call — candidates only, ~0.3 s of the reference's code)
        chunk = call_frames
        ncalls = 2 if n == 8192 else (5 if n * call_frames <= (1 << 23) else (4 if n * call_frames <= (1 << 24) else 2))
        nframes = n_learn + ncalls * chunk
        band = pkg.synth.SyntheticBand(n, seed=78, on_frame=n_learn + 25, off_frame=nframes - 20)
        raw = band.frames_cs8(nframes) if fmt == "cs8" else (band.frames_cu8(nframes) if fmt == "cu8" else band.frames_cf32(nframes))
        if fmt == "cf32":
            iq_c = raw
        elif fmt == "cs8":
            iq_c = (raw[..., 0].astype(np.float32) / np.float32(128.0) + 1j * (raw[..., 1].astype(np.float32) / np.float32(128.0))).astype(np.complex64)
        else:
            iq_c = ((raw[..., 0].astype(np.float32) - np.float32(127.5)) / np.float32(127.5) + 1j * ((raw[..., 1].astype(np.float32) - np.float32(127.5)) / np.float32(127.5))).astype(np.complex64)
        t = (1_000 + 100 * np.arange(nframes)).astype(np.int64)
        if O.have_ref() and fmt != "cu8":
            O.ref().orc_set_fft_backend(2)  # (MKL's FFTW3 interface where it is there — candidates only are compared on this stream, and it is a hundred megasamples)
            r = O.RefChain(n, fs, center - fs // 2, center + fs // 2).process(iq_c, t)
            off = np.zeros(nframes + 1, np.int32)
            off[1:] = np.cumsum([len(c) for c in r["cands"]])
            ref = {"avg": r["avg"], "cand_off": off, "cand_idx": np.concatenate(r["cands"]).astype(np.int32)}
            del r
        else:
            ch = O.oracle_chain(fs, center, fft_size=n, decim=1, in_format=in_format, max_batch=chunk)
            cutsr = [(0, n_learn)] + [(a, a + chunk) for a in range(n_learn, nframes, chunk)]
            routs = [ch.process(raw[a:b], t_ms=t[a:b]) for a, b in cutsr]
            ref = {k: np.concatenate([o[k] for o in routs]) for k in ("avg", "cand_idx")}
            ref["cand_off"] = np.concatenate([[0], np.cumsum(np.concatenate([np.diff(o["cand_off"]) for o in routs]))]).astype(np.int32)
        del iq_c
    eng = pkg.SpectrumEngine(fs, center, fft_size=n, decim=1, in_format=in_format, max_batch=max(chunk, n_learn), learn_frames=n_learn, flags=flags, device_id=dev.index)
    d_iq, d_out = [], []
    cuts = [(0, n_learn)] + [(a, min(a + chunk, nframes)) for a in range(n_learn, nframes, chunk)]  # (the learning frames as a call of their own: the calls behind it overlap)
    for a, b in cuts:
        x = raw[a:b]
        d_iq.append(torch.from_numpy(x.view(np.float32) if x.dtype == np.complex64 else x).to(dev))
        d_out.append(dict(off=torch.zeros(b - a + 1, dtype=torch.int32, device=dev), idx=torch.empty((b - a) * 1024, dtype=torch.int32, device=dev),
                          # (8192 points at the timed call size: the dB plane handed out like the timed region's — power mode)
                          psd=torch.empty((b - a, n), dtype=torch.float32, device=dev) if (own_stream and n == 8192) else None))
    torch.cuda.synchronize()
    for d, o in zip(d_iq, d_out):
        eng.process_device(d, d.shape[0], psd=o["psd"], cand_off=o["off"], cand_idx=o["idx"])
    eng.sync()
    st = eng.stats()
    offs = [o["off"].cpu().numpy() for o in d_out]
    doff = np.concatenate([[0], np.cumsum(np.concatenate([np.diff(x) for x in offs]))]).astype(np.int32)
    didx = np.concatenate([o["idx"].cpu().numpy()[:x[-1]] for o, x in zip(d_out, offs)])
    eng.close()
    a_, b_ = cand_set(doff, didx), cand_set(ref["cand_off"], ref["cand_idx"])
    near = np.abs(ref["avg"] - np.float32(8.0)) < BAND
    outside = [(f, i) for (f, i) in a_ ^ b_ if not near[f, i]]
    if outside:
        raise AssertionError(f"timed path: candidate lists differ from the reference outside the {BAND} dB band: {sorted(outside)[:6]}")
    res["timed_path"] = {"what": f"{len(cuts)} ss_process_device calls of <= {chunk} frames, {'dB plane handed out, candidates compared' if (own_stream and n == 8192) else 'candidates only'}, no synchronisation in between", "reference_candidates": len(b_),
                         "inside_1e-3_dB_band": len(a_ ^ b_), "calls_overlapped": st["calls_overlapped"], "tiles_total": st["tiles_total"],
                         "tiles_tested": st["tiles_tested"], "tiles_culled": st["tiles_culled"], "wait_fallbacks": st["wait_fallbacks"]}
    return res


# ---------------------------------------------------------------------------------------------- launcher
PMC_FILES = {"fetch": "pmc_fetch.csv", "write": "pmc_write.csv"}  # under profiles/<PMC_SET>/, one counter per rocprofv3 pass
# the committed passes of each configuration's command line (config 2: round 4, scripts/r04/s8.sh; configs 3 and 5: round 6, scripts/r06/s17.sh)
PMC_SET = {2: "r04/s8_cfg2", 3: "r06/s17_cfg3", 5: "r06/s17_cfg5"}  # (config 2's kernel has not changed its traffic since; its default line measures it live: live_pmc_traffic)


def traffic_from_profiles(config: int, kernel_match: str, threads_per_launch: int | None = None):
    """Fabric bytes per launch of one kernel from the committed PMC passes of this configuration's command (`rocprofv3 --pmc
    FETCH_SIZE` and `--pmc WRITE_SIZE`, separate runs, --kernel-trace only), optionally for launches of one grid size only;
    bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (round-1 calibration of gfx950's FETCH_SIZE).
    Counters cannot be collected from inside a run, so this is not a live figure: None when the files or the shape are absent."""
    import csv
    base = PMC_SET.get(config)
    if base is None:
        return None
    out = {}
    for kind, name in PMC_FILES.items():
        path = os.path.join(ROOT, "profiles", base + "_" + name)
        if not os.path.exists(path):
            return None
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
                if kernel_match in r["Kernel_Name"] and (threads_per_launch is None or int(r["Grid_Size"]) == threads_per_launch)]
        if not vals:
            return None
        out[kind] = sum(vals) / len(vals)
    return {"bytes_per_launch": round((2.0 * out["fetch"] + out["write"]) * 1024.0), "fetch_kib": round(out["fetch"], 1), "write_kib": round(out["write"], 1),
            "source": f"profiles/{base}_pmc_fetch.csv, _pmc_write.csv (rocprofv3 --pmc passes of this configuration's command line, not this run)"}


def live_pmc_traffic(sub_argv, kernel_match: str, threads_per_launch: int | None, timeout_s: float = 90.0):
    """roofline.traffic, live: counters cannot be read from inside a run, so rank 0 — once the timed region is over and the GPU idle —
    runs this command line once more in a short form under `rocprofv3 --pmc FETCH_SIZE` and again under `--pmc WRITE_SIZE`
    (separate passes, --kernel-trace only, from /tmp with TMPDIR=/tmp: MI355X_MICROARCH.md's recipe) on the same box, and averages
    the counter over the launches of the dominant kernel's steady-state shape. bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
    (gfx950's FETCH_SIZE counts 64-byte requests in KiB as if they were 32-byte ones: round-1 calibration against a copy kernel,
    DESIGN.md 4). None, with the reason, when rocprofv3 is not there, times out or finds no such launch."""
    import csv
    import glob
    import shutil
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, "rocprofv3 not found"
    out, n_launches = {}, 0
    env = dict(os.environ, TMPDIR="/tmp")
    t0 = time.perf_counter()
    for kind, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        with tempfile.TemporaryDirectory(dir="/tmp", prefix="ss_pmc_") as d:
            cmd = [rocprof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__), *sub_argv]
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            except subprocess.TimeoutExpired:
                return None, f"rocprofv3 --pmc {counter}: no answer within {timeout_s:.0f} s"
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {counter}: rc {r.returncode}, {len(files)} counter files: {(r.stderr or r.stdout)[-160:]}"
            vals = [float(row["Counter_Value"]) for f in files for row in csv.DictReader(open(f))
                    if row.get("Counter_Name") == counter and kernel_match in row["Kernel_Name"] and (threads_per_launch is None or int(row["Grid_Size"]) == threads_per_launch)]
            if not vals:
                return None, f"rocprofv3 --pmc {counter}: no launch of {kernel_match} with {threads_per_launch} threads in the pass"
            out[kind] = sum(vals) / len(vals)
            n_launches = len(vals)
    return {"bytes_per_launch": round((2.0 * out["fetch"] + out["write"]) * 1024.0), "fetch_kib": round(out["fetch"], 1), "write_kib": round(out["write"], 1),
            "launches_averaged": n_launches, "seconds": round(time.perf_counter() - t0, 1),
            "how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two passes, --kernel-trace only) of this command line with --steps 30, run by this process on this box "
                   "behind the timed region; launches of the steady-state shape; bytes = (2 FETCH_SIZE + WRITE_SIZE) KiB"}, None


def is_preset(args) -> bool:
    """The command line is the configuration's own (the committed PMC passes are of that one)."""
    want = dict(fft=8192, frames=1024, fmt="cf32", sample_rate=None, no_psd_out=False)
    want.update({k: v for k, v in CONFIGS.get(args.config or 2, {}).items() if k in want})
    if want["sample_rate"] is None:
        want["sample_rate"] = 2_048_000 * (want["fft"] // 8192 if want["fft"] >= 8192 else 1)
    return (all(getattr(args, k) == v for k, v in want.items()) and not args.planes and not args.spectrogram and args.decim == 1 and not args.no_cull
            and not args.sync_every_step and not (args.diag_lib or args.lib) and args.start_level == 8.0)


def chain_kernels(n: int, fmt: str, nb: int | None = None, detect_mode: bool = False):
    """The launches one call of the chain takes, as the library's timing slots name them (include/specscan.h SS_KSLOT_*), with
    what each must move per sample given the decomposition (its inputs once + its outputs once). detect_mode: the
    call hands out no plane (candidates only)."""
    in_b = 8.0 if fmt == "cf32" else 2.0
    if n == 8192:
        return [("step", "k_scan_step", "k_scan_step: load+window+FFT+dB of call k, carrying the 21x21 mean + threshold of call k-2 and the candidate lists "
                 "of call k-4 as further roles of the same launch; consecutive launches alternate over two queues and overlap", in_b + 4.0)]
    if n < 16384:
        return [("step", "k_fft", "load+window+FFT+dB (one launch); detect and emit stages follow as launches of their own", in_b + 4.0)]
    if n == 1 << 20 and os.environ.get("SS_FFT_TWOPASS") != "0":  # 1024 x 1024 in two passes (csrc/fft1024_kernels.h; the three-pass form is a switch of the diagnostics build)
        return [("step", "k_fft_cols1024", "k_fft_cols1024_plan: column half of the two-pass FFT (load, Hamming taps formed from one table entry per thread, 1024-point FFTs in "
                 "registers + LDS, twiddle -> work buffer), 16 columns x 1024 rows per 1024-thread workgroup; its first 128 workgroups run the plan of the call before "
                 "(which averaging tiles can hold a candidate, from the run maxima that call's row half left)", in_b + 8.0),
                ("rows", "k_scan_step", "k_scan_step with the ROW half as its FFT role (fft_rows1024_tile: 1024-point FFTs -> dB -> noise-relative rows straight into the averager "
                 "ring's buffer, no dB plane in detect mode, + run maxima for the tile culling), carrying the listed averaging tiles of call k-1 and the candidate lists of call k-2", 12.0),
                ("plan", "k_plan_long", "k_plan_long as a launch of its own (SS_PLAN_FUSED=0 of the diagnostics build; the product runs the plan at the front of the next column launch)", 0.0)]
    if n in (65536, 131072) and fmt != "cf32" and detect_mode and os.environ.get("SS_DIF8") != "0" and os.environ.get("SS_CULL_65536") != "0" and os.environ.get("SS_ROWS256_STEP") != "0":
        # 65536 / 131072 points, int8 IQ, calls that keep no plane (round 5): the radix-8 / radix-16 fold — no work buffer, one launch per call whatever its length
        q = n // 8192
        return [("step", f"true, false, {8 if q == 8 else 9}>", f"k_scan_step (KIND {8 if q == 8 else 9}), one launch per call: the radix-{q} decimation-in-frequency fold in the load stage of the "
                 f"8192-point transform (csrc/fft65536_dif8.h) — {q // 2} workgroups per frame, each folding the whole int8 frame (LDS-DMA pieces, Hamming taps formed, W_{q} rotations) into the "
                 f"8192 points of residues r and r + {q // 2} and running the 8192-point transform on both -> dB -> dB rows, in blocks of 32 Q bins, straight into the averager ring's "
                 "buffer (no work buffer, no dB plane) + run maxima for the tile culling —, carrying the plan of call k-1 (which averaging tiles can hold a candidate), the listed tiles of "
                 "call k-2 (21x21 mean + threshold on those rows) and the candidate lists of call k-3", in_b + 4.0)]
    if n == 65536 and os.environ.get("SS_CULL_65536") != "0" and os.environ.get("SS_ROWS256_STEP") != "0" and os.environ.get("SS_MERGE_65536") != "0" and (nb is None or nb <= 128):
        # 65536 points, detect-mode calls of up to 128 frames as the product runs them since session 36 of round 4: ONE launch per call
        return [("step", "k_scan_step", "k_scan_step (KIND 7), one launch per call: the column half of the four-step FFT of call k (load, Hamming taps formed from one table entry per "
                 "thread, 256-point FFTs, twiddle -> one of two work buffers) and, dispatched behind its tiles, the ROW half of call k-1 (256-point FFTs -> dB -> noise-relative rows "
                 "straight into the averager ring's buffer, no dB plane in detect mode, + run maxima for the tile culling), carrying the plan of call k-2 (which averaging tiles can hold "
                 "a candidate), the listed tiles of call k-3 (21x21 mean + threshold) and the candidate lists of call k-4", in_b + 8.0 + 12.0)]
    if n == 65536 and os.environ.get("SS_CULL_65536") != "0" and os.environ.get("SS_ROWS256_STEP") != "0":
        # ... calls of more frames (and SS_MERGE_65536=0): both halves of the FFT are FFT roles of k_scan_step launches of their own
        return [("step", "k_scan_step", "k_scan_step (KIND 2): column half of the four-step FFT of call k (load, Hamming taps formed from one table entry per thread, 256-point FFTs, "
                 "twiddle -> work buffer), carrying the plan of call k-1 (which averaging tiles can hold a candidate), the listed tiles of call k-2 (21x21 mean + threshold, "
                 "the first 64 pairs on workgroups of their own) and the candidate lists of call k-3", in_b + 8.0),
                ("rows", "k_scan_step", "k_scan_step (KIND 6): row half (256-point FFTs -> dB -> noise-relative rows straight into the averager ring's buffer, no dB plane in "
                 "detect mode, + run maxima for the tile culling); carries nothing", 12.0),
                ("plan", "k_plan_long", "k_plan_long as a launch of its own (drains only: in a run of calls the plan is a role of the column launch)", 0.0)]
    if n == 1 << 18 and os.environ.get("SS_ROWS1024X256") != "0" and os.environ.get("SS_CULL") != "0":
        # 262144 points — the size getFft picks at 61.44 MS/s — as they ship since round 6: 256-point column tiles + the 1024-point row tile, culled
        if detect_mode and os.environ.get("SS_MERGE_65536") != "0" and (nb is None or nb <= 32):
            return [("step", "k_scan_step", "k_scan_step (KIND 12), one launch per call: the column half of the four-step FFT of call k (256-point column tiles -> one of two work "
                     "buffers) and, dispatched behind its tiles, the ROW half of call k-1 (8 rows of 1024 points per workgroup -> dB rows straight into the averager ring's buffer, no "
                     "dB plane in detect mode, + run maxima by atomic maxima), carrying the plan of call k-2, the listed tiles of call k-3 and the candidate lists of call k-4", in_b + 8.0 + 12.0)]
        return [("step", "k_scan_step", "k_scan_step (KIND 10 / 12): column half of the four-step FFT of call k (load, window, 256-point FFTs, twiddle -> work buffer), carrying the plan of "
                 "call k-1 (which averaging tiles can hold a candidate, from the run maxima the row tiles left), the listed tiles of call k-2 and the candidate lists of call k-3", in_b + 8.0),
                ("rows", "k_fft_rows1024_psd", "k_fft_rows1024_psd<8>: row half — 8 rows of 1024 points per workgroup (four interleaved 256-point FFTs and a radix-4 step) -> dB -> dB rows "
                 "straight into the averager ring's buffer (no dB plane in detect mode) + the run maxima for the tile culling (atomic maxima on keys); carries nothing", 12.0),
                ("plan", "k_plan_long", "k_plan_long as a launch of its own (drains only: in a run of calls the plan is a role of the column launch)", 0.0)]
    n2 = n // 256
    ks = [("step", "k_scan_step", "k_scan_step: column half of the four-step FFT of call k (load, window, 256-point FFTs, twiddle -> work buffer), carrying the "
           "21x21 mean + threshold of call k-1 (the tiles the plan listed) and the candidate lists of call k-2 as further roles", in_b + 8.0)]
    if n2 >= 2048:
        ks.append(("sub", "k_fft_sub_dft", f"k_fft_sub_dft: radix-{n2 // 256} step of the {n2}-point rows, in place in the work buffer", 16.0))
    ks.append(("rows", "k_fft_rows", "row half: 256-point FFTs -> dB rows (+ run maxima and the averager ring rows for the tile culling)", 12.0))
    ks.append(("plan", "k_plan_long", "k_plan_long: which averaging tiles of the call can hold a candidate, from the run maxima the rows kernel left (one list per call)", 0.0))
    return ks


def kernel_tally(chain, slots, n: int, nb: int, peak_gbs: float = HBM_PEAK_GBS):
    """roofline.kernels from the library's tally of the sampled launches — slots = {slot: (total ms, launches, frames those launches
    covered)} (ss_kernel_timing_read_frames). Everything is per LAUNCH: a call the library takes through in chunks (2^20 points beyond
    16 frames, the four-step form of 65536 points beyond 256) has several launches per slot, each over its chunk's frames, and its
    bytes are those of the chunk — dividing a call's bytes by a launch's duration (round 4) gave fractions of peak above 1."""
    kernels = []
    for slot, match, what, bps in chain:
        ms_k, cnt_k, fr_k = slots.get(slot, (0.0, 0, 0))
        if not cnt_k:
            continue
        us = ms_k / cnt_k * 1e3
        frames_per_launch = fr_k / cnt_k if fr_k else float(nb)
        kb = bps * frames_per_launch * n
        kernels.append({"slot": slot, "match": match, "what": what, "us": round(us, 2), "launches_timed": cnt_k, "frames_per_launch": round(frames_per_launch, 2),
                        "launches_per_call": round(nb / frames_per_launch, 2) if frames_per_launch else None,
                        "bytes_per_launch_it_must_move": kb, "gbs": round(kb / us / 1e3, 1) if us else None,
                        "frac_of_peak": round(kb / us / 1e3 / peak_gbs, 4) if us else None})
    return kernels


def rank_plan(rank: int, local_rank: int, world: int, ndev: int, args, n: int) -> dict:
    """Which device a rank takes and which of the side legs of the line it runs. One process per GPU: rank r of a node works on device
    LOCAL_RANK (modulo the device count only where a box has fewer GPUs than ranks — ranks then share devices over gloo and the line
    says so). The legs beside the timed region — CPU baseline, live PMC passes, the `also` runs, the parity sample — are rank 0's, at
    N = 1 only: at N > 1 the timed region is all a rank does (tests/test_dist_gloo.py pins this on the CPU)."""
    single = world == 1 and rank == 0
    default2 = (args.config or 2) == 2 and n == 8192
    return {"device_index": local_rank % max(ndev, 1), "shares_device": world > max(ndev, 1),
            "cpu_baseline": single and not args.no_cpu_baseline,
            "live_pmc": single and not args.sub and not args.no_live_pmc and is_preset(args) and default2,
            "also": single and not args.sub and not args.no_also and default2 and not (args.diag_lib or args.lib),
            "parity": single and not args.no_parity and (not args.no_cpu_baseline or args.sub)}


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args, argv) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks, one per GPU (RCCL). On a box with fewer GPUs than
    ranks the ranks share devices over gloo (functional only; the JSON line says so)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if "SS_DIST_BACKEND" not in env:
        ndev = 0
        if not args.launch_check:
            import torch
            ndev = torch.cuda.device_count()
        if ndev < args.gpus:
            env["SS_DIST_BACKEND"] = "gloo"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__), *argv]
    return subprocess.call(cmd, env=env)


def parse_args(argv):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--gpus", type=int, default=None)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=None, choices=sorted(CONFIGS), help="BASELINE.json config number (1-5); explicit flags win over the preset")
    ap.add_argument("--frames", type=int, default=None, help="frames per batch (BASELINE config 2: 1024)")
    ap.add_argument("--fft", type=int, default=None)
    ap.add_argument("--sample-rate", type=int, default=None)
    ap.add_argument("--shard", default=None, choices=["bands", "frames"], help="N > 1: one band per rank, or contiguous frame ranges of one band")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fmt", default=None, choices=["cf32", "cs8", "cu8"], help="IQ sample format in HBM (the headline is cf32)")
    ap.add_argument("--spectrogram", action="store_true", help="also run the Spectrogram side branch (SS_FLAG_SPECTROGRAM) every batch")
    ap.add_argument("--no-psd-out", action="store_true", default=None, help="detect mode: the caller takes candidates only, no PSD plane is handed out")
    ap.add_argument("--planes", action="store_true", help="full mode: the rel and avg planes are handed out as well (20 B/sample)")
    ap.add_argument("--decim", type=int, default=1, help="frame decimation D: items of N*D samples, the first N of each are scanned (reference: 5 at 2.048 MS/s)")
    ap.add_argument("--sync-every-step", action="store_true", help="ss_sync after every step: no overlap between consecutive calls (what a caller that reads every result before the next call sees)")
    ap.add_argument("--sets", type=int, default=0, help="input batches / output sets in rotation (0 = enough to exceed 2.5x the Infinity Cache, at least 6)")
    ap.add_argument("--time-every", type=int, default=0, help="attach start/stop events to the launches of every k-th call INSIDE the timed region (0 = none there: the launches are timed in 32 more steps behind it)")
    ap.add_argument("--no-kernel-timing", action="store_true", help="do not attach per-launch events to the FFT kernel (roofline omitted)")
    ap.add_argument("--preheat-ms", type=float, default=400.0, help="run untimed steps for this long before the W warm-up steps: the GPU's clocks take a few hundred steps to settle (a cold start reads ~10 %% slow)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--diag-lib", action="store_true", help="load libspecscan_diag.so (-DSS_DIAG: SS_* environment variables select alternative implementations; measurement runs only)")
    ap.add_argument("--lib", default=None, help="path of an A/B build of the diagnostics library (build.build_variant; measurement runs only)")
    ap.add_argument("--start-level", type=float, default=8.0, help="Device::m_startLevel in dB over the learned ceiling (reference default 8)")
    ap.add_argument("--no-cull", action="store_true", help="SS_FLAG_NO_CULL: evaluate every averaging tile, also those whose segment maxima rule out a candidate (the data-independent cost of the chain)")
    ap.add_argument("--launch-check", action="store_true", help="exercise launcher, rendezvous, config broadcast and max-over-ranks timing only (no GPU work)")
    ap.add_argument("--sync-engine-first", action="store_true", help="end of the timed region as in round 2: ss_sync, then torch.cuda.synchronize() (A/B; the default lets the device-wide synchronisation do the waiting)")
    ap.add_argument("--host-wait", default="auto", choices=["auto", "spin", "yield", "block"], help="how the host thread waits for the device (hipSetDeviceFlags: hipDeviceScheduleAuto / Spin / Yield / BlockingSync)")
    ap.add_argument("--no-also", action="store_true", help="default line only: do not append the short runs of BASELINE configs 3 and 5 (`also`)")
    ap.add_argument("--also-all", action="store_true", help="`also` with two more call sizes (config 3 in 256-frame calls, config 5 in 64-frame calls)")
    ap.add_argument("--sub", action="store_true", help="(internal) this process is one of the `also` runs of another bench.py")
    ap.add_argument("--no-live-pmc", action="store_true", help="roofline.traffic stays null: do not run the two short rocprofv3 --pmc passes of this command line behind the timed region (default line of config 2 at N = 1 only)")
    ap.add_argument("--no-parity", action="store_true", help="no parity sample beside the numbers (the `also` runs carry one each unless told otherwise)")
    args = ap.parse_args(argv)
    preset = dict(CONFIGS.get(args.config or 2, {}))
    args.cpu_only = bool(preset.pop("cpu_only", False))
    for k, v in preset.items():
        if getattr(args, k, None) in (None, False):
            setattr(args, k, v)
    args.gpus = args.gpus or 1
    args.frames = args.frames or 1024
    args.fft = args.fft or 8192
    args.fmt = args.fmt or "cf32"
    args.shard = args.shard or "bands"
    args.no_psd_out = bool(args.no_psd_out)
    if args.sample_rate is None:
        args.sample_rate = 2_048_000 * (args.fft // 8192 if args.fft >= 8192 else 1)
    return args


PCIE_GBS = 63.0  # PCIe Gen5 x16 (MI355X_MICROARCH.md): the bound of every path that takes HOST buffers


def drop_in_lines():
    """The drop-in path's rate at the long transforms (what a GNU Radio block sees: host buffers in, results back, PCIe inclusive) —
    never the headline `value` (its inputs are not resident in HBM), but the number a maintainer gets after INTEGRATION.md's diff:
      ss_process   one work() call on pageable host buffers, every stage drained before it returns (include/specscan.h), candidates only
      ss_feed_*    the pipelined form for a source that owns its buffers: pinned staging, H2D of batch k + 1 beside the chain of batch k
    for BASELINE config 3 (65536 points, int8 IQ, 128-frame calls) and config 5 (2^20 points, CF32, 16-frame calls), each beside its
    PCIe bound (input bytes per sample over 63 GB/s)."""
    import numpy as np
    import rtl_sdr_scanner_cpp_amd as pkg
    res = []
    for cfg_no, n, fs, fmt, nb in ((3, 65536, 20_000_000, "cs8", 128), (5, 1 << 20, 61_440_000, "cf32", 16)):
        try:
            in_format = {"cf32": 0, "cs8": 1}[fmt]
            band = pkg.synth.SyntheticBand(n, seed=9, on_frame=40, off_frame=10_000)
            learn = band.frames_cs8(32) if fmt == "cs8" else band.frames_cf32(32)
            batch = band.frames_cs8(nb) if fmt == "cs8" else band.frames_cf32(nb)
            kw = dict(fft_size=n, decim=1, in_format=in_format, learn_frames=32, max_batch=max(nb, 32))
            eng = pkg.SpectrumEngine(fs, 145_000_000, **kw)
            # (candidate capacity 2^20 per call, as a block with output buffers of its own has: the wrapper's default — nframes x N entries,
            # two fresh 64 MiB numpy arrays per call at 2^20 points, unmapped again when the call returns — made THIS entry ten times
            # slower at the end of a long-lived process than alone in one: the copy in took 19-29 ms instead of 2.4 whenever such arrays
            # had just been unmapped, profiles/r06/s10_summary.txt)
            ccap = 1 << 20
            eng.process(learn, want=(), cand_cap=ccap)
            for _ in range(2):
                eng.process(batch, want=(), cand_cap=ccap)
            reps = 6
            t0 = time.perf_counter()
            for _ in range(reps):
                eng.process(batch, want=(), cand_cap=ccap)
            dt_proc = (time.perf_counter() - t0) / reps
            # ... and its pieces, each alone (the 2^20-point entry read 1.8 GS/s on round 5's driver box and 6.2 on the builder's):
            # the same pageable buffer to the device through the runtime (what hipMemcpy2DAsync of ss_process does: the runtime
            # pins the caller's pages for the copy — cheap on 2 MiB pages, a page-table walk per 4 KiB page otherwise), and the
            # chain on resident frames, drained (ss_process_device + ss_sync)
            import torch
            dev = torch.device("cuda", torch.cuda.current_device())
            host_t = torch.from_numpy(batch.view(np.float32) if batch.dtype == np.complex64 else batch)
            d_t = torch.empty_like(host_t, device=dev)
            d_off = torch.zeros(nb + 1, dtype=torch.int32, device=dev)
            d_idx = torch.empty(1 << 20, dtype=torch.int32, device=dev)
            d_t.copy_(host_t)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                d_t.copy_(host_t)
                torch.cuda.synchronize()
            dt_h2d = (time.perf_counter() - t0) / reps
            eng.process_device(d_t, nb, cand_off=d_off, cand_idx=d_idx)
            eng.sync()
            t0 = time.perf_counter()
            for _ in range(reps):
                eng.process_device(d_t, nb, cand_off=d_off, cand_idx=d_idx)
                eng.sync()
            dt_chain = (time.perf_counter() - t0) / reps
            try:
                thp = open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip()
            except OSError:
                thp = None
            pieces = {"total": round(dt_proc * 1e3, 2), "h2d_pageable_alone": round(dt_h2d * 1e3, 2), "chain_resident_drained": round(dt_chain * 1e3, 3),
                      "h2d_gbs": round(batch.nbytes / dt_h2d / 1e9, 1), "transparent_hugepage": thp}
            del d_t, d_off, d_idx
            eng.close()
            eng = pkg.SpectrumEngine(fs, 145_000_000, **kw)
            eng.process(learn, want=())
            feed = eng.feed(depth=3, cand_cap=1 << 20)
            for _ in range(3):
                feed.acquire()[:nb] = batch
                feed.submit(nb)
            for _ in range(3):
                feed.collect()
            reps, sub, done = 12, 0, 0
            t0 = time.perf_counter()
            while done < reps:
                while sub < reps and feed.pending < 3:
                    feed.acquire()  # (a source that writes into the pinned slot itself: nothing copied on the host)
                    feed.submit(nb)
                    sub += 1
                feed.collect()
                done += 1
            dt_feed = (time.perf_counter() - t0) / reps
            feed.close()
            eng.close()
            in_b = 2.0 if fmt == "cs8" else 8.0
            res.append({"baseline_config": cfg_no, "variant": "drop_in_path", "fft_size": n, "frames_per_call": nb, "fmt": fmt,
                        "ss_process_MSps": round(nb * n / dt_proc / 1e6, 1), "ss_feed_MSps": round(nb * n / dt_feed / 1e6, 1),
                        "pcie_bound_MSps": round(PCIE_GBS * 1e9 / in_b / 1e6, 1), "ss_process_pieces_ms": pieces,
                        "what": "host buffers in, candidate lists back, PCIe inclusive: ss_process (pageable buffers, drained every call) and ss_feed_* (pinned "
                                "staging, depth 3, copy of batch k + 1 beside the chain of batch k); bound = 63 GB/s over the input's bytes per sample"})
        except Exception as e:  # the default line must not depend on these
            res.append({"baseline_config": cfg_no, "variant": "drop_in_path", "error": f"{type(e).__name__}: {str(e)[:200]}"})
    return res


def also_lines(all_sizes: bool = False):
    """Beside the default line, as short runs of this script in processes of their own: the default configuration once more with
    every averaging tile evaluated (`--no-cull`: the data-independent cost of the chain — the reference evaluates every bin of every
    frame, transmission.cpp:88-96), and BASELINE configs 3 and 5 (one GPU each). Every entry has ms_per_step, the chain's rate, every
    kernel of the chain with its own duration and rate, what the library says it culled (ss_get_stats), and — except the last — a
    parity sample of its own against the reference (host path with every plane, and the timed device path in the entry's own mode)."""
    res = []
    runs = [(2, 40, ["--no-cull"], "no_cull"), (3, 200, [], None), (3, 60, ["--frames", "512", "--no-parity"], None), (5, 100, [], None),
            # the sizes the reference itself would run these two signals at (getFft, utils/radio_utils.cpp:98-104: the first power of two
            # whose bins are at most 250 Hz wide): 131072 points at 20 MS/s, 262144 at 61.44 MS/s — 8.4 MS per call like configs 3 and 5
            (3, 60, ["--fft", "131072", "--frames", "64"], "getFft_131072"), (5, 60, ["--fft", "262144", "--frames", "32"], "getFft_262144")]
    if all_sizes:  # (--also-all: config 3 also in 256-frame calls — two rounds of workgroups per launch instead of one —, config 5 also in 64-frame calls: four chunks of 16)
        runs[3:3] = [(3, 100, ["--frames", "256"], None)]
        runs.insert(5, (5, 40, ["--frames", "64"], None))
    for cfg_no, steps, extra, variant in runs:
        cmd = [sys.executable, os.path.abspath(__file__), "--config", str(cfg_no), "--gpus", "1", "--steps", str(steps), "--warmup", "5",
               "--preheat-ms", "150", "--no-cpu-baseline", "--sub", *extra]
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
            line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
            j = json.loads(line)
            entry = {"baseline_config": cfg_no, "fft_size": j["config"]["fft_size"], "frames_per_batch": j["config"]["frames_per_batch"], "workload": j["config"]["workload"], "metric": j["metric"], "value": j["value"], "unit": j["unit"],
                     "steps": j["steps"], "ms_per_step": j["ms_per_step"], "psd_plane_out": j["config"]["psd_plane_out"], "tile_culling": j["config"]["tile_culling"],
                     "tiles": j["config"].get("tiles"), "candidates_per_batch": j["config"]["candidates_per_batch"],
                     "roofline_chain": j["roofline_chain"], "kernels": j["roofline"]["kernels"], "parity": j.get("parity")}
            if variant:
                entry = {"variant": variant, **entry}
            res.append(entry)
        except Exception as e:  # the default line must not depend on these
            res.append({"baseline_config": cfg_no, **({"variant": variant} if variant else {}), "error": f"{type(e).__name__}: {str(e)[:200]}"})
    return res


# ---------------------------------------------------------------------------------------------- the line the driver parses
FULL_LINE_FILE = "bench_full.json"  # beside bench.py (and under gpurun_out/ when that exists): everything the run measured


def _short(text, limit: int):
    if text is None:
        return None
    text = str(text)
    return text if len(text) <= limit else text[:limit - 3] + "..."


def _compact_parity(p):
    """A parity block in a few numbers: who it was checked against, how many reference candidates, how many of them inside the
    1e-3 dB threshold band (counted, not compared), bins outside the bare 1e-4 per plane, the rms distance to an fp64 FFT over all
    bins as a ratio engine / reference, and the timed path's own leg."""
    if not p:
        return None
    if "failed" in p:
        return {"failed": _short(p["failed"], 200)}
    out = {"against": "oracle/_ref" if "_ref" in p.get("against", "") else "oracle/liboracle.so", "frames": p.get("frames"),
           "reference_candidates": p.get("reference_candidates"), "inside_band": p.get("inside_1e-3_dB_band"),
           "outside_bare_1e-4": {k: v["n"] for k, v in (p.get("outside_bare_1e-4") or {}).items()},
           "worst_dB": max([v["worst_dB"] for v in (p.get("outside_bare_1e-4") or {}).values()] or [0.0]),
           "rel_linear_p999": ((p.get("rel_linear") or {}).get("psd") or {}).get("p99.9"),
           "engine_over_reference_rms": (p.get("all_bins_vs_fp64_fft_dB") or {}).get("engine_over_reference_rms"),
           "engine_over_reference_median": (p.get("all_bins_vs_fp64_fft_dB") or {}).get("engine_over_reference_median"),
           "engine_over_reference_p99": (p.get("all_bins_vs_fp64_fft_dB") or {}).get("engine_over_reference_p99")}
    t = p.get("timed_path")
    if t:
        out["timed_path"] = {"calls": _short(t.get("what"), 60), "reference_candidates": t.get("reference_candidates"), "inside_band": t.get("inside_1e-3_dB_band"),
                             "tiles_tested": t.get("tiles_tested"), "tiles_culled": t.get("tiles_culled"), "wait_fallbacks": t.get("wait_fallbacks")}
    return out


def _compact_also(e):
    if "error" in e:
        return {k: (_short(v, 160) if k == "error" else v) for k, v in e.items()}
    if e.get("variant") == "drop_in_path":
        return {k: e.get(k) for k in ("variant", "baseline_config", "fft_size", "frames_per_call", "fmt", "ss_process_MSps", "ss_feed_MSps", "pcie_bound_MSps", "ss_process_pieces_ms") if k in e}
    par = e.get("parity")
    out = {"config": e.get("baseline_config"), "fft_size": e.get("fft_size"), "frames_per_call": e.get("frames_per_batch"), "value": e.get("value"), "ms_per_step": e.get("ms_per_step"),
           "chain_frac": (e.get("roofline_chain") or {}).get("frac"), "pmc_B_per_sample": (e.get("roofline_chain") or {}).get("pmc_bytes_per_sample_from_profiles"),
           "tile_culling": e.get("tile_culling"), "evaluated_frac": (e.get("tiles") or {}).get("evaluated_frac"),
           "kernels_us": [k.get("us") for k in (e.get("kernels") or [])],
           "parity": None if not par else ("FAILED" if "failed" in par else
                                          {"ok": True, "reference_candidates": (par.get("timed_path") or par).get("reference_candidates"),
                                           "inside_band": (par.get("timed_path") or {}).get("inside_1e-3_dB_band", par.get("inside_1e-3_dB_band")),
                                           "engine_over_reference_rms": (par.get("all_bins_vs_fp64_fft_dB") or {}).get("engine_over_reference_rms")})}
    if e.get("variant"):
        out = {"variant": e["variant"], **out}
    return {k: v for k, v in out.items() if v is not None}


def compact_line(full: dict) -> dict:
    """The ONE line the driver parses (the last line of stdout): the contract's keys, `roofline` and `cpu_baseline` in a few numbers each,
    the parity sample's verdict and one short record per `also` entry — a few KB whatever the run measured. Everything else (every kernel
    of every chain with its description, the parity blocks' quantiles, the traffic passes' details) is the full form in bench_full.json;
    round 5's single ~30 KB line was more than the driver parses. tests/test_bench_helpers.py holds this to < 6 KB."""
    c = full.get("config") or {}
    keep_c = ("workload", "baseline_config", "fft_size", "frames_per_batch", "bands", "shard", "psd_plane_out", "tile_culling", "tiles", "candidates_per_batch",
              "input_sets", "working_set_mib", "dist_backend", "ranks_share_devices", "device_index_of_ranks", "candidates_last_batch_of_ranks", "host_enqueue_ms_per_step", "diag_lib", "file_fed")
    out = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    out["config"] = {k: (_short(c[k], 260) if k == "workload" else c[k]) for k in keep_c if k in c}
    r = full.get("roofline")
    if r:
        keep_r = ("bound", "achieved", "peak", "unit", "frac", "kernel_us", "kernel_timing", "launches", "launches_in_flight", "achieved_if_launches_did_not_overlap",
                  "algorithmic_bytes_per_launch", "traffic", "traffic_over_algorithmic", "traffic_gbs", "traffic_frac_of_peak")
        out["roofline"] = {"kernel": _short(r.get("kernel"), 120), **{k: r[k] for k in keep_r if k in r}}
        if r.get("traffic") is None and isinstance(r.get("traffic_live"), dict) and "unavailable" in r["traffic_live"]:
            out["roofline"]["traffic_unavailable"] = _short(r["traffic_live"]["unavailable"], 120)
        if len(r.get("kernels") or []) > 1:  # chains of several launches: each kernel's own duration and rate on the bytes the design makes it move
            out["roofline"]["kernels"] = [{"slot": k["slot"], "us": k["us"], "frames_per_launch": k["frames_per_launch"], "frac_of_peak_on_design_bytes": k["frac_of_peak"]} for k in r["kernels"]]
    else:
        out["roofline"] = None
    rc = full.get("roofline_chain")
    if rc:
        out["roofline_chain"] = {k: rc.get(k) for k in ("algorithmic_bytes_per_sample", "achieved", "peak", "unit", "frac")}
    cb = full.get("cpu_baseline")
    out["cpu_baseline"] = None if not cb else {**{k: cb.get(k) for k in ("value", "unit", "cores", "kind", "one_thread", "cpu_model")}, "sample": _short(cb.get("sample"), 200)}
    if "parity" in full:
        out["parity"] = _compact_parity(full["parity"])
    if "also" in full:
        out["also"] = [_compact_also(e) for e in full["also"]]
    out["full"] = FULL_LINE_FILE
    return out


def emit_line(full: dict, sub: bool = False) -> None:
    """Write the full form to bench_full.json (beside bench.py; a copy under gpurun_out/ where that directory is) and print the compact
    line as the last line of stdout. An `also` run of another bench.py (--sub) hands its full form to its parent on stdout instead and
    writes nothing."""
    text = json.dumps(full)
    if sub:
        print(text, flush=True)
        return
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, FULL_LINE_FILE), "w") as f:
                    f.write(text + "\n")
            except OSError:
                pass
    print(json.dumps(compact_line(full)), flush=True)



def drop_in_probe(stage):
    """Diagnosis (scripts/r06/s10.sh, profiles/r06/s10_summary.txt): with SS_BENCH_DROPIN_AT=<stage> the drop-in lines are measured at that point
    of the run — together with a few raw copies of a pageable 128 MiB buffer — and the run goes on. Without the variable: nothing."""
    if os.environ.get("SS_BENCH_DROPIN_AT") != stage:
        return
    import numpy as np
    import rtl_sdr_scanner_cpp_amd as pkg
    if True:
        if os.environ.get("SS_TRACE_PROCESS"):
            pkg.engine.use_diag_library(True)  # (the diagnostics build prints ss_process's phases)
        for e in drop_in_lines():
            print("PROBE", stage, {k: e.get(k) for k in ("fft_size", "ss_process_MSps", "ss_process_pieces_ms", "error")}, file=sys.stderr, flush=True)
        # ... is it the SOURCE? the same context fed from a synthetic batch (made by numpy arithmetic) and from one fresh allocation
        n_, nb_ = 1 << 20, 16
        band_ = pkg.synth.SyntheticBand(n_, seed=9, on_frame=40, off_frame=10_000)
        learn_, synth_ = band_.frames_cf32(32), band_.frames_cf32(nb_)
        fresh_ = np.empty((nb_, n_), np.complex64)
        fresh_[:] = synth_
        e_ = pkg.SpectrumEngine(61_440_000, 145_000_000, fft_size=n_, decim=1, in_format=0, learn_frames=32, max_batch=32)
        e_.process(learn_, want=())
        for name_, src_ in (("synthetic batch", synth_), ("fresh np.empty filled once", fresh_), ("synthetic batch again", synth_)):
            e_.process(src_, want=(), cand_cap=1 << 20)
            t0 = time.perf_counter()
            for _ in range(3):
                e_.process(src_, want=(), cand_cap=1 << 20)
            print("PROBE", stage, f"ss_process fed from {name_}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms", file=sys.stderr, flush=True)
        e_.close()
        # ... and a pageable 128 MiB buffer into device memory obtained NOW from hipMalloc, against memory torch's allocator holds
        import ctypes as C
        import torch
        hip = C.CDLL("libamdhip64.so")
        hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        hip.hipFree.argtypes = [C.c_void_p]
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        nbytes = 128 << 20
        host = np.ones(nbytes, np.uint8)
        fresh = C.c_void_p()
        assert hip.hipMalloc(C.byref(fresh), nbytes) == 0
        held = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        for name, ptr in (("fresh hipMalloc", fresh.value), ("torch's pool", held.data_ptr())):
            hip.hipMemcpy(ptr, host.ctypes.data, nbytes, 1)
            t0 = time.perf_counter()
            for _ in range(3):
                hip.hipMemcpy(ptr, host.ctypes.data, nbytes, 1)
            print("PROBE", stage, f"hipMemcpy of 128 MiB pageable into {name}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms", file=sys.stderr, flush=True)
        hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
        hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        for flag, name in ((1, "a new non-blocking stream"), (0, "a new blocking stream")):
            st = C.c_void_p()
            assert hip.hipStreamCreateWithFlags(C.byref(st), flag) == 0
            hip.hipMemcpyAsync(fresh.value, host.ctypes.data, nbytes, 1, st)
            hip.hipStreamSynchronize(st)
            t0 = time.perf_counter()
            for _ in range(3):
                hip.hipMemcpyAsync(fresh.value, host.ctypes.data, nbytes, 1, st)
                hip.hipStreamSynchronize(st)
            print("PROBE", stage, f"hipMemcpyAsync + hipStreamSynchronize on {name}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms", file=sys.stderr, flush=True)
        hip.hipFree(fresh)


# ---------------------------------------------------------------------------------------------- the measured job
def run(args):
    import numpy as np
    import rtl_sdr_scanner_cpp_amd as pkg
    from rtl_sdr_scanner_cpp_amd import dist

    _probe = drop_in_probe
    # RCCL ("nccl") over xGMI in production; SS_DIST_BACKEND=gloo lets the multi-rank path be exercised on a box
    # with fewer GPUs than ranks (ranks then share devices, results are functional only)
    backend = os.environ.get("SS_DIST_BACKEND", "nccl")
    rank, local_rank, world = dist.init(backend)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    n, fs, nb = args.fft, args.sample_rate, args.frames

    cfg0 = None
    if rank == 0:
        cfg0 = dict(fft_size=n, sample_rate=fs, decim=args.decim, in_format={"cf32": 0, "cs8": 1, "cu8": 2}[args.fmt], grouping_x=21, grouping_y=21,
                    start_level_mdB=int(round(args.start_level * 1000)), learn_frames=100, learn_ms=2000, max_batch=nb, band0_center=140_000_000,
                    band_spacing=max(2_000_000, fs), n_bands=world if args.shard == "bands" else 1, seed=0)

    if args.launch_check:
        import torch
        cfg = dist.broadcast_config(cfg0, device="cpu")
        dist.barrier()
        t0 = time.perf_counter()
        time.sleep(0.01 * (rank + 1))
        dist.barrier()
        elapsed = dist.max_over_ranks(time.perf_counter() - t0, device="cpu")
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "backend": backend, "fft_size": int(cfg["fft_size"]), "n_bands": int(cfg["n_bands"]),
                              "shard": args.shard, "elapsed_s": round(elapsed, 4)}), flush=True)
        if world > 1:
            import torch.distributed as td
            td.destroy_process_group()
        return

    import torch
    ndev = torch.cuda.device_count()
    if ndev == 0:
        raise SystemExit("bench.py needs an MI355X: the spectral-scan engine has no CPU path")
    plan = rank_plan(rank, local_rank, world, ndev, args, n)
    device_index = plan["device_index"]
    torch.cuda.set_device(device_index)
    if args.host_wait != "auto":
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so.7")  # (the runtime torch has loaded: same soname)
        rc = hip.hipSetDeviceFlags(ctypes.c_uint({"spin": 1, "yield": 2, "block": 4}[args.host_wait]))
        if rc != 0:
            raise RuntimeError(f"hipSetDeviceFlags({args.host_wait}) failed: {rc}")
    dev = torch.device("cuda", device_index)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")
    cfg = dist.broadcast_config(cfg0, device=coll_dev)  # the only collective of the whole job (RCCL, < 1 KiB)
    shard_frames = args.shard == "frames" and world > 1
    band = 0 if args.shard == "frames" else dist.bands_for_rank(int(cfg["n_bands"]), rank, world)[0]

    eng_kw = dict(fft_size=int(cfg["fft_size"]), decim=int(cfg["decim"]), in_format=int(cfg["in_format"]), grouping_x=int(cfg["grouping_x"]),
                  grouping_y=int(cfg["grouping_y"]), start_level=cfg["start_level_mdB"] / 1000.0, learn_frames=int(cfg["learn_frames"]),
                  max_batch=nb, device_id=device_index,
                  flags=(pkg.abi.SS_FLAG_SPECTROGRAM if args.spectrogram else 0) | (pkg.abi.SS_FLAG_NO_CULL if args.no_cull else 0))
    if args.diag_lib or args.lib:
        pkg.engine.use_diag_library(args.lib or True)
    _probe("before_engine")
    eng = pkg.SpectrumEngine(int(cfg["sample_rate"]), dist.band_center(cfg, band), **eng_kw)
    _probe("engine_created")

    # ---- working set: `nsets` distinct input batches and output sets in rotation, well past the Infinity Cache ----
    in_bytes = nb * n * args.decim * (8 if args.fmt == "cf32" else 2)
    out_bytes = nb * n * 4 * ((0 if args.no_psd_out else 1) + (2 if args.planes else 0))
    nsets = args.sets or max(6, -(-int(2.5 * L3_BYTES) // max(1, in_bytes + out_bytes)))
    nsets = min(nsets, 64)
    nsets = -(-nsets // 12) * 12  # a multiple of the library's launch queues (2, 3 or 4): every buffer reuse is then ordered by stream order alone (include/specscan.h)
    gen = dist.synthetic_stream(cfg, band)  # one continuous frame stream of the band: noise + gated wide-band transmissions
    # the learning frames (the reference's default: 100 frames — NOISE_LEARNING_TIME = 2000 ms at 50 frames per second — whatever the
    # batch size: calls of fewer frames learn over several calls); every rank of a frame-sharded band learns from these same frames
    n_first = -(-int(cfg["learn_frames"]) // nb)
    to_dev = lambda a: torch.from_numpy(a.view(np.float32) if a.dtype == np.complex64 else a).to(dev)  # noqa: E731
    d_first = [to_dev(gen(nb)) for _ in range(n_first)]
    halo_frames = 0
    if shard_frames:
        # rank r owns frames [lo, lo + (warmup + steps) * nb) of the stream; it re-reads the halo in front of its range after
        # an averager reset (dist.scan_frame_range: the tile boundary at or below lo - 20, so 32-47 frames), nothing is exchanged
        lo = n_first * nb + rank * (args.warmup + args.steps) * nb
        start = ((lo // 16) * 16 - 20) // 16 * 16
        halo_frames = lo - start
        gen = dist.synthetic_stream(cfg, band, start_frame=start, stream_seed=1000 + rank)
    d_halo = [to_dev(gen(min(nb, halo_frames - p))) for p in range(0, halo_frames, nb)]
    base = [to_dev(gen(nb)) for _ in range(min(nsets, 4))]
    # further batches: the generated ones with their frames rotated (distinct memory, same statistics)
    d_iq = [base[k] if k < len(base) else torch.roll(base[k % len(base)], shifts=37 * k, dims=0).contiguous() for k in range(nsets)]
    cap = nb * 1024
    nout = nsets  # (the stages of up to five consecutive calls are in flight at once: every call has its own output set)
    outs = [dict(psd=None if args.no_psd_out else torch.empty((nb, n), dtype=torch.float32, device=dev), off=torch.zeros(nb + 1, dtype=torch.int32, device=dev),
                 idx=torch.empty(cap, dtype=torch.int32, device=dev), avg=torch.empty(cap, dtype=torch.float32, device=dev),
                 rel_plane=torch.empty((nb, n), dtype=torch.float32, device=dev) if args.planes else None,
                 avg_plane=torch.empty((nb, n), dtype=torch.float32, device=dev) if args.planes else None)
            for _ in range(nout)]
    torch.cuda.synchronize()
    _probe("working_set")
    counter = [0]

    def step(iq=None):
        k = counter[0]
        counter[0] += 1
        o = outs[k % nout]
        src = d_iq[k % nsets] if iq is None else iq
        eng.process_device(src, src.shape[0], psd=o["psd"], rel=o["rel_plane"], avg=o["avg_plane"], cand_off=o["off"], cand_idx=o["idx"], cand_avg=o["avg"])
        if args.sync_every_step:
            eng.sync()

    for d in d_first:  # noise learning (identical on every rank of a frame-sharded band)
        step(d)
    if shard_frames and rank > 0:
        eng.sync()
        eng.reset()
        for h in d_halo:
            step(h)
    preheat_steps = 0
    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < args.preheat_ms:  # setup, like the learning phase: clocks and caches settle
        for _ in range(50):
            step()
        eng.sync()
        preheat_steps += 50
    _probe("learned")
    for _ in range(max(args.warmup, 1)):
        step()
    eng.sync()
    _probe("warmed_up")
    # No event-timed launch inside the timed region (round 6): each one costs its queue ~13 us (the start packet ~7 us before it, the stop
    # packet ~6 us before the queue's next launch: profiles/r03/s37_timeline_k20.txt) — one was 2.7 % of the driver's 20-step run, and a
    # chain of two launches per call with every 8th call sampled paid 3 us per call (5 % of a 65 us call of 262144 points). The launches
    # are timed in 32 more steps right behind the timed region instead (below); --time-every k puts events on every k-th call inside it.
    every = 0
    if not args.no_kernel_timing and args.time_every > 0:
        every = args.time_every
        eng.kernel_timing(every)
    import gc
    gc.collect()
    if not os.environ.get("SS_BENCH_KEEP_GC"):  # (A/B: scripts/r06/s27.sh)
        gc.disable()  # (as timeit does: no collector pause of the interpreter inside a region of half a millisecond)
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    eng.flush()  # the deferred detect / emit stages of the last two steps belong to the timed work
    t_enq = time.perf_counter()
    if args.sync_engine_first:  # (round 2's order: the library's own wait first, then the contract's device-wide one on an idle device — two wake-ups)
        eng.sync()
    t_eng = time.perf_counter()
    torch.cuda.synchronize()  # the contract's synchronisation IS the wait: hipDeviceSynchronize covers every stream of the device, the library's queues included
    t_dev = time.perf_counter()
    dist.barrier()
    t1 = time.perf_counter()
    gc.enable()
    eng.sync()  # (settles the library's own bookkeeping; nothing left to wait for)
    kern_ms, launches, slots = 0.0, 0, {}
    if every:
        slots = eng.kernel_timing_read_frames()  # {slot: (ms, launches, frames those launches covered)}
        kern_ms, launches, _ = slots["step"]
        eng.kernel_timing(0)
    # Right behind the timed region — same clocks, same working set, nothing of it inside — 32 more steps with every 4th call's launches
    # timed give eight samples per kernel of the chain; they are what roofline.kernel_us is made of.
    kern_after = None
    if not args.no_kernel_timing and not every:
        eng.kernel_timing(4)
        for _ in range(32):
            step()
        eng.sync()
        slots = eng.kernel_timing_read_frames()
        kern_ms, launches, _ = slots["step"]
        eng.kernel_timing(0)
        every = 4
        if launches:
            kern_after = {"us": round(kern_ms / launches * 1e3, 2), "launches": launches, "what": "32 more steps right behind the timed region, every 4th call's launches timed"}
    # A rank's interval ends when ITS device is idle (t_dev, behind the contract's synchronisation); the barrier behind it brackets the region,
    # and the job's time is the slowest rank's: MAX over ranks. (Until round 6 the interval ran to t1, behind the barrier — at N = 1 the same
    # thing, at N > 1 it put the collective's own latency, 30-100 us of a 500 us region, into every rank's time: an artefact in the scaling curve.)
    elapsed = dist.max_over_ranks(t_dev - t0, device=coll_dev)
    devices_of_ranks = dist.ints_of_ranks(device_index, device=coll_dev)  # (rank r -> the device it worked on: LOCAL_RANK modulo the box's device count)
    cands_of_ranks = dist.ints_of_ranks(int(outs[(counter[0] - 1) % nout]["off"][-1].item()), device=coll_dev)
    ncand = int(outs[(counter[0] - 1) % nout]["off"][-1].item())
    _probe("timed")
    lib_stats = eng.stats()  # what the library says it did (counters from creation: learning, preheat, warm-up and the timed steps)
    _probe("stats")

    if rank == 0:
        samples_per_step = nb * n * world
        value = samples_per_step * args.steps / elapsed / 1e6
        kern_avg_s = kern_ms / max(launches, 1) / 1e3
        step_s = elapsed / args.steps
        # every launch of the chain, kernel by kernel (start/stop events on the launches of the sampled calls)
        kernels = kernel_tally(chain_kernels(n, args.fmt, nb, args.no_psd_out and not args.planes), slots, n, nb)
        for k in kernels:
            slot, match = k["slot"], k.pop("match")
            # launches of the steady-state shape only. 8192 points: 1024 + 20 FFT, 128 emit and 4 plan workgroups of 512 threads; long
            # transforms: the column tiles + one emit workgroup per frame (the listed tiles ride on the column workgroups)
            two_pass = n == 1 << 20 and os.environ.get("SS_FFT_TWOPASS") != "0"
            fold = n in (65536, 131072) and args.fmt != "cf32" and args.no_psd_out and not args.planes and os.environ.get("SS_DIF8") != "0"
            if n == 8192:
                shape = (nb + 20 + nb // 8 + 4) * 512
            elif fold:
                shape = None  # (one launch per call whatever its length: every launch of the kernel in the committed pass is one)
            elif two_pass:  # column half: 64 workgroups of 1024 threads per frame behind the 128 that run the plan of the call before; row half: 128 of 512 per frame + one emit workgroup per frame
                shape = {"step": (nb * 64 + 128) * 1024, "rows": (nb * 128 + nb + 64) * 512}.get(slot)  # (+ 64 detect workgroups for the first listed pairs)
            elif n == 65536 and os.environ.get("SS_CULL_65536") != "0" and os.environ.get("SS_ROWS256_STEP") != "0" and os.environ.get("SS_MERGE_65536") != "0" and nb <= 128:
                # one launch per call: 8 column tiles and 8 row tiles per frame + 128 plan + one emit workgroup per frame + 64 detect workgroups
                shape = {"step": (nb * 16 + 128 + nb + 64) * 512}.get(slot)
            elif n == 65536 and os.environ.get("SS_CULL_65536") != "0" and os.environ.get("SS_ROWS256_STEP") != "0":
                # column launch: 8 column tiles per frame + 128 plan + one emit workgroup per frame + 64 detect workgroups; row launch: 8 row tiles per frame
                shape = {"step": (nb * 8 + 128 + nb + 64) * 512, "rows": nb * 8 * 512}.get(slot)
            else:
                shape = (nb * (n // 8192) + (nb if n >= 65536 else -(-nb // 8))) * 512 if slot == "step" and n >= 16384 else None
            tp = traffic_from_profiles(args.config or 2, match, shape) if is_preset(args) else None
            k["pmc_bytes_per_launch_from_profiles"] = tp["bytes_per_launch"] if tp else None
        dom = max(kernels, key=lambda k: k["us"] * k["launches_per_call"]) if kernels else None
        abps = algo_bytes_per_sample(args.fmt, True)  # the FFT kernel always writes its dB row
        chain_bps = algo_bytes_per_sample(args.fmt, not args.no_psd_out) + (8.0 if args.planes else 0.0)
        # ALGORITHMIC bytes per launch / mean launch duration; for the long transforms the dominant kernel of the chain
        # (its share of what the DESIGN moves — work buffers included — is roofline.kernels[].gbs, not this)
        literal = abps * nb * n / kern_avg_s / 1e9 if (launches and n == 8192) else (chain_bps * dom["frames_per_launch"] * n / dom["us"] / 1e3 if dom and dom["us"] else None)
        # Consecutive launches of k_scan_step overlap on two hardware queues (deep pipelining): a launch lasts about
        # twice as long as the GPU spends per launch. in_flight = mean launch duration / wall time per launch; the kernel's
        # achieved rate is its bytes over duration / in_flight (= the literal figure when launches do not overlap).
        # (Other sizes: launches in order on one stream, nothing overlaps.)
        in_flight = max(1.0, kern_avg_s / step_s) if (launches and n == 8192) else 1.0
        achieved = literal * in_flight if literal is not None else None
        chain_gbs = chain_bps * nb * n / step_s / 1e9  # per GPU
        pmc_chain = [k["pmc_bytes_per_launch_from_profiles"] * k["launches_per_call"] for k in kernels if k["pmc_bytes_per_launch_from_profiles"]]
        out = {
            "metric": "iq_msamples_per_sec_scanned_8192pt_fft" if n == 8192 else f"iq_msamples_per_sec_scanned_{n}pt_fft",
            "value": round(value, 1), "unit": "MS/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(step_s * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{n}-pt FFT, {fs / 1e6:.3f} MS/s, {nb}-frame batches, {args.fmt.upper()} IQ resident in HBM, full chain "
                                   "(window+FFT+dB -> noise-relative -> 21x21 mean -> threshold -> candidate lists), "
                                   + ("one band per GPU" if not shard_frames else "one band, a contiguous frame range per GPU (halo re-read, no exchange)"),
                       "baseline_config": args.config or 2, "fft_size": n, "frames_per_batch": nb, "bands": int(cfg["n_bands"]), "shard": args.shard if world > 1 else None,
                       "halo_frames": halo_frames, "candidates_per_batch": ncand,
                       "spectrogram_branch": bool(args.spectrogram), "psd_plane_out": not args.no_psd_out, "rel_avg_planes_out": bool(args.planes), "frame_decimation": args.decim, "sync_every_step": bool(args.sync_every_step), "host_wait": args.host_wait, "diag_lib": bool(args.diag_lib or args.lib),
                       # read back from the library (ss_get_stats), not mirrored from its policy: is tile culling on for this context, and of the
                       # averaging tiles of every batch so far how many went through the test and how many were proven empty and never evaluated
                       "tile_culling": bool(lib_stats["culling"]),
                       "tiles": {"total": lib_stats["tiles_total"], "tested": lib_stats["tiles_tested"], "culled": lib_stats["tiles_culled"],
                                 "evaluated_frac": round(1.0 - lib_stats["tiles_culled"] / max(lib_stats["tiles_total"], 1), 4), "wait_fallbacks": lib_stats["wait_fallbacks"]},
                       "calls": {"overlapped": lib_stats["calls_overlapped"], "in_order": lib_stats["calls_in_order"], "drains": lib_stats["drains"], "demotions": lib_stats["demotions"]},
                       "preheat_steps": preheat_steps, "input_sets": nsets, "output_sets": nout, "working_set_mib": round((nsets * in_bytes + nout * out_bytes) / 2**20, 1),
                       "dist_backend": backend if world > 1 else None, "ranks_share_devices": bool(world > ndev),
                       "device_index_of_ranks": devices_of_ranks, "candidates_last_batch_of_ranks": cands_of_ranks,
                       "host_enqueue_ms_per_step": round((t_enq - t0) / args.steps * 1e3, 4),
                       # where the end of the timed region goes: the chain's own drain + wait, then the contract's device-wide synchronisation and barrier
                       "tail_us": {"engine_sync": round((t_eng - t_enq) * 1e6, 1) if args.sync_engine_first else None, "device_synchronize": round((t_dev - t_eng) * 1e6, 1), "barrier": round((t1 - t_dev) * 1e6, 1)}},
            "roofline": {"bound": "hbm", "kernel": dom["what"] if dom else None,
                         "achieved": None if achieved is None else round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": None if achieved is None else round(achieved / HBM_PEAK_GBS, 4),
                         "kernel_us": dom["us"] if dom else None, "launches": dom["launches_timed"] if dom else 0,
                         "kernel_timing": (kern_after["what"] if kern_after else f"start/stop events on every {every}th launch inside the timed region") if every else None,
                         "launches_in_flight": round(in_flight, 2) if dom else None,
                         "achieved_if_launches_did_not_overlap": None if literal is None else round(literal, 1),
                         # SURVEY.md 8d's figure — IQ in (+ the dB row in power mode) — times the samples one launch of the dominant kernel covers; and,
                         # for the long transforms, what the DESIGN makes that kernel move (its inputs once + its outputs once: work buffers included)
                         "algorithmic_bytes_per_launch": abps * nb * n if n == 8192 else chain_bps * (dom["frames_per_launch"] if dom else nb) * n,
                         "design_bytes_per_launch": dom["bytes_per_launch_it_must_move"] if dom else None,
                         "achieved_on_design_bytes": dom["gbs"] if dom else None,
                         "traffic": None,  # (filled in below by two short rocprofv3 --pmc passes of this command line on this box: live_pmc_traffic) ...
                         # ... the committed passes of the same command line (for 8192 points: launches of the steady-state shape, 1024 + 20 FFT, 128 emit and 4 plan workgroups of 512 threads)
                         "traffic_from_profiles": ({"bytes_per_launch": dom["pmc_bytes_per_launch_from_profiles"]} if dom and dom["pmc_bytes_per_launch_from_profiles"] else None),
                         "kernels": kernels},
            "roofline_chain": {"bound": "hbm", "what": "whole step (every kernel of the chain + launch gaps), per GPU",
                               "algorithmic_bytes_per_sample": chain_bps, "achieved": round(chain_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(chain_gbs / HBM_PEAK_GBS, 4),
                               "pmc_bytes_per_sample_from_profiles": round(sum(pmc_chain) / (nb * n), 2) if pmc_chain else None},
        }
        if plan["live_pmc"] and dom:
            sub_argv = ["--gpus", "1", "--steps", "30", "--warmup", "5", "--preheat-ms", "0", "--no-cpu-baseline", "--no-also", "--no-parity", "--no-live-pmc", "--sub"]
            try:
                live, why = live_pmc_traffic(sub_argv, "k_scan_step", (nb + 20 + nb // 8 + 4) * 512)
            except Exception as e:  # (the line must come out whatever the profiler does)
                live, why = None, f"{type(e).__name__}: {e}"[:200]
            out["roofline"]["traffic"] = live["bytes_per_launch"] if live else None
            out["roofline"]["traffic_live"] = live if live else {"unavailable": why}
            if live:
                out["roofline"]["traffic_over_algorithmic"] = round(live["bytes_per_launch"] / (abps * nb * n), 3)
                # what the memory system really moves per second of the timed region: the counted bytes of a launch over the wall time a
                # launch takes (= ms_per_step: one launch per step) — beside `achieved`, which counts the algorithmic bytes only
                out["roofline"]["traffic_gbs"] = round(live["bytes_per_launch"] / step_s / 1e9, 1)
                out["roofline"]["traffic_frac_of_peak"] = round(live["bytes_per_launch"] / step_s / 1e9 / HBM_PEAK_GBS, 4)
        if plan["also"]:
            out["also"] = [] if os.environ.get("SS_BENCH_SKIP_ALSO_RUNS") else also_lines(args.also_all)  # (the switch: diagnosing the drop-in lines alone, scripts/r06/s10.sh)
            _probe("before_close")
            eng.close()  # (the drop-in lines below make contexts of their own; this one's memory goes back first)
            _probe("closed")
            out["also"] += drop_in_lines()
        if plan["cpu_baseline"]:
            out["cpu_baseline"] = cpu_baseline(n, fs, args.cpu_seconds)
        elif world == 1:
            out["cpu_baseline"] = None
        if plan["parity"]:  # (the `also` runs drop the CPU timing, not the parity sample)
            try:
                out["parity"] = parity_sample(n, fs, args.fmt, args.no_cull, nb)
            except AssertionError as e:
                out["parity"] = {"failed": str(e)[:300]}
        emit_line(out, sub=args.sub)

    if world > 1:
        import torch.distributed as td
        td.destroy_process_group()


def file_fed_cpu_chain(n: int, fs: int, budget_s: float):
    """BASELINE config 1 as it is worded — a synthetic 2.048 MS/s IQ *file*: a dump named and laid out like the reference's own
    (`./full_<date>_<time>_<centre>_<rate>_fc.raw`, utils/radio_utils.cpp:78-84, written through the FileSink restatement as
    sdr_device.cpp:173-181 does) is read back by RawIqReader — items of N*D samples, the first N of each handed on, the
    Decimator's output (decimator.h:15-22) — and scanned by ONE chain of the reference's own code on one thread, as one band is
    one chain. Returns the scanned and the ingested rate."""
    import tempfile
    import numpy as np
    import rtl_sdr_scanner_cpp_amd as pkg
    from rtl_sdr_scanner_cpp_amd import replay
    from oracle import oracle as O
    center, decim = 145_000_000, max(1, int(fs / n / 50))  # D = 5 at 2.048 MS/s (ss_default_config)
    items, chunk = 1200, 60
    band = pkg.synth.SyntheticBand(n, decim=decim, seed=7, on_frame=130, off_frame=900)
    use_ref = O.have_ref()
    backend = 2 if (O.ref() if use_ref else O.lib()).orc_set_fft_backend(2) == 0 else 0
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, replay.make_raw_file_name("full", "fc", center, fs)[2:])
        sink = replay.RawFileSink(8)  # FileSink<gr_complex>
        sink.start_recording(path)
        for _ in range(items // chunk):
            sink.work(band.frames_cf32(chunk).reshape(-1))
        sink.stop_recording()
        sink.close()
        size = os.path.getsize(path)
        frames = np.empty((chunk, n), np.complex64)
        chain = O.RefChain(n, fs, center - fs // 2, center + fs // 2) if use_ref else O.oracle_chain(fs, center, fft_size=n, decim=1, max_batch=chunk)
        scanned, passes, t_ms = 0, 0, 0
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < budget_s:
            reader = replay.RawIqReader(path, replay.KIND_CF32, n, decim)
            while True:
                got = reader.read_into(frames, chunk)
                if got == 0:
                    break
                if use_ref:
                    chain.process(frames[:got], t_ms + 20 * np.arange(got))
                    t_ms += 20 * got
                else:
                    chain.process(frames[:got], want=(), cand_cap=chunk * n)
                scanned += got
            reader.close()
            passes += 1
        el = time.perf_counter() - t0
    return {"value": round(scanned * n / el / 1e6, 3), "unit": "MS/s", "ingested_msps": round(scanned * n * decim / el / 1e6, 3), "threads": 1,
            "frame_decimation": decim, "file_bytes": size, "items_in_file": items, "passes_over_the_file": passes, "seconds": round(el, 2),
            "kind": "reference" if use_ref else "port", "fft": "MKL FFTW3 interface" if backend == 2 else "built-in radix-2",
            "path": "RawFileSink -> ./full_*_fc.raw -> RawIqReader (first N of each N*D item) -> one chain of the reference's code"}


def run_cpu_only(args):
    """BASELINE config 1: the reference's CPU path on a synthetic 2.048 MS/s IQ file, no GPU. `value` is the
    file-fed single chain; cpu_baseline beside it is the same code on in-memory frames, one thread and all threads."""
    n, fs = args.fft, args.sample_rate
    fed = file_fed_cpu_chain(n, fs, max(4.0, args.cpu_seconds / 2))
    cb = cpu_baseline(n, fs, args.cpu_seconds, stages=True)
    out = {"metric": "iq_msamples_per_sec_scanned_8192pt_fft" if n == 8192 else f"iq_msamples_per_sec_scanned_{n}pt_fft",
           "value": fed["value"], "unit": "MS/s", "n_gpus": 0, "steps": 0, "warmup": 0, "ms_per_step": None, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"synthetic {fs / 1e6:.3f} MS/s IQ file, {n}-pt FFT, single band, CPU reference path (no GPU), one chain on one thread", "baseline_config": 1,
                      "file_fed": fed},
           "roofline": None, "cpu_baseline": cb}
    emit_line(out)


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.cpu_only:
        return run_cpu_only(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if args.config and "--gpus" not in " ".join(argv):
            argv = [*argv, "--gpus", str(args.gpus)]
        raise SystemExit(self_launch(args, argv))
    run(args)


if __name__ == "__main__":
    main()
