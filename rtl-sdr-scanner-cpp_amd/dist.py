"""One process per GPU. Bands (independent scan chains, one per SDR device in the reference:
sources/main.cpp:50-59) shard across ranks with no data-path collective; the only exchange is a
broadcast of the POD scan configuration from rank 0 at start-up (RCCL over xGMI when the backend is
"nccl", gloo in the CPU tests), mirroring the config reload in sources/main.cpp:37-49."""
from __future__ import annotations

import os

import numpy as np

CONFIG_FIELDS = ("fft_size", "sample_rate", "decim", "in_format", "grouping_x", "grouping_y", "start_level_mdB",
                 "learn_frames", "learn_ms", "max_batch", "band0_center", "band_spacing", "n_bands", "seed")


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend: str):
    """init_process_group from the torchrun environment; returns (rank, local_rank, world). No-op at world 1."""
    rank, local_rank, world = env_world()
    if world > 1:
        import torch.distributed as td
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if not td.is_initialized():
            td.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def broadcast_config(cfg: dict | None, device="cpu") -> dict:
    """Rank 0 passes the dict, everyone gets it back. int64 POD vector, < 1 KiB."""
    import torch
    import torch.distributed as td
    if not (td.is_available() and td.is_initialized()) or td.get_world_size() == 1:
        return dict(cfg)
    buf = torch.zeros(len(CONFIG_FIELDS), dtype=torch.int64, device=device)
    if td.get_rank() == 0:
        buf = torch.tensor([int(cfg[k]) for k in CONFIG_FIELDS], dtype=torch.int64, device=device)
    td.broadcast(buf, src=0)
    vals = buf.cpu().tolist()
    return dict(zip(CONFIG_FIELDS, vals))


def bands_for_rank(n_bands: int, rank: int, world: int) -> list[int]:
    """band b -> rank b mod world."""
    return [b for b in range(n_bands) if b % world == rank]


def frame_ranges(nframes: int, rank: int, world: int, halo: int):
    """Contiguous frame range of one band for this rank plus the `halo` frames before it that must be
    re-read so the 21-frame averager window is complete (no exchange between ranks)."""
    per = (nframes + world - 1) // world
    lo, hi = min(nframes, rank * per), min(nframes, (rank + 1) * per)
    return max(0, lo - halo), lo, hi


def max_over_ranks(seconds: float, device="cpu") -> float:
    import torch
    import torch.distributed as td
    if not (td.is_available() and td.is_initialized()) or td.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    td.all_reduce(t, op=td.ReduceOp.MAX)
    return float(t.item())


def ints_of_ranks(value: int, device="cpu") -> list[int]:
    """One integer per rank, known to every rank afterwards (a sum over one-hot vectors: a few bytes, outside any timed region)."""
    import torch
    import torch.distributed as td
    if not (td.is_available() and td.is_initialized()) or td.get_world_size() == 1:
        return [int(value)]
    t = torch.zeros(td.get_world_size(), dtype=torch.int64, device=device)
    t[td.get_rank()] = int(value)
    td.all_reduce(t, op=td.ReduceOp.SUM)
    return [int(v) for v in t.cpu().tolist()]


def barrier():
    import torch.distributed as td
    if td.is_available() and td.is_initialized() and td.get_world_size() > 1:
        td.barrier()


def band_center(cfg: dict, band: int) -> int:
    return int(cfg["band0_center"]) + band * int(cfg["band_spacing"])


def synthetic_batch(cfg: dict, band: int, nframes: int) -> np.ndarray:
    from . import synth
    b = synth.SyntheticBand(int(cfg["fft_size"]), decim=int(cfg["decim"]), seed=int(cfg["seed"]) + band,
                            on_frame=int(cfg["learn_frames"]) + 30, off_frame=nframes - 50)
    fmt = int(cfg.get("in_format", 0))
    return b.frames_cf32(nframes) if fmt == 0 else (b.frames_cs8(nframes) if fmt == 1 else b.frames_cu8(nframes))


def synthetic_stream(cfg: dict, band: int, start_frame: int = 0, stream_seed: int | None = None):
    """A continuous frame stream of one band for the benchmark: `gen(nframes)` returns the next frames in the band's sample
    format. Transmissions switch on 30 frames after the learning phase and then toggle every 400 frames. The comb phases
    come from the band's seed (the same transmitters for every rank that scans the band), the noise from `stream_seed`."""
    from . import synth
    learn = int(cfg["learn_frames"])
    seed = int(cfg["seed"]) + band
    b = synth.SyntheticBand(int(cfg["fft_size"]), decim=int(cfg["decim"]), seed=seed, on_frame=learn + 30, off_frame=learn + 430, period=800,
                            start_frame=start_frame)
    if stream_seed is not None:
        b.rng = np.random.default_rng(stream_seed)
    fmt = int(cfg.get("in_format", 0))

    def gen(nframes: int) -> np.ndarray:
        return b.frames_cf32(nframes) if fmt == 0 else (b.frames_cs8(nframes) if fmt == 1 else b.frames_cu8(nframes))

    return gen


def scan_frame_range(chain, frames, lo: int, hi: int, learn_frames: int, max_batch: int, align: int = 16, halo: int = 20):
    """One rank's share of a recorded band, frames [lo, hi) of ``frames`` ([nframes, N*D] items), with no exchange between
    ranks (BASELINE config 5): every rank first runs the learning prefix [0, learn_frames) itself so that
    all ranks hold the same noise ceiling, resets the averager, and re-reads a halo before its range — the engine restarts
    its sliding sums every ``align`` = 16 frames since the last reset, so a frame's mean depends on the 20 frames
    (GROUPING_Y - 1) before the FIRST frame of its tile: the halo starts at the tile boundary at or below
    floor(lo / 16) * 16 - 20, which also puts the rank's tiles where they fall in a single-rank run. The outputs of halo
    frames are dropped. Returns per-frame candidate lists for [lo, hi), bit-identical to a single-rank scan
    for every frame past the averager's warm-up (lo >= learn_frames + halo)."""
    if lo >= hi:
        return []
    pos = 0
    while pos < learn_frames:  # noise ceiling: identical on every rank
        n = min(max_batch, learn_frames - pos)
        chain.process(frames[pos:pos + n], want=())
        pos += n
    # a frame's time mean slides from the first frame of its tile, whose own sum reaches `halo` frames further back
    start = max(learn_frames, (((lo // align) * align - halo) // align) * align)
    if start > learn_frames:
        chain.reset()  # Averager::reset: ring and frame counter to zero; the engine's tile origin moves to `start`
    # frames of the range that lie inside the learning prefix never have candidates (noise_learner.cpp:45-51)
    out = [np.zeros(0, np.int32) for _ in range(max(0, min(hi, learn_frames) - lo))]
    pos = start
    while pos < hi:
        n = min(max_batch, hi - pos)
        r = chain.process(frames[pos:pos + n], want=())
        off, idx = r["cand_off"], r["cand_idx"]
        for f in range(n):
            if pos + f >= max(lo, learn_frames):
                out.append(idx[off[f]:off[f + 1]].copy())
        pos += n
    return out
