"""The multi-rank GPU path on the one device a test box has: `bench.py --gpus 2` starts two rank processes
(torch.distributed.run, gloo because the box has fewer GPUs than ranks), each creates its own HIP context and ss_ctx on device
LOCAL_RANK % device_count, allocates its own working set in HBM, takes the POD configuration from rank 0's broadcast and scans ITS
band; rank 0 prints ONE JSON line with the max-over-ranks time. No 8-GPU node was ever offered to a round, so this is the closest a
test gets to the driver's N > 1 runs: two real HIP contexts in two rank processes with GPU work in both."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(argv, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["SS_DIST_BACKEND"] = "gloo"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]  # ONE JSON line, from rank 0
    assert len(lines[0]) < 6000
    return json.loads(lines[0])


def test_two_ranks_on_one_device_scan_their_own_bands():
    import torch
    ndev = torch.cuda.device_count()
    d = _run(["--gpus", "2", "--steps", "10", "--warmup", "3", "--preheat-ms", "50", "--no-cpu-baseline", "--no-also", "--no-parity"])
    c = d["config"]
    assert d["n_gpus"] == 2 and d["steps"] == 10 and d["scaling"] == "weak" and d["value"] > 0
    assert c["bands"] == 2 and c["dist_backend"] == "gloo" and c["fft_size"] == 8192 and c["frames_per_batch"] == 1024
    assert c["ranks_share_devices"] is (ndev < 2)
    assert c["device_index_of_ranks"] == [0 % ndev, 1 % ndev]  # rank r -> device LOCAL_RANK modulo the box's device count
    assert len(c["candidates_last_batch_of_ranks"]) == 2 and all(n > 1000 for n in c["candidates_last_batch_of_ranks"])  # both bands were really scanned
    assert c["candidates_last_batch_of_ranks"][0] != c["candidates_last_batch_of_ranks"][1]  # ... and they are different bands
    assert "cpu_baseline" not in d or d["cpu_baseline"] is None  # the side legs are rank 0's at N = 1 only
    assert "also" not in d and "parity" not in d
    # whole-job value: the samples of BOTH ranks over the max-over-ranks time
    assert abs(d["value"] - 2 * 1024 * 8192 / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 0.01


def test_two_ranks_share_one_band_by_frame_ranges():
    """config 5's sharding on a size that is quick: one band, a contiguous frame range per rank, every rank learns from the same
    prefix and re-reads a halo (no exchange)."""
    d = _run(["--gpus", "2", "--shard", "frames", "--fft", "65536", "--frames", "64", "--fmt", "cs8", "--no-psd-out", "--steps", "8", "--warmup", "2", "--preheat-ms", "50",
              "--no-cpu-baseline", "--no-also", "--no-parity"])
    c = d["config"]
    assert d["n_gpus"] == 2 and c["bands"] == 1 and c["shard"] == "frames" and c["fft_size"] == 65536
    assert len(c["device_index_of_ranks"]) == 2 and d["value"] > 0
