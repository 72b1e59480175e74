"""The N>1 path on CPU: two processes, gloo backend. Exercises exactly what bench.py does around the
GPU work — config broadcast from rank 0 (the job's only collective), band sharding, barrier and the
max-over-ranks timing reduction — with the oracle standing in for the GPU so that each rank really
scans a different band."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, os.environ["SS_ROOT"])
    import numpy as np
    import rtl_sdr_scanner_cpp_amd as pkg
    from rtl_sdr_scanner_cpp_amd import dist
    from oracle import oracle as O
    rank, local_rank, world = dist.init("gloo")
    cfg0 = None
    if rank == 0:
        cfg0 = dict(fft_size=256, sample_rate=64000, decim=1, in_format=0, grouping_x=21, grouping_y=21, start_level_mdB=8000,
                    learn_frames=20, learn_ms=2000, max_batch=128, band0_center=140_000_000, band_spacing=2_000_000, n_bands=4, seed=5)
    cfg = dist.broadcast_config(cfg0)
    bands = dist.bands_for_rank(int(cfg["n_bands"]), rank, world)
    out = {"rank": rank, "cfg": cfg, "bands": bands, "cands": {}}
    dist.barrier()
    t0 = time.perf_counter()
    for b in bands:
        iq = dist.synthetic_batch(cfg, b, 128)
        ch = O.oracle_chain(int(cfg["sample_rate"]), dist.band_center(cfg, b), fft_size=int(cfg["fft_size"]), decim=1,
                            learn_frames=int(cfg["learn_frames"]), max_batch=128, start_level=cfg["start_level_mdB"] / 1000.0)
        out["cands"][str(b)] = int(ch.process(iq, want=())["cand_off"][-1])
    mine = time.perf_counter() - t0 + (0.25 if rank == 1 else 0.0)
    out["mine"], out["max"] = mine, dist.max_over_ranks(mine)
    out["frames"] = dist.frame_ranges(1000, rank, world, 20)
    # frame-range sharding of ONE band (SURVEY.md 8e-2): this rank's contiguous range with a re-read halo, no exchange
    nfr = 400
    rec = dist.synthetic_batch(cfg, 0, nfr)
    mk = lambda: O.oracle_chain(int(cfg["sample_rate"]), dist.band_center(cfg, 0), fft_size=int(cfg["fft_size"]), decim=1,
                                learn_frames=int(cfg["learn_frames"]), max_batch=128, start_level=cfg["start_level_mdB"] / 1000.0)
    _, lo, hi = dist.frame_ranges(nfr, rank, world, 20)
    mine_lists = dist.scan_frame_range(mk(), rec, lo, hi, int(cfg["learn_frames"]), 128)
    whole = dist.scan_frame_range(mk(), rec, 0, nfr, int(cfg["learn_frames"]), 128)
    same = sum(int(np.array_equal(a, b)) for a, b in zip(mine_lists, whole[lo:hi]))
    out["shard"] = {"lo": lo, "hi": hi, "frames_equal": same, "cands": int(sum(len(c) for c in mine_lists))}
    print("RESULT " + json.dumps(out), flush=True)
""")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_band_sharding_over_gloo(oracle_mod, tmp_path):
    import json
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   SS_ROOT=ROOT)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    res = []
    for p in procs:
        so, se = p.communicate(timeout=180)
        assert p.returncode == 0, se[-2000:]
        res.append(json.loads([ln for ln in so.splitlines() if ln.startswith("RESULT ")][0][7:]))
    res.sort(key=lambda r: r["rank"])
    assert res[0]["cfg"] == res[1]["cfg"] and res[1]["cfg"]["fft_size"] == 256 and res[1]["cfg"]["seed"] == 5  # broadcast reached rank 1
    assert res[0]["bands"] == [0, 2] and res[1]["bands"] == [1, 3]  # band b -> rank b mod world, disjoint and complete
    assert all(v > 0 for r in res for v in r["cands"].values())  # every band was really scanned
    assert abs(res[0]["max"] - res[1]["max"]) < 1e-9 and res[0]["max"] >= max(res[0]["mine"], res[1]["mine"]) - 1e-9
    assert res[0]["frames"] == [0, 0, 500] and res[1]["frames"] == [480, 500, 1000]  # 20-frame halo re-read, no exchange
    # the two ranges tile the recording, and every rank reproduces the single-rank lists of its frames (the oracle's
    # never-re-zeroed running sums may flip a bin at the threshold when restarted from a halo: allow a handful)
    assert (res[0]["shard"]["lo"], res[0]["shard"]["hi"], res[1]["shard"]["lo"], res[1]["shard"]["hi"]) == (0, 200, 200, 400)
    for r in res:
        assert r["shard"]["frames_equal"] >= (r["shard"]["hi"] - r["shard"]["lo"]) - 3, r["shard"]
    assert res[0]["shard"]["cands"] + res[1]["shard"]["cands"] > 500


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` without a launcher re-executes itself under torch.distributed.run (one rank per GPU; gloo
    where the box has fewer GPUs than ranks): launcher, rendezvous on 127.0.0.1, config broadcast from rank 0, barrier and
    max-over-ranks timing are exercised end to end by --launch-check, which stops short of the GPU work."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    for argv, want in ((["--gpus", "2", "--launch-check"], dict(n_gpus=2, n_bands=2, shard="bands", fft_size=8192)),
                       (["--config", "5", "--gpus", "4", "--launch-check"], dict(n_gpus=4, n_bands=1, shard="frames", fft_size=1 << 20))):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout  # ONE JSON line, from rank 0
        d = json.loads(lines[0])
        assert d["launch_check"] is True and d["backend"] == "gloo"
        for k, v in want.items():
            assert d[k] == v, (k, d)
