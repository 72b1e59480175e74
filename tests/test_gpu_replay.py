"""Replay front end on the GPU: a raw dump written with the reference's naming goes through the
pipelined feed (ss_feed_*: pinned slots, H2D overlapped with the chain) and must give, bit for bit, what the
synchronous boundary (ss_process) gives on the same samples, and match the oracle. Run with -m gpu."""
import numpy as np
import pytest

import rtl_sdr_scanner_cpp_amd as pkg
from rtl_sdr_scanner_cpp_amd import replay
from parity import check_candidates, check_plane, floor_tolerance, dont_care_limit

pytestmark = pytest.mark.gpu

FS, CENTER = 2_048_000, 145_000_000


def _write_dump(tmp_path, iq_items, kind):
    """iq_items: [items, N*D] complex64 or [items, N*D, 2] int8 — the continuous stream the SDR source produced."""
    import time
    t = time.struct_time((2025, 3, 7, 9, 5, 1, 0, 0, -1))
    if kind == replay.KIND_CF32:
        name = replay.make_raw_file_name("full", "fc", CENTER, FS, t)
    else:
        name = replay.make_raw_file_name("full", "cs8", CENTER, FS, t)
    path = tmp_path / name[2:]
    sink = replay.RawFileSink(8 if kind == replay.KIND_CF32 else 2)  # FileSink<gr_complex>(1)
    sink.start_recording(str(path))
    sink.work(iq_items)
    sink.close()
    return str(path)


def _concat(results):
    counts = np.concatenate([np.diff(r["cand_off"]) for r in results])
    out = {"cand_off": np.concatenate([[0], np.cumsum(counts)]).astype(np.int32),
           "cand_idx": np.concatenate([r["cand_idx"] for r in results]),
           "cand_avg": np.concatenate([r["cand_avg"] for r in results])}
    if "psd" in results[0]:
        out["psd"] = np.concatenate([r["psd"] for r in results])
    return out


@pytest.mark.parametrize("n,decim,fmt,nframes,batch", [(1024, 1, "cf32", 300, 64), (2048, 3, "cs8", 150, 40), (8192, 1, "cf32", 200, 128)])
def test_replay_equals_synchronous_boundary_and_oracle(tmp_path, oracle_mod, n, decim, fmt, nframes, batch):
    learn = 30
    band = pkg.synth.SyntheticBand(n, decim=decim, seed=11, on_frame=learn + 5, off_frame=nframes - 3)
    if fmt == "cf32":
        iq, kind = band.frames_cf32(nframes), replay.KIND_CF32
    else:
        iq, kind = band.frames_cs8(nframes), replay.KIND_CS8
    path = _write_dump(tmp_path, iq, kind)
    info = replay.parse_raw_file_name(path)
    assert (info.frequency, info.sample_rate, info.kind) == (CENTER, FS, kind)
    kw = dict(fft_size=n, decim=decim, learn_frames=learn, max_batch=batch, **replay.engine_overrides_for(info))

    stats = replay.ReplayStats()
    eng = pkg.SpectrumEngine(info.sample_rate, info.frequency, **kw)
    got = _concat(list(replay.replay_file(eng, path, batch=batch, depth=3, want_psd=True, stats=stats)))
    assert stats.frames == nframes and stats.batches == -(-nframes // batch) and stats.candidates == got["cand_off"][-1]

    sync = pkg.SpectrumEngine(info.sample_rate, info.frequency, **kw)
    outs = [sync.process(iq[a:a + batch]) for a in range(0, nframes, batch)]
    want = _concat([{k: o[k] for k in ("cand_off", "cand_idx", "cand_avg", "psd")} for o in outs])
    for k in ("psd", "cand_off", "cand_idx", "cand_avg"):
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)
    assert got["cand_off"][-1] > 100

    ora = oracle_mod.oracle_chain(info.sample_rate, info.frequency, **kw)
    routs = [ora.process(iq[a:a + batch]) for a in range(0, nframes, batch)]
    ref = _concat([{k: o[k] for k in ("cand_off", "cand_idx", "cand_avg", "psd")} for o in routs])
    ref["avg"] = np.concatenate([o["avg"] for o in routs])
    # psd plane against the oracle's; candidate lists against the oracle's
    check_plane("psd", got["psd"], ref["psd"], floor=floor_tolerance(ref["psd"]))
    ncand, ndc = check_candidates(got["cand_off"], got["cand_idx"], ref["cand_off"], ref["cand_idx"], ref["avg"], 8.0)
    assert ncand > 100 and ndc <= dont_care_limit(ncand)


def test_replay_with_timestamps_learns_on_the_clock(tmp_path):
    """frame_period_ms drives the NoiseLearner's 2000 ms window (noise_learner.cpp:23) exactly like t_ms on ss_process."""
    n, nframes, batch = 1024, 220, 50
    band = pkg.synth.SyntheticBand(n, seed=12, on_frame=130, off_frame=215)
    iq = band.frames_cf32(nframes)
    path = _write_dump(tmp_path, iq, replay.KIND_CF32)
    kw = dict(fft_size=n, decim=1, max_batch=batch)
    eng = pkg.SpectrumEngine(FS, CENTER, **kw)
    got = _concat(list(replay.replay_file(eng, path, batch=batch, frame_period_ms=20.0)))
    sync = pkg.SpectrumEngine(FS, CENTER, **kw)
    t_ms = np.round(np.arange(nframes) * 20.0).astype(np.int64)
    outs = [sync.process(iq[a:a + batch], t_ms=t_ms[a:a + batch]) for a in range(0, nframes, batch)]
    want = _concat([{k: o[k] for k in ("cand_off", "cand_idx", "cand_avg")} for o in outs])
    for k in ("cand_off", "cand_idx", "cand_avg"):
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)
    assert got["cand_off"][-1] > 100
    assert (np.diff(got["cand_off"])[:101] == 0).all()  # 2000 ms at 20 ms per frame: frames 0..100 learn


def test_feed_slot_discipline():
    eng = pkg.SpectrumEngine(FS, CENTER, fft_size=1024, decim=1, max_batch=8, learn_frames=2)
    feed = eng.feed(depth=2, cand_cap=1024)
    with pytest.raises(pkg.abi.SpecscanError):
        feed.collect()  # nothing pending
    with pytest.raises(pkg.abi.SpecscanError):
        feed.submit(4)  # nothing acquired
    rng = np.random.default_rng(0)
    for k in range(2):
        buf = feed.acquire()
        assert buf.shape == (8, 1024) and buf.dtype == np.complex64
        buf[:] = (rng.standard_normal((8, 1024)) + 1j * rng.standard_normal((8, 1024))).astype(np.complex64) * 0.05
        feed.submit(8, tag=100 + k)
    assert feed.pending == 2
    with pytest.raises(pkg.abi.SpecscanError):
        feed.acquire()  # both slots in flight: collect first
    r0 = feed.collect()
    assert (r0["nframes"], r0["tag"], r0["status"]) == (8, 100, 0)
    feed.acquire()  # the collected slot is free again
    with pytest.raises(pkg.abi.SpecscanError):
        feed.submit(9)  # > max_batch
    feed.submit(3, tag=7)
    assert feed.collect()["tag"] == 101 and feed.collect()["nframes"] == 3 and feed.pending == 0
    feed.close()


def test_cpp_replay_host_matches_python_path(tmp_path):
    """host/specscan_replay — a C++ program on the C ABI alone — replays the same dump: same candidate total, and its
    `_power.raw` (the reference's DEBUG_SAVE_FULL_POWER dump, sdr_device.cpp:173-176) equals the PSD plane bit for bit."""
    import json
    import os
    import subprocess
    tool = pkg.build.build_replay_tool()
    n, nframes, batch, learn = 2048, 180, 64, 25
    band = pkg.synth.SyntheticBand(n, seed=13, on_frame=learn + 5, off_frame=nframes - 3)
    iq = band.frames_cf32(nframes)
    path = _write_dump(tmp_path, iq, replay.KIND_CF32)
    out_dir = tmp_path / "power"
    out_dir.mkdir()
    r = subprocess.run([tool, path, "--fft", str(n), "--decim", "1", "--batch", str(batch), "--learn-frames", str(learn),
                        "--power-dir", str(out_dir)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    eng = pkg.SpectrumEngine(FS, CENTER, fft_size=n, decim=1, learn_frames=learn, max_batch=batch)
    want = _concat(list(replay.replay_file(eng, path, batch=batch, want_psd=True)))
    assert (rep["frames"], rep["batches"], rep["candidates"]) == (nframes, -(-nframes // batch), int(want["cand_off"][-1]))
    files = os.listdir(out_dir)
    assert files == ["full_20250307_090501_%d_%d_power.raw" % (CENTER, FS)]
    power = np.fromfile(out_dir / files[0], np.float32).reshape(-1, n)
    np.testing.assert_array_equal(power, want["psd"])
    # usage errors and bad names do not reach the GPU
    assert subprocess.run([tool], capture_output=True).returncode == 2
    assert subprocess.run([tool, str(tmp_path / "nonsense.bin")], capture_output=True).returncode == 2
