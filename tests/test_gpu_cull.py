"""Tile culling (csrc/detect_fused.h): a 16-frame x 256-bin averaging tile whose per-segment PSD maxima (left by the FFT
role) show that no bin can reach start_level is not evaluated. The decision must be EXACT: the candidate lists with
culling equal those with SS_FLAG_NO_CULL bit for bit, and both equal the reference's own code (oracle/_ref) — on the
host-buffer path, on the deep-pipelined device path, with ignored ranges, weak signals just around the threshold, degenerate
frames, int8 input and calls that do not start on a tile boundary. Needs an MI355X: run with -m gpu."""
import numpy as np
import pytest

import rtl_sdr_scanner_cpp_amd as pkg
from parity import BAND, cand_set, dont_care_limit

pytestmark = pytest.mark.gpu

N, FS, CENTER = 8192, 2_048_000, 145_000_000


def _lists(o):
    return [o["cand_idx"][o["cand_off"][f]:o["cand_off"][f + 1]] for f in range(len(o["cand_off"]) - 1)]


def _same(a, b):
    assert len(a) == len(b)
    for f, (x, y) in enumerate(zip(a, b)):
        assert np.array_equal(x, y), (f, x[:8], y[:8])


def _engine(flags=0, **kw):
    kw.setdefault("max_batch", 1024)
    return pkg.SpectrumEngine(FS, CENTER, fft_size=N, decim=1, flags=flags, **kw)


def _run_host(eng, iq, chunk, t=None, want=("psd",)):
    outs = [eng.process(iq[a:a + chunk], t_ms=None if t is None else t[a:a + chunk], want=want) for a in range(0, len(iq), chunk)]
    lists = []
    for o in outs:
        lists += _lists(o)
    avgs = np.concatenate([o["cand_avg"] for o in outs])
    return lists, avgs


@pytest.mark.parametrize("chunk", [1024, 200, 77])
def test_culled_lists_equal_unculled_lists_and_the_reference(ref_mod, chunk):
    nframes = 1024
    band = pkg.synth.SyntheticBand(N, seed=31, on_frame=150, off_frame=600, period=700)
    iq = band.frames_cf32(nframes)
    t = (10_000 + 20 * np.arange(nframes)).astype(np.int64)
    a, a_avg = _run_host(_engine(), iq, chunk, t)
    b, b_avg = _run_host(_engine(pkg.abi.SS_FLAG_NO_CULL), iq, chunk, t)
    _same(a, b)
    assert np.array_equal(a_avg, b_avg)
    assert sum(len(x) for x in a) > 20_000
    ref_mod.ref().orc_set_fft_backend(0)
    r = ref_mod.RefChain(N, FS, CENTER - FS // 2, CENTER + FS // 2).process(iq, t)
    got = {(f, int(i)) for f, x in enumerate(a) for i in x}
    want = {(f, int(i)) for f, x in enumerate(r["cands"]) for i in x}
    near = np.abs(r["avg"] - np.float32(8.0)) < BAND
    outside = [(f, i) for (f, i) in got ^ want if not near[f, i]]
    assert not outside, sorted(outside)[:10]
    assert len(got ^ want) <= dont_care_limit(len(want))


def test_signals_around_the_threshold_and_degenerate_frames():
    """Combs from 4 dB below to 6 dB above start_level (tiles whose bound sits near the cut), an all-zero frame (-inf
    rows), a NaN frame and a huge frame: culled == unculled, list by list."""
    nframes = 640
    rng = np.random.default_rng(5)
    lists = {}
    for rel_db in (16.0, 18.5, 20.0, 22.0, 26.0):
        band = pkg.synth.SyntheticBand(N, seed=int(rel_db * 10), rel_db=rel_db, on_frame=130, off_frame=520, centres=(0.05, -0.11, 0.23, -0.31, 0.37, -0.45, 0.49))
        iq = band.frames_cf32(nframes)
        iq[300] = 0
        iq[340, :100] = np.nan
        iq[380] *= 1e15
        iq[420, rng.integers(0, N, 50)] = np.inf
        outs = {}
        for name, flags in (("cull", 0), ("nocull", pkg.abi.SS_FLAG_NO_CULL)):
            outs[name] = _run_host(_engine(flags, max_batch=320), iq, 320, want=())
        _same(outs["cull"][0], outs["nocull"][0])
        assert np.array_equal(outs["cull"][1], outs["nocull"][1], equal_nan=True)
        lists[rel_db] = sum(len(x) for x in outs["cull"][0])
    print("\ncandidates by comb level:", lists)
    assert lists[26.0] > 10_000


def test_ignored_ranges_int8_and_unaligned_calls():
    nframes = 900
    band = pkg.synth.SyntheticBand(N, seed=9, on_frame=120, off_frame=800)
    iq8 = band.frames_cs8(nframes)
    ignored = [CENTER + 300_000, CENTER + 420_000, CENTER - 600_000, CENTER - 500_000]
    cuts = [0, 130, 131, 300, 563, 900]  # calls of 130, 1, 169, 263, 337 frames: most do not start on a 16-frame boundary
    res = {}
    for name, flags in (("cull", 0), ("nocull", pkg.abi.SS_FLAG_NO_CULL)):
        eng = _engine(flags, in_format=pkg.abi.SS_FMT_CS8, ignored=ignored, max_batch=400)
        lists = []
        for a, b in zip(cuts[:-1], cuts[1:]):
            lists += _lists(eng.process(iq8[a:b], want=()))
        res[name] = lists
    _same(res["cull"], res["nocull"])
    assert sum(len(x) for x in res["cull"]) > 20_000


def test_deep_pipelined_device_calls_culled_equal_unculled():
    import torch
    nb, ncalls = 256, 9
    band = pkg.synth.SyntheticBand(N, seed=4, on_frame=130, off_frame=1500, period=1900)
    iq = band.frames_cf32(nb * ncalls)
    dev = torch.device("cuda", 0)
    res = {}
    for name, flags in (("cull", 0), ("nocull", pkg.abi.SS_FLAG_NO_CULL)):
        eng = _engine(flags, max_batch=nb)
        d_iq = [torch.from_numpy(iq[k * nb:(k + 1) * nb].view(np.float32)).to(dev) for k in range(ncalls)]
        outs = [dict(psd=torch.empty((nb, N), dtype=torch.float32, device=dev), off=torch.zeros(nb + 1, dtype=torch.int32, device=dev),
                     idx=torch.empty(nb * 512, dtype=torch.int32, device=dev), avg=torch.empty(nb * 512, dtype=torch.float32, device=dev)) for _ in range(ncalls)]
        for k in range(ncalls):  # no sync in between: five calls in flight on the library's two queues
            eng.process_device(d_iq[k], nb, psd=outs[k]["psd"], cand_off=outs[k]["off"], cand_idx=outs[k]["idx"], cand_avg=outs[k]["avg"])
        eng.sync()
        lists = []
        for o in outs:
            off, idx = o["off"].cpu().numpy(), o["idx"].cpu().numpy()
            lists += [idx[off[f]:off[f + 1]].copy() for f in range(nb)]
        res[name] = lists
    _same(res["cull"], res["nocull"])
    assert sum(len(x) for x in res["cull"]) > 50_000


def test_short_calls_between_long_ones():
    """Calls of 1 .. 15 frames leave state behind for the next user of their buffers (list headers, mask words): the long
    calls that follow must not see it. (Found by tests/test_gpu_fuzz.py: a 2-frame call's emit stage cleared only two of the
    sixteen copies of its list counts.)"""
    sizes = [300, 2, 7, 300, 15, 40, 1, 500, 3, 3, 260, 16, 90]
    nframes = sum(sizes)
    band = pkg.synth.SyntheticBand(N, seed=12, on_frame=110, off_frame=1400)
    iq = band.frames_cf32(nframes)
    res = {}
    for name, flags in (("cull", 0), ("nocull", pkg.abi.SS_FLAG_NO_CULL)):
        eng = _engine(flags, max_batch=512)
        lists, pos = [], 0
        for sz in sizes:
            lists += _lists(eng.process(iq[pos:pos + sz], want=()))
            pos += sz
        res[name] = lists
    _same(res["cull"], res["nocull"])
    assert sum(len(x) for x in res["cull"]) > 50_000


# ---- long transforms (N = 256 x N2: csrc/detect_fused.h, k_plan_long): the rows kernel leaves the maximum of every 32-bin run
# per frame and writes the averager ring itself; tiles whose 36 rows cannot reach start_level are not evaluated — rows from
# BEFORE the batch included, so short calls cull too. Same bar: culled == unculled, list by list, key by key.
# The product library culls at 65536 points too since session 20 of round 4 (the plan, detect and emit stages of a call ride on the
# column launches of the next three calls); the 65536-point cases below run on the diagnostics build with the switch
# set explicitly, so that they test the culling path whatever the default is.
@pytest.fixture
def cull_65536(monkeypatch, diag_lib):
    monkeypatch.setenv("SS_CULL_65536", "1")
def _long_engine(n, fs, flags=0, **kw):
    return pkg.SpectrumEngine(fs, CENTER, fft_size=n, decim=1, flags=flags, **kw)


def _long_session(n, fs, iq, cuts, flags, in_format, retune_at=(), reset_at=(), learn_frames=24, t=None):
    eng = _long_engine(n, fs, flags, in_format=in_format, max_batch=max(b - a for a, b in zip(cuts[:-1], cuts[1:])), learn_frames=learn_frames)
    lists, avgs = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        if a in retune_at:  # away and back: another centre frequency has another noise ceiling (learnt from scratch), the averager keeps its rows
            eng.set_frequency_range(CENTER + fs - fs // 2, CENTER + fs + fs // 2)
            o = eng.process(iq[a:a + 30], want=())
            lists += _lists(o)
            avgs.append(o["cand_avg"])
            eng.set_frequency_range(CENTER - fs // 2, CENTER + fs // 2)
        if a in reset_at:
            eng.reset()
        o = eng.process(iq[a:b], t_ms=None if t is None else t[a:b], want=())
        lists += _lists(o)
        avgs.append(o["cand_avg"])
    return lists, np.concatenate(avgs)


def test_long_rows_65536_culled_equal_unculled_and_the_reference(ref_mod, cull_65536):
    n, fs, nframes = 65536, 20_000_000, 300
    band = pkg.synth.SyntheticBand(n, seed=41, on_frame=70, off_frame=230)
    iq8 = band.frames_cs8(nframes)
    cuts = [0, 40, 168, 184, 191, 192, 300]  # the last learning frame opens the second call; calls of 128, 16, 7, 1 and 108 frames
    t = (10_000 + 50 * np.arange(nframes)).astype(np.int64)
    a = _long_session(n, fs, iq8, cuts, 0, pkg.abi.SS_FMT_CS8, t=t)
    b = _long_session(n, fs, iq8, cuts, pkg.abi.SS_FLAG_NO_CULL, pkg.abi.SS_FMT_CS8, t=t)
    _same(a[0], b[0])
    assert np.array_equal(a[1], b[1])
    assert sum(len(x) for x in a[0]) > 10_000
    iq = (iq8[..., 0].astype(np.float32) / np.float32(128.0) + 1j * (iq8[..., 1].astype(np.float32) / np.float32(128.0))).astype(np.complex64)
    ref_mod.ref().orc_set_fft_backend(0)
    r = ref_mod.RefChain(n, fs, CENTER - fs // 2, CENTER + fs // 2).process(iq, t)
    got = {(f, int(i)) for f, x in enumerate(a[0]) for i in x}
    want = {(f, int(i)) for f, x in enumerate(r["cands"]) for i in x}
    near = np.abs(r["avg"] - np.float32(8.0)) < BAND
    outside = [(f, i) for (f, i) in got ^ want if not near[f, i]]
    assert not outside, sorted(outside)[:10]
    assert len(got ^ want) <= dont_care_limit(len(want))


def test_long_rows_retune_reset_and_degenerate_frames(cull_65536):
    n, fs, nframes = 65536, 20_000_000, 260
    band = pkg.synth.SyntheticBand(n, seed=43, on_frame=50, off_frame=240, rel_db=19.0, centres=(0.05, -0.11, 0.23, -0.31, 0.37, -0.45, 0.49))
    iq = band.frames_cf32(nframes)
    iq[100] = 0           # -inf rows
    iq[140, :100] = np.nan
    iq[150] *= 1e15
    cuts = [0, 64, 96, 128, 160, 192, 224, 260]
    res = {}
    for name, flags in (("cull", 0), ("nocull", pkg.abi.SS_FLAG_NO_CULL)):
        res[name] = _long_session(n, fs, iq, cuts, flags, pkg.abi.SS_FMT_CF32, retune_at=(160,), reset_at=(224,))
    _same(res["cull"][0], res["nocull"][0])
    assert np.array_equal(res["cull"][1], res["nocull"][1], equal_nan=True)
    assert sum(len(x) for x in res["cull"][0]) > 3_000


def test_long_rows_one_million_points_short_calls():
    n, fs, nframes = 1 << 20, 61_440_000, 80
    band = pkg.synth.SyntheticBand(n, seed=47, on_frame=30, off_frame=70)
    iq8 = band.frames_cs8(nframes)
    cuts = [0, 16, 32, 48, 59, 64, 80]  # 16-frame calls (every row a tile reads lies before the batch or in it), one of 11, one of 5
    res = {}
    for name, flags in (("cull", 0), ("nocull", pkg.abi.SS_FLAG_NO_CULL)):
        res[name] = _long_session(n, fs, iq8, cuts, flags, pkg.abi.SS_FMT_CS8, learn_frames=6)
    _same(res["cull"][0], res["nocull"][0])
    assert np.array_equal(res["cull"][1], res["nocull"][1])
    assert sum(len(x) for x in res["cull"][0]) > 1_000


@pytest.mark.parametrize("nb,ncalls", [(64, 6), (128, 11), (48, 14)])
def test_long_rows_device_calls_without_sync_culled_equal_unculled(cull_65536, nb, ncalls):
    """(128 x 11, 48 x 14: the averager ring of a 65536-point context holds three batches and the averager's reach, so these calls
    send it back to the front of its buffer several times while two detect stages still wait — csrc/ring_place.h, det_lag2.)"""
    _device_calls_culled_equal_unculled(65536, 20_000_000, nb, ncalls)


@pytest.mark.parametrize("nb,ncalls", [(32, 7), (16, 12), (80, 4), (11, 9)])
def test_long_rows_262144_device_calls_without_sync_culled_equal_unculled(nb, ncalls):
    """262144 points — the size getFft picks at 61.44 MS/s — through round 6's path: 256-point column tiles, the 1024-point row tile with
    run maxima (a value per 8 bins, plan layout 3) and ring rows, the 65536-point two-launch pipeline (plan of call k - 1, detect(k - 2)
    and emit(k - 3) on the column launch of call k). 80-frame calls go through in chunks of 64; 11-frame calls do not start on a tile
    boundary and slide the ring's window along its buffer."""
    _device_calls_culled_equal_unculled(1 << 18, 61_440_000, nb, ncalls, min_candidates=2_000)


def test_long_rows_262144_host_calls_retune_reset_and_degenerate_frames():
    n, fs, nframes = 1 << 18, 61_440_000, 150
    band = pkg.synth.SyntheticBand(n, seed=53, on_frame=40, off_frame=140, rel_db=19.0, centres=(0.05, -0.11, 0.23, -0.31, 0.37, -0.45, 0.49))
    iq = band.frames_cf32(nframes)
    iq[60] = 0           # -inf rows
    iq[75, :100] = np.nan
    iq[85] *= 1e15
    cuts = [0, 30, 62, 63, 95, 110, 128, 150]
    res = {}
    for name, flags in (("cull", 0), ("nocull", pkg.abi.SS_FLAG_NO_CULL)):
        res[name] = _long_session(n, fs, iq, cuts, flags, pkg.abi.SS_FMT_CF32, retune_at=(95,), reset_at=(128,), learn_frames=20)
    _same(res["cull"][0], res["nocull"][0])
    assert np.array_equal(res["cull"][1], res["nocull"][1], equal_nan=True)
    assert sum(len(x) for x in res["cull"][0]) > 1_000


# the forms the 65536- and 2^20-point chains went through in round 4 (switches of the diagnostics build; SS_DIF8=0:
# int8 input through the four-step chain, as CF32 input still goes): each of them in detect mode with calls in flight, culled == unculled
@pytest.mark.parametrize("env,n,fs,nb,ncalls", [
    ({"SS_DIF8": "0", "SS_MERGE_65536": "0"}, 65536, 20_000_000, 128, 7),   # two launches per call (session 20's form; still what calls of more than 128 frames and calls that keep a plane take)
    ({"SS_DIF8": "0", "SS_MERGE_65536": "0"}, 65536, 20_000_000, 48, 9),
    ({"SS_DIF8": "0", "SS_DET_LAG2": "0"}, 65536, 20_000_000, 128, 7),
    ({"SS_DIF8": "0", "SS_ROWS256_STEP": "0"}, 65536, 20_000_000, 128, 7),
    ({"SS_DIF8": "0", "SS_LIST_FIRST": "0", "SS_EMIT_ON_ROWS": "1"}, 65536, 20_000_000, 48, 9),
    ({"SS_DIF8": "0", "SS_PLAN_FUSED": "0", "SS_WIN_CALC": "0"}, 65536, 20_000_000, 64, 6),
    ({"SS_DIF8": "0"}, 65536, 20_000_000, 128, 7),   # round 4's one-launch form (KIND 7) on int8 input: what int8 calls took before the fold
    ({"SS_PLAN_FUSED": "0", "SS_WIN_CALC": "0", "SS_LIST_FIRST": "0"}, 1 << 20, 61_440_000, 16, 7),
    # the radix-8 fold's launches (round 5): the listed pairs on detect workgroups of their own where the shipped form puts them behind the
    # fold's workgroups (more workgroups than CUs) and the other way round, the dispatch order of sessions 9-29
    ({"SS_LIST_FIRST_FOLD": "65"}, 65536, 20_000_000, 128, 7),
    ({"SS_LIST_FIRST_FOLD": "1", "SS_STEP_ORDER": "F*,E*,D*"}, 65536, 20_000_000, 48, 9),
    ({"SS_STEP_ORDER": "D64,F*,E*,P*,D*"}, 65536, 20_000_000, 128, 7),
    # 262144 points (round 6): the plan as a launch of its own, every listed pair behind the column tiles
    ({"SS_PLAN_FUSED": "0", "SS_LIST_FIRST": "0"}, 1 << 18, 61_440_000, 32, 6),
], ids=lambda v: "-".join(f"{k}={x}" for k, x in v.items()) if isinstance(v, dict) else str(v))
def test_long_rows_intermediate_forms_culled_equal_unculled(cull_65536, monkeypatch, env, n, fs, nb, ncalls):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    _device_calls_culled_equal_unculled(n, fs, nb, ncalls, min_candidates=500)


def _device_calls_culled_equal_unculled(n, fs, nb, ncalls, min_candidates=5_000):
    import torch
    total = nb * ncalls
    band = pkg.synth.SyntheticBand(n, seed=49, on_frame=min(70, total // 3), off_frame=total - min(90, total // 5))
    iq8 = band.frames_cs8(total)
    dev = torch.device("cuda", 0)
    res = {}
    for name, flags in (("cull", 0), ("nocull", pkg.abi.SS_FLAG_NO_CULL)):
        eng = _long_engine(n, fs, flags, in_format=pkg.abi.SS_FMT_CS8, max_batch=nb, learn_frames=24)
        d_iq = [torch.from_numpy(iq8[k * nb:(k + 1) * nb]).to(dev) for k in range(ncalls)]
        outs = [dict(off=torch.zeros(nb + 1, dtype=torch.int32, device=dev), idx=torch.empty(nb * 4096, dtype=torch.int32, device=dev),
                     avg=torch.empty(nb * 4096, dtype=torch.float32, device=dev)) for _ in range(ncalls)]
        for k in range(ncalls):  # stages deferred from call to call, nothing waited for in between; offsets only for the last call but one
            eng.process_device(d_iq[k], nb, cand_off=outs[k]["off"], cand_idx=None if k == ncalls - 2 else outs[k]["idx"],
                               cand_avg=None if k == ncalls - 2 else outs[k]["avg"])
        eng.sync()
        lists = []
        for k, o in enumerate(outs):
            off, idx = o["off"].cpu().numpy(), o["idx"].cpu().numpy()
            lists += [off.copy()] if k == ncalls - 2 else [idx[off[f]:off[f + 1]].copy() for f in range(nb)]
        res[name] = lists
    _same(res["cull"], res["nocull"])
    assert sum(len(x) for x in res["cull"]) > min_candidates


# ---- random detect-mode sessions: what test_gpu_fuzz.py does not reach (it asks for every plane, and a stage that hands out rel /
# avg planes is never culled). Random size, format, call sizes from one frame up, learning inside or across calls, retunes with and
# without a reset, ignored ranges, zero-frame calls, host and device entry points mixed — culled == unculled, list by list.
def _cull_scenario(seed):
    rng = np.random.default_rng(7000 + seed)
    n = [8192, 65536, 8192, 65536, 1 << 20][seed % 5] if seed % 10 != 9 else 16384  # (16384: a long transform without culling support)
    if seed % 10 == 7:
        n = 131072  # (round 5: the size getFft picks at 20 MS/s — int8 sessions go through the radix-16 fold, CF32 sessions through round 2's path)
    if seed % 10 == 3:
        n = 1 << 18  # (round 6: the size getFft picks at 61.44 MS/s — 256-point columns, the 1024-point row tile, plan layout 3)
    fs = {8192: 2_048_000, 16384: 4_096_000, 65536: 20_000_000, 131072: 20_000_000, 1 << 18: 61_440_000, 1 << 20: 61_440_000}[n]
    nframes = {8192: int(rng.integers(300, 700)), 16384: 200, 65536: int(rng.integers(120, 260)), 131072: int(rng.integers(100, 180)), 1 << 18: int(rng.integers(90, 150)), 1 << 20: 72}[n]
    max_batch = {8192: int(rng.choice([64, 200, 512])), 16384: 64, 65536: int(rng.choice([16, 48, 128])), 131072: int(rng.choice([16, 40, 64])), 1 << 18: int(rng.choice([16, 32, 72])), 1 << 20: 16}[n]
    fmt = str(rng.choice(["cf32", "cs8"])) if n < (1 << 20) else "cs8"
    if n == 131072 and seed % 20 == 7:
        fmt = "cs8"
    learn = int(rng.integers(5, 50)) if n < (1 << 20) else 6
    return rng, n, fs, nframes, max_batch, fmt, learn


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("SS_FUZZ_CULL_SEEDS", "10"))))
def test_random_detect_mode_sessions_culled_equal_unculled(seed, cull_65536):
    _random_session(seed)


@pytest.mark.parametrize("seed", range(20))
def test_random_sessions_across_the_wrap_of_the_frame_counter(seed, cull_65536, monkeypatch):
    """The long transforms index their run maxima with the 30-bit form of the frame counter (RowsExtra::abs0), and a replay at the
    8192-point rate gets to 2^30 frames in twenty seconds: the same random sessions with the counter started so that it wraps in the
    middle of each (SS_ABS_START, diagnostics build; ss_reset starts there again). The engine that evaluates every tile does not
    look at the maxima: culled == unculled across the wrap is the culling path's wrap-safety."""
    *_, nframes, _, _, _ = _cull_scenario(seed)
    monkeypatch.setenv("SS_ABS_START", str((1 << 30) - max(40, nframes // 2) - seed))
    _random_session(seed)


def _random_session(seed):
    import torch
    rng, n, fs, nframes, max_batch, fmt, learn = _cull_scenario(seed)
    on_frame, off_frame = learn + int(rng.integers(3, 40)), nframes - int(rng.integers(4, 20))
    band = pkg.synth.SyntheticBand(n, seed=300 + seed, on_frame=on_frame, off_frame=off_frame, rel_db=float(rng.choice([18.0, 20.0, 25.0])))
    iq = band.frames_cf32(nframes) if fmt == "cf32" else band.frames_cs8(nframes)
    in_format = pkg.abi.SS_FMT_CF32 if fmt == "cf32" else pkg.abi.SS_FMT_CS8
    ign = []
    if rng.random() < 0.5:
        lo = CENTER + int(rng.integers(-fs // 3, fs // 4))
        ign = [lo, lo + fs // 20]
    # the session's script, drawn once and played to both engines
    script, pos = [], 0
    while pos < nframes:
        size = int(min(nframes - pos, rng.integers(1, max_batch + 1)))
        ev = rng.random()
        script.append(("retune_reset" if ev < 0.04 else "retune" if ev < 0.08 else "reset" if ev < 0.11 else "zero" if ev < 0.14 else "", pos, size,
                       bool(rng.random() < 0.5)))
        pos += size
    dev = torch.device("cuda", 0)
    res = {}
    for name, flags in (("cull", 0), ("nocull", pkg.abi.SS_FLAG_NO_CULL)):
        eng = pkg.SpectrumEngine(fs, CENTER, fft_size=n, decim=1, in_format=in_format, learn_frames=learn, max_batch=max_batch, ignored=ign, flags=flags)
        lists, pending, away = [], [], False
        for what, a, size, on_device in script:
            if what in ("retune", "retune_reset"):
                away = not away
                off = fs if away else 0
                eng.set_frequency_range(CENTER + off - fs // 2, CENTER + off + fs // 2)
            if what in ("reset", "retune_reset"):
                eng.reset()
            if what == "zero":
                eng.process(iq[:0], want=())
            if on_device:  # asynchronous entry point: stages deferred from call to call, results fetched at the end
                d = torch.from_numpy(iq[a:a + size].view(np.float32) if fmt == "cf32" else iq[a:a + size]).to(dev)
                o = dict(off=torch.zeros(size + 1, dtype=torch.int32, device=dev), idx=torch.empty(size * 2048, dtype=torch.int32, device=dev), iq=d)
                eng.process_device(d, size, cand_off=o["off"], cand_idx=o["idx"])
                pending.append((len(lists), size, o))
                lists += [None] * size
            else:
                lists += _lists(eng.process(iq[a:a + size], want=()))
        eng.sync()
        for at, size, o in pending:
            off, idx = o["off"].cpu().numpy(), o["idx"].cpu().numpy()
            lists[at:at + size] = [idx[off[f]:off[f + 1]].copy() for f in range(size)]
        res[name] = lists
    _same(res["cull"], res["nocull"])
    if off_frame - on_frame >= 40 and not any(w for w, *_ in script):  # a transmission well inside an undisturbed session: there is something to find
        assert sum(len(x) for x in res["cull"]) > 0
