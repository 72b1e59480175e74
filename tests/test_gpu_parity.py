"""Parity of the HIP engine against the CPU oracle and the reference-made golden planes, through the
C ABI (ss_process / ss_process_device). Needs an MI355X: run with -m gpu."""
import glob
import os

import numpy as np
import pytest

import rtl_sdr_scanner_cpp_amd as pkg
from parity import check_all, check_plane, dont_care_limit

pytestmark = pytest.mark.gpu

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "ref_chain_*.npz")))


def _chunks(total, chunk):
    pos = 0
    while pos < total:
        yield pos, min(total, pos + chunk)
        pos += chunk


def _run(chain, iq, chunk, t_ms=None, hooks=None):
    outs = []
    for a, b in _chunks(iq.shape[0], chunk):
        if hooks and a in hooks:
            hooks[a](chain)
        outs.append(chain.process(iq[a:b], t_ms=None if t_ms is None else t_ms[a:b]))
    res = {k: np.concatenate([o[k] for o in outs]) for k in ("psd", "rel", "avg", "cand_idx", "cand_avg")}
    counts = np.concatenate([np.diff(o["cand_off"]) for o in outs])
    res["cand_off"] = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    return res


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
@pytest.mark.parametrize("chunk", [5, 50])
def test_engine_matches_reference_made_golden(path, chunk):
    """Planes and candidates produced by the reference's own compiled PSD/NoiseLearner/Transmission code
    (tests/golden/make_golden.py), timestamps included."""
    g = np.load(path)
    n, fs, center = int(g["n"]), int(g["fs"]), int(g["center"])
    retune_at = int(g["retune_at"])
    lo, hi = center - fs // 2, center + fs // 2
    eng = pkg.SpectrumEngine(fs, center, fft_size=n, decim=1, max_batch=64, ignored=g["ignored"])
    hooks = None
    if retune_at >= 0:
        assert retune_at % chunk == 0
        hooks = {retune_at: lambda c: (c.set_frequency_range(lo + fs, hi + fs), c.reset())}
    got = _run(eng, g["iq"], chunk, t_ms=g["t_ms"], hooks=hooks)
    ref = {k: g[k] for k in ("psd", "rel", "avg", "cand_off", "cand_idx")}
    errs, ncand, ndc = check_all(got, ref)
    assert ncand > 100 and ndc <= dont_care_limit(ncand), (ncand, ndc, errs)


CASES = [
    # n, fs, decim, fmt, nframes, chunk, learn, seed   (combs switch on 5 frames after learning, off 3 before the end)
    (64, 16_000, 1, "cf32", 120, 17, 20, 1),
    (1024, 256_000, 1, "cf32", 150, 64, 30, 2),
    (2048, 512_000, 3, "cf32", 90, 32, 25, 3),
    (512, 128_000, 1, "cf32", 5000, 5000, 40, 12),  # one long call: more frames than wave slots, per-frame offsets beyond one scan trip
    (4096, 1_024_000, 1, "cs8", 80, 80, 22, 4),
    (8192, 2_048_000, 1, "cf32", 96, 48, 24, 5),
    (8192, 2_048_000, 5, "cu8", 70, 70, 21, 6),
    (16384, 4_096_000, 1, "cf32", 64, 30, 10, 7),
    (32768, 6_000_000, 1, "cu8", 60, 24, 8, 10),    # 256 x 128 four-step (what getFft picks for a 6 MS/s Airspy)
    (65536, 20_000_000, 1, "cs8", 56, 25, 6, 8),
    (131072, 20_000_000, 1, "cf32", 40, 20, 4, 9),  # what getFft(20 MS/s, 250) picks: the reference's HackRF regime
]


@pytest.mark.parametrize("n,fs,decim,fmt,nframes,chunk,learn,seed", CASES)
def test_engine_matches_oracle(oracle_mod, n, fs, decim, fmt, nframes, chunk, learn, seed):
    center = 145_000_000
    band = pkg.synth.SyntheticBand(n, decim=decim, seed=seed, on_frame=learn + 5, off_frame=nframes - 3)
    if fmt == "cf32":
        iq, in_format = band.frames_cf32(nframes), pkg.abi.SS_FMT_CF32
    elif fmt == "cs8":
        iq, in_format = band.frames_cs8(nframes), pkg.abi.SS_FMT_CS8
    else:
        iq, in_format = band.frames_cu8(nframes), pkg.abi.SS_FMT_CU8
    kw = dict(fft_size=n, decim=decim, in_format=in_format, learn_frames=learn, max_batch=max(chunk, 8))
    eng = pkg.SpectrumEngine(fs, center, **kw)
    orc = oracle_mod.oracle_chain(fs, center, **kw)
    got, ref = _run(eng, iq, chunk), _run(orc, iq, chunk)
    errs, ncand, ndc = check_all(got, ref)
    assert ncand > (50 if n >= 256 else 5), "the test vector must produce detections"
    assert ndc <= dont_care_limit(ncand), (ncand, ndc)
    thr_g, ready_g = eng.read_noise()
    thr_o, ready_o = orc.read_noise()
    assert ready_g and ready_o
    check_plane("noise ceiling", thr_g[None], thr_o[None])


ALTERNATIVES = [  # (environment, fft size, sample rate, format): every measurement hook of DESIGN.md 6f meets the same contract
    ({"SS_BACKEND": "unfused"}, 8192, 2_048_000, "cf32"),
    ({"SS_FFT_IMPL": "generic"}, 8192, 2_048_000, "cf32"),
    ({"SS_FFT_TW": "0", "SS_FFT_SWZ": "0"}, 8192, 2_048_000, "cs8"),
    ({"SS_FFT_TW": "1"}, 8192, 2_048_000, "cu8"),
    ({"SS_PIPELINE": "0"}, 8192, 2_048_000, "cf32"),
    ({"SS_CULL": "0"}, 8192, 2_048_000, "cf32"),
    ({"SS_STEP_ORDER": "|F5,D3,E1"}, 8192, 2_048_000, "cf32"),
    ({"SS_STEP_ORDER": "E*|D77,F99", "SS_STEP_PRIO_FFT": "2", "SS_STEP_PRIO_OTHER": "1"}, 8192, 2_048_000, "cf32"),
    ({"SS_FFT_IMPL": "generic"}, 2048, 512_000, "cf32"),
    ({"SS_FFT_IMPL": "generic"}, 65536, 20_000_000, "cs8"),
    ({"SS_FFT_ROWSR": "0", "SS_FFT_SUB": "1"}, 131072, 20_000_000, "cf32"),
    ({"SS_FFT_XCDMAP": "0"}, 262144, 20_000_000, "cf32"),
    ({"SS_EMIT_WIDE": "0"}, 65536, 20_000_000, "cs8"),
    ({"SS_STEP_LONG": "0"}, 65536, 20_000_000, "cs8"),
    ({"SS_STEP_LONG": "0"}, 16384, 4_000_000, "cf32"),
    ({"SS_FFT_TWOPASS": "0"}, 1 << 20, 61_440_000, "cs8"),  # 2^20 points as 256 x 4096 in three passes (round 3's form; the product takes 1024 x 1024 in two)
    ({"SS_C1024_WIDE": "0"}, 1 << 20, 61_440_000, "cs8"),   # ... in two passes with 8-column tiles as k_scan_step's FFT role (the product: 16-column tiles, a launch of their own, the ROW tiles as the role)
    ({"SS_PLAN_FUSED": "0", "SS_WIN_CALC": "0"}, 1 << 20, 61_440_000, "cf32"),  # the plan as a launch of its own, window taps from the table (round 4 before session 14)
    ({"SS_CULL_65536": "0"}, 65536, 20_000_000, "cs8"),     # 65536 points as until session 19 of round 4: every tile evaluated, columns as the FFT role with all passengers
    ({"SS_ROWS256_STEP": "0", "SS_WIN_CALC": "0"}, 65536, 20_000_000, "cs8"),  # ... culled, rows and plan as launches of their own
    ({"SS_DET_LAG2": "0"}, 65536, 20_000_000, "cs8"),       # ... detect(k - 1) on the row launch
    ({"SS_MERGE_65536": "0"}, 65536, 20_000_000, "cs8"),    # ... two launches per call also for detect-mode calls of up to 128 frames
    ({"SS_EMIT_ON_ROWS": "1", "SS_LIST_FIRST": "0"}, 65536, 20_000_000, "cf32"),
    ({"SS_PLAN_FIRST": "32"}, 8192, 2_048_000, "cf32"),     # 8192 points: the first pairs of every list on detect workgroups of their own
    ({"SS_HALO_MAXIMA": "0"}, 8192, 2_048_000, "cf32"),     # 8192 points: the halo frames leave no maxima, the tiles at a batch's start are evaluated untested (until session 36 of round 5)
    # round 6
    ({"SS_ROWS1024X256": "0"}, 262144, 61_440_000, "cf32"),  # 262144 points on round 2's path (k_fft_rows256xR_psd, every tile evaluated)
    ({"SS_MERGE_65536": "0"}, 262144, 61_440_000, "cs8"),   # ... culled, two launches per call (KIND 10 + k_fft_rows1024_psd<8>)
    ({"SS_PLAN_FUSED": "0", "SS_LIST_FIRST": "0"}, 262144, 61_440_000, "cf32"),  # ... the plan (plan_x256_run) as a launch of its own
    ({"SS_DIF8_SINGLE_MAX": "32"}, 65536, 20_000_000, "cs8"),  # the fold with ONE residue per workgroup for short calls (KIND 11: measured, not kept)
]


@pytest.mark.parametrize("env,n,fs,fmt", ALTERNATIVES, ids=lambda v: "-".join(f"{k}={x}" for k, x in v.items()) if isinstance(v, dict) else str(v))
def test_alternative_implementations_meet_the_contract(oracle_mod, monkeypatch, diag_lib, env, n, fs, fmt):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    nframes, learn = 48, 6
    band = pkg.synth.SyntheticBand(n, seed=14, on_frame=learn + 5, off_frame=nframes - 3)
    iq, in_format = (band.frames_cf32(nframes), pkg.abi.SS_FMT_CF32) if fmt == "cf32" else (band.frames_cs8(nframes), pkg.abi.SS_FMT_CS8)
    kw = dict(fft_size=n, decim=1, in_format=in_format, learn_frames=learn, max_batch=20)
    got = _run(pkg.SpectrumEngine(fs, 145_000_000, **kw), iq, 20)
    ref = _run(oracle_mod.oracle_chain(fs, 145_000_000, **kw), iq, 20)
    errs, ncand, ndc = check_all(got, ref)
    assert ncand > 50 and ndc <= dont_care_limit(ncand), (ncand, ndc)


@pytest.mark.parametrize("n,fmt", [(512, "cs8"), (1024, "cf32"), (2048, "cs8"), (2048, "cf32"), (4096, "cu8"), (8192, "cf32"), (16384, "cs8"),
                                   (65536, "cf32"), (131072, "cs8"), (262144, "cf32"), (1 << 20, "cs8")])
def test_psd_of_a_frame_does_not_depend_on_its_position_in_the_batch(n, fmt):
    """Kernels that take several frames per workgroup must round every frame the same way (the two unrolled halves of the
    2048-point kernel once did not: the compiler chose the FMA operand per call site): frame-range sharding and
    'any cut of the stream gives the same bits' all rest on it."""
    band = pkg.synth.SyntheticBand(n, seed=8, on_frame=5, off_frame=400)
    nf = 24 if n <= 16384 else 10
    iq, in_format = {"cf32": (band.frames_cf32, pkg.abi.SS_FMT_CF32), "cs8": (band.frames_cs8, pkg.abi.SS_FMT_CS8),
                     "cu8": (band.frames_cu8, pkg.abi.SS_FMT_CU8)}[fmt]
    iq = iq(nf)
    kw = dict(fft_size=n, decim=1, in_format=in_format, learn_frames=2, max_batch=nf)
    ref = pkg.SpectrumEngine(250 * n, 145_000_000, **kw).process(iq, want=("psd",))["psd"]
    for off in (1, 2, 3, 5):
        got = pkg.SpectrumEngine(250 * n, 145_000_000, **kw).process(iq[off:], want=("psd",))["psd"]
        np.testing.assert_array_equal(got, ref[off:], err_msg=f"batch starting at frame {off}")


def test_gpu_fft_is_as_accurate_as_the_cpu_fp32_ffts(oracle_mod):
    """Against an fp64 evaluation of the same definition (window, FFT, |X|^2/fs), the engine's PSD is as close
    as the oracle's fp32 FFT is (and as MKL's FFTW interface is, tests/test_oracle_fft.py): the parity
    tolerances are set by fp32 itself, not by this kernel."""
    n, fs = 8192, 2_048_000
    band = pkg.synth.SyntheticBand(n, seed=41, on_frame=0, off_frame=100)
    iq = band.frames_cf32(16)
    kw = dict(fft_size=n, decim=1, learn_frames=4, max_batch=16)
    got = pkg.SpectrumEngine(fs, 145_000_000, **kw).process(iq, want=("psd",))["psd"].astype(np.float64)
    ref = oracle_mod.oracle_chain(fs, 145_000_000, **kw).process(iq, want=("psd",))["psd"].astype(np.float64)
    k = np.arange(n)
    w = (0.54 - 0.46 * np.cos(2 * np.pi * k / (n - 1))).astype(np.float32).astype(np.float64)
    exact = np.abs(np.fft.fftshift(np.fft.fft(iq.astype(np.complex128) * w, axis=1), axes=1)) ** 2 / fs
    rms = np.sqrt(exact.mean(axis=1, keepdims=True))

    def amp_err(db):  # radial amplitude error of each bin in units of the frame's spectrum RMS
        return np.abs(np.sqrt(10.0 ** (db / 10.0)) - np.sqrt(exact)) / rms

    e_gpu, e_orc = amp_err(got), amp_err(ref)
    weak = exact < exact.mean(axis=1, keepdims=True)
    # the FFT proper: on weak bins (where the rounding floor decides) the kernel is as good as the CPU fp32 FFT
    assert e_gpu[weak].max() <= 1.5 * e_orc[weak].max()
    assert np.sqrt((e_gpu[weak] ** 2).mean()) < 3e-7
    # the dB conversion (hardware log2 + constant offset, fft_kernels.h psd_db): a few ulp of a ~-50 dB value
    exact_db = 10.0 * np.log10(exact)
    assert np.abs(got - exact_db)[~weak].max() < 2e-5 and np.abs(ref - exact_db)[~weak].max() < 2e-5


@pytest.mark.parametrize("logn,rows", [(20, ""), (20, "1"), (19, ""), (19, "0"), (18, ""), (18, "0"), (17, "0")])
def test_one_million_point_frames(oracle_mod, logn, rows, monkeypatch, diag_lib):
    """BASELINE config 5 frame size (2^20) and the sizes below it, detect chain on a few frames. Rows of the four-step
    transform: one register-pass kernel per row of 512 .. 2048 points (default up to 2^19; forced for 2^20 with
    SS_FFT_ROWSR=1), or radix-A step + 256-point rows (default at 2^20, forced with "0"; generic LDS rows below 2^19)."""
    if rows:
        monkeypatch.setenv("SS_FFT_ROWSR", rows)
    n, fs, center = 1 << logn, 61_440_000, 400_000_000
    band = pkg.synth.SyntheticBand(n, seed=9, on_frame=2, off_frame=100, comb_width=48)
    iq = band.frames_cf32(6)
    kw = dict(fft_size=n, decim=1, learn_frames=2, max_batch=8, grouping_y=3)
    got = pkg.SpectrumEngine(fs, center, **kw).process(iq)
    ref = oracle_mod.oracle_chain(fs, center, **kw).process(iq)
    errs, ncand, ndc = check_all(got, ref)
    assert ncand > 50 and ndc <= 2


def test_ignored_ranges_and_partial_range(oracle_mod):
    n, fs, center = 1024, 256_000, 433_000_000
    band = pkg.synth.SyntheticBand(n, seed=11, on_frame=40, off_frame=100)
    iq = band.frames_cf32(110)
    ign = [center + 40_000, center + 52_000, center - 80_000, center - 60_000]
    kw = dict(fft_size=n, decim=1, learn_frames=15, max_batch=128, ignored=ign, range_lo=center - 100_000, range_hi=center + 90_000)
    got = pkg.SpectrumEngine(fs, center, **kw).process(iq)
    ref = oracle_mod.oracle_chain(fs, center, **kw).process(iq)
    check_all(got, ref)
    full = oracle_mod.oracle_chain(fs, center, **{**kw, "ignored": [], "range_lo": center - fs // 2, "range_hi": center + fs // 2}).process(iq)
    assert len(full["cand_idx"]) > len(ref["cand_idx"]) > 0  # the restrictions removed something, not everything


def test_retune_keeps_noise_per_centre_and_resets_averager(oracle_mod):
    """SdrDevice::setFrequencyRange sequence (sdr_device.cpp:54-80): new centre learns its own ceiling,
    going back finds the old one (NoiseLearner never forgets, noise_learner.h:33), the averager restarts."""
    n, fs, c0, c1 = 512, 128_000, 100_000_000, 100_128_000
    band = pkg.synth.SyntheticBand(n, seed=12, on_frame=30, off_frame=10_000)
    kw = dict(fft_size=n, decim=1, learn_frames=12, max_batch=64)
    eng, orc = pkg.SpectrumEngine(fs, c0, **kw), oracle_mod.oracle_chain(fs, c0, **kw)
    for step, centre in enumerate((c0, c1, c0, c1)):
        iq = band.frames_cf32(60)
        for ch in (eng, orc):
            if step:
                ch.set_frequency_range(centre - fs // 2, centre + fs // 2)
                ch.reset()
        got, ref = eng.process(iq), orc.process(iq)
        check_all(got, ref)
        if step >= 2:  # ceiling already known: no learning frames, only the 20-frame averager warm-up
            assert (ref["rel"][0] != -100).all() and (ref["avg"][19] == -100).all() and (ref["avg"][20] != -100).all()
    eng.reset_noise(), orc.reset_noise()
    iq = band.frames_cf32(40)
    check_all(eng.process(iq), orc.process(iq))


def test_timestamp_learning(oracle_mod):
    n, fs, center = 256, 64_000, 145_000_000
    band = pkg.synth.SyntheticBand(n, seed=13, on_frame=70, off_frame=120, comb_width=12)
    iq = band.frames_cf32(130)
    t = (77 + 37 * np.arange(130)).astype(np.int64)  # 2000 ms have elapsed on frame 55, which still learns
    kw = dict(fft_size=n, decim=1, max_batch=64)
    got = _run(pkg.SpectrumEngine(fs, center, **kw), iq, 50, t_ms=t)
    ref = _run(oracle_mod.oracle_chain(fs, center, **kw), iq, 50, t_ms=t)
    check_all(got, ref)
    assert (ref["rel"][55] == -100).all() and (ref["rel"][56] != -100).all()


def test_edge_cases(oracle_mod):
    n, fs, center = 256, 64_000, 145_000_000
    kw = dict(fft_size=n, decim=1, learn_frames=5, max_batch=32)
    eng = pkg.SpectrumEngine(fs, center, **kw)
    r = eng.process(np.zeros((0, n), np.complex64))  # empty batch
    assert r["cand_off"].tolist() == [0] and r["psd"].shape == (0, n)
    with pytest.raises(pkg.abi.SpecscanError) as e:  # batch larger than max_batch
        eng.process(np.zeros((33, n), np.complex64))
    assert e.value.status == pkg.abi.SS_ERR_BATCH
    z = eng.process(np.zeros((3, n), np.complex64))  # all-zero IQ: log10f(0) = -inf like the reference
    assert np.isneginf(z["psd"]).all()
    # candidate overflow: offsets stay exact, list is truncated, status says so
    band = pkg.synth.SyntheticBand(n, seed=14, on_frame=5, off_frame=60, comb_width=12)
    iq = band.frames_cf32(32)
    e2, o2 = pkg.SpectrumEngine(fs, center, **kw), oracle_mod.oracle_chain(fs, center, **kw)
    got, ref = e2.process(iq, cand_cap=7), o2.process(iq, cand_cap=7)
    assert got["status"] == ref["status"] == pkg.abi.SS_ERR_CAND_OVERFLOW
    np.testing.assert_array_equal(got["cand_off"], ref["cand_off"])
    np.testing.assert_array_equal(got["cand_idx"], ref["cand_idx"])
    for bad in (dict(fft_size=100), dict(fft_size=32), dict(grouping_x=20), dict(decim=0), dict(in_format=7)):
        with pytest.raises(pkg.abi.SpecscanError):
            pkg.SpectrumEngine(fs, center, **{**kw, **bad})


def test_read_window_addresses_batch_and_ring_rows(oracle_mod):
    n, fs, center = 512, 128_000, 145_000_000
    band = pkg.synth.SyntheticBand(n, seed=15, on_frame=20, off_frame=70)
    kw = dict(fft_size=n, decim=1, learn_frames=8, max_batch=64, flags=pkg.abi.SS_FLAG_KEEP_PLANES)
    eng, orc = pkg.SpectrumEngine(fs, center, **kw), oracle_mod.oracle_chain(fs, center, **kw)
    for _ in range(2):
        iq = band.frames_cf32(40)
        got, ref = eng.process(iq), orc.process(iq)
    for plane, key in ((pkg.abi.SS_PLANE_PSD, "psd"), (pkg.abi.SS_PLANE_REL, "rel"), (pkg.abi.SS_PLANE_AVG, "avg")):
        np.testing.assert_array_equal(eng.read_window(plane, 17, 100, 164), got[key][17, 100:164])
    for back in (1, 7, 20):  # ring rows from before the batch (Transmission::getBestIndex)
        a = eng.read_window(pkg.abi.SS_PLANE_REL, -back, 0, n)
        b = orc.read_window(pkg.abi.SS_PLANE_REL, -back, 0, n)
        check_plane("ring row", a[None], b[None])


def test_device_resident_entry_point_matches_host_entry_point():
    import torch
    n, fs, center = 8192, 2_048_000, 145_000_000
    band = pkg.synth.SyntheticBand(n, seed=16, on_frame=30, off_frame=90)
    iq = band.frames_cf32(100)
    kw = dict(fft_size=n, decim=1, learn_frames=10, max_batch=128)
    host = pkg.SpectrumEngine(fs, center, **kw).process(iq)
    eng = pkg.SpectrumEngine(fs, center, **kw)
    dev = torch.device("cuda:0")
    d_iq = torch.from_numpy(iq.view(np.float32)).to(dev)
    planes = [torch.empty((100, n), dtype=torch.float32, device=dev) for _ in range(3)]
    off = torch.zeros(101, dtype=torch.int32, device=dev)
    cap = 100 * n
    idx = torch.empty(cap, dtype=torch.int32, device=dev)
    cav = torch.empty(cap, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()  # (torch's fills run on torch's stream, the library writes from its own)
    torch.cuda.synchronize()
    eng.process_device(d_iq, 100, *planes, off, idx, cav)
    eng.sync()
    for t, k in zip(planes, ("psd", "rel", "avg")):
        np.testing.assert_array_equal(t.cpu().numpy(), host[k])
    np.testing.assert_array_equal(off.cpu().numpy(), host["cand_off"])
    total = int(off[-1])
    np.testing.assert_array_equal(idx[:total].cpu().numpy(), host["cand_idx"])
    np.testing.assert_array_equal(cav[:total].cpu().numpy(), host["cand_avg"])


def test_tracker_on_engine_planes_follows_the_reference(oracle_mod):
    """End of the drop-in: engine planes + candidates -> host-side tracker -> the Scanner's (shift Hz, flush) list.
    The tracker itself is bit-exact against the reference on identical planes (tests/test_signal_tracker.py); fed
    with the engine's planes (equal to the oracle's within the fp32 FFT rounding floor) it must report the same
    transmissions at the same tuned frequencies; a start or stop may move by one frame when a smoothed value sits
    on the threshold."""
    n, fs, center, nframes, dt = 1024, 256_000, 145_000_000, 330, 40
    band = pkg.synth.SyntheticBand(n, seed=34, on_frame=70, off_frame=190, comb_width=32)
    iq = band.frames_cf32(nframes)
    t = (1_000 + dt * np.arange(nframes)).astype(np.int64)
    kw = dict(fft_size=n, decim=1, max_batch=128)
    eng, orc = pkg.SpectrumEngine(fs, center, **kw), oracle_mod.oracle_chain(fs, center, **kw)
    got, ref = _run(eng, iq, 128, t_ms=t), _run(orc, iq, 128, t_ms=t)
    tk = dict(min_time_ms=400, timeout_ms=600)
    a = pkg.tracker.SignalTracker(n, fs, **tk).process_batch(t, got["avg"], got["rel"], got["cand_off"], got["cand_idx"])
    b = pkg.tracker.SignalTracker(n, fs, **tk).process_batch(t, ref["avg"], ref["rel"], ref["cand_off"], ref["cand_idx"])
    same = sum(np.array_equal(a[f][0], b[f][0]) for f in range(nframes))
    assert same >= nframes - 4, f"only {same} of {nframes} frames notify the same list"
    shifts_a = {int(s) for f in range(nframes) for s in a[f][0][:, 0]}
    shifts_b = {int(s) for f in range(nframes) for s in b[f][0][:, 0]}
    assert shifts_a == shifts_b and len(shifts_b) >= 4


def test_two_contexts_are_independent(oracle_mod):
    n, fs = 1024, 256_000
    a = pkg.SpectrumEngine(fs, 140_000_000, fft_size=n, decim=1, learn_frames=10, max_batch=64)
    b = pkg.SpectrumEngine(fs, 142_000_000, fft_size=n, decim=1, learn_frames=10, max_batch=64)
    ba = pkg.synth.SyntheticBand(n, seed=17, on_frame=30, off_frame=60)
    bb = pkg.synth.SyntheticBand(n, seed=18, on_frame=25, off_frame=55)
    xa, xb = ba.frames_cf32(64), bb.frames_cf32(64)
    ra1, rb1 = a.process(xa[:32]), b.process(xb[:32])
    ra2, rb2 = a.process(xa[32:]), b.process(xb[32:])
    oa = oracle_mod.oracle_chain(fs, 140_000_000, fft_size=n, decim=1, learn_frames=10, max_batch=64).process(xa)
    ob = oracle_mod.oracle_chain(fs, 142_000_000, fft_size=n, decim=1, learn_frames=10, max_batch=64).process(xb)
    check_plane("a", np.concatenate([ra1["avg"], ra2["avg"]]), oa["avg"])
    check_plane("b", np.concatenate([rb1["avg"], rb2["avg"]]), ob["avg"])


def test_contexts_on_their_own_threads_with_control_calls_from_a_third():
    """The reference runs one chain per device, each on GNU Radio's scheduler threads, and reads / retunes them from the
    Scanner thread (control entry points take the block's mutex). Four contexts driven from four threads while
    a fifth thread keeps reading their noise ceilings: every context must produce exactly what it produces alone."""
    import threading
    n, fs, nframes, chunk = 2048, 512_000, 160, 16
    seeds = [31, 32, 33, 34]
    bands = [pkg.synth.SyntheticBand(n, seed=s_, on_frame=30, off_frame=120) for s_ in seeds]
    data = [b.frames_cf32(nframes) for b in bands]

    def scan(eng, x, retune_at, out):
        for pos in range(0, nframes, chunk):
            if pos == retune_at:  # what SdrDevice::setFrequencyRange does to its chain (sdr_device.cpp:66-77)
                eng.set_frequency_range(150_000_000 - fs // 2, 150_000_000 + fs // 2)
                eng.reset()
            r = eng.process(x[pos:pos + chunk], want=("avg",))
            out.append((r["avg"].copy(), r["cand_off"].copy(), r["cand_idx"].copy()))

    def make(k):
        return pkg.SpectrumEngine(fs, 140_000_000 + 2_000_000 * k, fft_size=n, decim=1, learn_frames=10, max_batch=chunk)

    alone = []
    for k in range(4):
        out = []
        scan(make(k), data[k], 96 if k % 2 else -1, out)
        alone.append(out)
    engines = [make(k) for k in range(4)]
    together = [[] for _ in range(4)]
    stop = threading.Event()
    reads = [0]

    def poke():
        while not stop.is_set():
            for e in engines:
                thr, _learned = e.read_noise()
                assert thr.shape == (n,)
                reads[0] += 1

    workers = [threading.Thread(target=scan, args=(engines[k], data[k], 96 if k % 2 else -1, together[k])) for k in range(4)]
    poker = threading.Thread(target=poke)
    poker.start()
    for w in workers:
        w.start()
    for w in workers:
        w.join()
    stop.set()
    poker.join()
    assert reads[0] > 0
    for k in range(4):
        assert len(together[k]) == len(alone[k]) == nframes // chunk
        for (a1, o1, i1), (a2, o2, i2) in zip(alone[k], together[k]):
            np.testing.assert_array_equal(a1, a2)
            np.testing.assert_array_equal(o1, o2)
            np.testing.assert_array_equal(i1, i2)
    assert sum(int(o[-1]) for _, o, _ in alone[0]) > 100


@pytest.mark.parametrize("impl", ["detect", "standalone"])
def test_spectrogram_side_branch(oracle_mod, impl, monkeypatch, diag_lib):
    """Spectrogram::process/send (spectrogram.cpp:45-75) on the raw PSD: bin-decimated mean accumulated over frames,
    published as int8 by the C++ float -> int8 conversion. The float means agree to 1e-4 dB; the int8 bytes are
    identical wherever the reference's own mean is not within 1e-3 of an integer (there truncation decides)."""
    import ctypes as C
    monkeypatch.setenv("SS_SPEC_IMPL", impl)  # inside k_detect_fused (default) or the two stand-alone kernels
    n, fs, center = 8192, 2_048_000, 145_000_000
    band = pkg.synth.SyntheticBand(n, seed=51, on_frame=20, off_frame=90)
    iq = band.frames_cf32(150)
    eng = pkg.SpectrumEngine(fs, center, fft_size=n, decim=1, learn_frames=10, max_batch=64, flags=pkg.abi.SS_FLAG_SPECTROGRAM)
    orc = oracle_mod.oracle_chain(fs, center, fft_size=n, decim=1, learn_frames=10, max_batch=64)
    L = oracle_mod.lib()
    g = L.orc_spectrogram_create(n, fs)
    size = L.orc_spectrogram_size(g)
    assert size == 2048 and eng._lib.ss_spectrogram_size(eng._h) == size
    for a, b in ((0, 64), (64, 100), (100, 150)):  # two "send" intervals, ragged batches
        got_planes = eng.process(iq[a:b], want=("psd",))
        ref_planes = orc.process(iq[a:b], want=("psd",))
        for row in ref_planes["psd"]:
            L.orc_spectrogram_process(g, row.ctypes.data_as(C.POINTER(C.c_float)))
        if b in (100, 150):
            want8, wantf = np.zeros(size, np.int8), np.zeros(size, np.float32)
            cnt_ref = L.orc_spectrogram_send(g, want8.ctypes.data_as(C.POINTER(C.c_int8)), wantf.ctypes.data_as(C.POINTER(C.c_float)))
            got8, gotf, cnt = eng.spectrogram_read()
            assert cnt == cnt_ref == (100 if b == 100 else 50)
            assert np.max(np.abs(gotf - wantf)) < 1e-4 * 60
            frac = np.abs(wantf - np.trunc(wantf))
            decided = (frac > 1e-3) & (frac < 1 - 1e-3)
            np.testing.assert_array_equal(got8[decided], want8[decided])
            assert decided.mean() > 0.99
    assert eng.spectrogram_read()[2] == 0  # nothing accumulated since the last send
    L.orc_spectrogram_destroy(g)
