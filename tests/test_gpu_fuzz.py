"""Seeded random scenarios, engine vs oracle through the same boundary: random size, decimation, input format,
call sizes (1 frame to the whole run), retunes with Averager reset, and ignored ranges — the state machine of the
chain (learning, ring, warm-up, per-centre noise ceilings) exercised in combinations the hand-written cases do not hit.
Run with -m gpu."""
import numpy as np
import pytest

import rtl_sdr_scanner_cpp_amd as pkg
from parity import check_all

pytestmark = pytest.mark.gpu


def _scenario(seed):
    rng = np.random.default_rng(1000 + seed)
    sizes = [64, 128, 256, 512, 1024, 2048, 4096]
    if seed % 4 == 3:  # every fourth draw: the register-kernel sizes (8192-point kernel, 256-column four-step)
        sizes = [8192, 16384, 32768, 65536]
    n = int(rng.choice(sizes))
    decim = int(rng.choice([1, 1, 2, 5]))
    fmt = str(rng.choice(["cf32", "cf32", "cs8", "cu8"]))
    fs = int(n * rng.choice([200, 250, 125]))
    nframes = int(rng.integers(90, 220)) if n <= 8192 else int(rng.integers(60, 110))
    learn = int(rng.integers(5, 40))
    max_batch = int(rng.choice([8, 16, 64, 256]))
    return rng, n, decim, fmt, fs, nframes, learn, max_batch


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("SS_FUZZ_SEEDS", "16"))))
def test_random_scenario(oracle_mod, seed):
    rng, n, decim, fmt, fs, nframes, learn, max_batch = _scenario(seed)
    center = 145_000_000
    band = pkg.synth.SyntheticBand(n, decim=decim, seed=50 + seed, on_frame=learn + 3, off_frame=nframes - 4)
    iq = {"cf32": band.frames_cf32, "cs8": band.frames_cs8, "cu8": band.frames_cu8}[fmt](nframes)
    in_format = {"cf32": pkg.abi.SS_FMT_CF32, "cs8": pkg.abi.SS_FMT_CS8, "cu8": pkg.abi.SS_FMT_CU8}[fmt]
    ign = []
    if rng.random() < 0.5:
        lo = center + int(rng.integers(-fs // 3, fs // 4))
        ign = [lo, lo + fs // 20]
    kw = dict(fft_size=n, decim=decim, in_format=in_format, learn_frames=learn, max_batch=max_batch, ignored=ign)
    eng = pkg.SpectrumEngine(fs, center, **kw)
    orc = oracle_mod.oracle_chain(fs, center, **kw)
    retune_at = int(rng.integers(nframes // 2, nframes - 30)) if rng.random() < 0.5 else -1
    pos, outs_g, outs_o = 0, [], []
    while pos < nframes:
        size = int(min(nframes - pos, rng.integers(1, max_batch + 1)))
        if 0 <= retune_at < pos + size and retune_at >= pos:
            size = retune_at - pos if retune_at > pos else size
        if pos == retune_at:
            for c in (eng, orc):  # SdrDevice::setFrequencyRange: new centre (new noise ceiling to learn), Averager reset
                c.set_frequency_range(center + fs - fs // 2, center + fs + fs // 2)
                c.reset()
        outs_g.append(eng.process(iq[pos:pos + size]))
        outs_o.append(orc.process(iq[pos:pos + size]))
        pos += size

    def cat(outs):
        res = {k: np.concatenate([o[k] for o in outs]) for k in ("psd", "rel", "avg", "cand_idx", "cand_avg")}
        counts = np.concatenate([np.diff(o["cand_off"]) for o in outs])
        res["cand_off"] = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        return res

    got, ref = cat(outs_g), cat(outs_o)
    errs, ncand, ndc = check_all(got, ref)
    assert ndc <= max(3, ncand // 1000), (seed, ncand, ndc)
    # (a draw may produce no detection at all: bursts shorter than the 21-frame mean, or a retune right after they start)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("SS_FUZZ_SEEDS2", "10"))))
def test_random_scenario_variants(oracle_mod, seed):
    """The same with the less travelled options mixed in: wall-clock learning (per-frame timestamps), groupings other
    than 21 x 21 (the unfused back end), a caller that asks for no planes or only some, a small candidate capacity
    (SS_ERR_CAND_OVERFLOW keeps cand_off exact), and zero-length calls."""
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.choice([256, 512, 1024, 2048]))
    fs = n * 250
    center = 433_000_000
    nframes = int(rng.integers(120, 200))
    gy, gx = [(21, 21), (21, 21), (5, 7), (21, 5), (3, 21), (1, 1)][int(rng.integers(0, 6))]
    use_clock = rng.random() < 0.5
    period = float(rng.choice([10.0, 20.0, 33.3]))
    learn = int(rng.integers(10, 30))
    max_batch = int(rng.choice([16, 64, 128]))
    band = pkg.synth.SyntheticBand(n, seed=900 + seed, on_frame=(110 if use_clock and period < 30 else 75) if use_clock else learn + 3,
                                   off_frame=nframes - 4)
    iq = band.frames_cf32(nframes)
    kw = dict(fft_size=n, decim=1, learn_frames=learn, max_batch=max_batch, grouping_x=gx, grouping_y=gy)
    if use_clock:
        kw["learn_ms"] = 1500
    eng, orc = pkg.SpectrumEngine(fs, center, **kw), oracle_mod.oracle_chain(fs, center, **kw)
    t_all = np.round(np.arange(nframes) * period).astype(np.int64) + 1_700_000_000_000
    want = [("psd", "rel", "avg"), ("avg",), (), ("psd", "avg")][int(rng.integers(0, 4))]
    small_cap = rng.random() < 0.3
    pos, pairs = 0, []
    while pos < nframes:
        size = int(min(nframes - pos, rng.integers(0, max_batch + 1)))  # zero-length calls allowed
        t = t_all[pos:pos + size] if use_clock else None
        cap = 5 if small_cap else None
        g = eng.process(iq[pos:pos + size], t_ms=t, want=want, cand_cap=cap)
        o = orc.process(iq[pos:pos + size], t_ms=t, want=("psd", "rel", "avg"), cand_cap=cap)
        assert g["status"] == o["status"], (seed, pos, g["status"], o["status"])
        pairs.append((g, o))
        pos += size
    got = {k: np.concatenate([g[k] for g, _ in pairs]) for k in want}
    ref = {k: np.concatenate([o[k] for _, o in pairs]) for k in ("psd", "rel", "avg")}
    for key in ("cand_idx", "cand_avg"):
        got[key] = np.concatenate([g[key] for g, _ in pairs])
        ref[key] = np.concatenate([o[key] for _, o in pairs])
    cg = np.concatenate([np.diff(g["cand_off"]) for g, _ in pairs])
    co = np.concatenate([np.diff(o["cand_off"]) for _, o in pairs])
    if small_cap:
        # lists are truncated per call; the per-frame counts stay exact (up to the start_level band)
        assert np.abs(cg - co).sum() <= max(2, int(co.sum()) // 100)
        return
    got["cand_off"] = np.concatenate([[0], np.cumsum(cg)]).astype(np.int32)
    ref["cand_off"] = np.concatenate([[0], np.cumsum(co)]).astype(np.int32)
    errs, ncand, ndc = check_all(got, ref, gy=gy, gx=gx)
    assert ndc <= max(3, ncand // 1000), (seed, ncand, ndc)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("SS_FUZZ_SEEDS3", "8"))))
def test_entry_points_agree(seed):
    """The three ways into the chain — ss_process (host buffers), ss_process_device (HBM buffers), ss_feed_* (pinned,
    pipelined) — give the same bits on the same random stream and call sizes; ss_read_window returns what the planes
    hold (rows before the batch included)."""
    import torch
    rng = np.random.default_rng(9000 + seed)
    n = int(rng.choice([512, 1024, 4096, 8192]))
    fs, center = n * 250, 145_000_000
    nframes = int(rng.integers(100, 180))
    learn = int(rng.integers(8, 30))
    max_batch = int(rng.choice([16, 48, 64]))
    decim = int(rng.choice([1, 1, 3]))
    band = pkg.synth.SyntheticBand(n, decim=decim, seed=300 + seed, on_frame=learn + 3, off_frame=nframes - 4)
    iq = band.frames_cf32(nframes)
    kw = dict(fft_size=n, decim=decim, learn_frames=learn, max_batch=max_batch)
    host = pkg.SpectrumEngine(fs, center, flags=pkg.abi.SS_FLAG_KEEP_PLANES, **kw)
    devc = pkg.SpectrumEngine(fs, center, **kw)
    fed = pkg.SpectrumEngine(fs, center, **kw)
    feed = fed.feed(depth=3, cand_cap=max_batch * n, want_psd=True)
    dev = torch.device("cuda:0")
    sizes, pos = [], 0
    while pos < nframes:
        s_ = int(min(nframes - pos, rng.integers(1, max_batch + 1)))
        sizes.append(s_)
        pos += s_
    pos, pending = 0, []
    for s_ in sizes:
        chunk = iq[pos:pos + s_]
        h = host.process(chunk)
        # device entry point
        d_iq = torch.from_numpy(chunk.view(np.float32).copy()).to(dev)
        d_psd = torch.empty((s_, n), dtype=torch.float32, device=dev)
        d_avg = torch.empty((s_, n), dtype=torch.float32, device=dev)
        d_off = torch.zeros(s_ + 1, dtype=torch.int32, device=dev)
        d_idx = torch.zeros(s_ * n, dtype=torch.int32, device=dev)
        d_cav = torch.zeros(s_ * n, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()  # (torch's fills run on torch's stream, the library writes from its own)
        devc.process_device(d_iq, s_, psd=d_psd, avg=d_avg, cand_off=d_off, cand_idx=d_idx, cand_avg=d_cav)
        devc.sync()
        off = d_off.cpu().numpy()
        np.testing.assert_array_equal(off, h["cand_off"])
        np.testing.assert_array_equal(d_psd.cpu().numpy(), h["psd"])
        np.testing.assert_array_equal(d_avg.cpu().numpy(), h["avg"])
        np.testing.assert_array_equal(d_idx.cpu().numpy()[: off[-1]], h["cand_idx"])
        np.testing.assert_array_equal(d_cav.cpu().numpy()[: off[-1]], h["cand_avg"])
        # pipelined feed: submit now, compare when collected
        buf = feed.acquire()
        buf[:s_] = chunk.reshape(s_, -1)[:, :n]  # already decimated: the first N samples of each item
        feed.submit(s_, tag=pos)
        pending.append((pos, h))
        if feed.pending == 3 or pos + s_ == nframes:
            while pending:
                p0, hh = pending.pop(0)
                r = feed.collect()
                assert r["tag"] == p0
                np.testing.assert_array_equal(r["cand_off"], hh["cand_off"])
                np.testing.assert_array_equal(r["cand_idx"], hh["cand_idx"])
                np.testing.assert_array_equal(r["cand_avg"], hh["cand_avg"])
                np.testing.assert_array_equal(r["psd"], hh["psd"])
        # windows of the planes the host context kept, a few random ones per call
        for _ in range(3):
            f = int(rng.integers(0, s_))
            lo = int(rng.integers(0, n - 1))
            hi = int(rng.integers(lo + 1, min(n, lo + 200) + 1))
            for plane, key in ((pkg.abi.SS_PLANE_PSD, "psd"), (pkg.abi.SS_PLANE_REL, "rel"), (pkg.abi.SS_PLANE_AVG, "avg")):
                np.testing.assert_array_equal(host.read_window(plane, f, lo, hi), h[key][f, lo:hi])
        if pos > 0:  # ring rows: frame -k is the k-th newest frame before this call (what getBestIndex walks back over)
            k = int(rng.integers(1, min(20, pos) + 1))
            np.testing.assert_array_equal(host.read_window(pkg.abi.SS_PLANE_REL, -k, 0, n), all_rel[pos - k])
        all_rel = h["rel"] if pos == 0 else np.concatenate([all_rel, h["rel"]])
        pos += s_
    feed.close()


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("SS_FUZZ_SEEDS4", "6"))))
@pytest.mark.parametrize("impl", ["detect", "standalone"])
def test_random_spectrogram_sessions(oracle_mod, seed, impl, monkeypatch, diag_lib):
    """The spectrogram side branch under random sizes (its output size follows min(16384, getFft(fs, 1000))), call sizes,
    send times and a retune (the accumulator is per centre frequency, spectrogram.cpp:29-60); accumulated inside the
    detect kernel (decimation factors up to 64) or by the stand-alone kernels."""
    import ctypes as C
    monkeypatch.setenv("SS_SPEC_IMPL", impl)
    rng = np.random.default_rng(12000 + seed)
    n = int(rng.choice([1024, 4096, 8192, 16384]))
    fs = n * int(rng.choice([1, 4, 16, 125, 250, 1000, 2000]))  # decimation factors 512 (always stand-alone), 128, 32, 8, 4, 1, 1
    center = 100_000_000
    nframes = int(rng.integers(80, 160))
    max_batch = int(rng.choice([8, 32, 64]))
    band = pkg.synth.SyntheticBand(n, seed=77 + seed, on_frame=20, off_frame=nframes - 5)
    iq = band.frames_cf32(nframes)
    kw = dict(fft_size=n, decim=1, learn_frames=10, max_batch=max_batch)
    eng = pkg.SpectrumEngine(fs, center, flags=pkg.abi.SS_FLAG_SPECTROGRAM, **kw)
    orc = oracle_mod.oracle_chain(fs, center, **kw)
    L = oracle_mod.lib()
    accs = {}  # centre -> oracle accumulator (the reference keeps one container per centre frequency)

    def acc_for(c):
        if c not in accs:
            accs[c] = L.orc_spectrogram_create(n, fs)
        return accs[c]

    size = L.orc_spectrogram_size(acc_for(center))
    assert eng._lib.ss_spectrogram_size(eng._h) == size
    cur = center
    retune_at = int(rng.integers(30, nframes - 30)) if rng.random() < 0.6 else -1
    pos = sends = 0
    while pos < nframes:
        s_ = int(min(nframes - pos, rng.integers(1, max_batch + 1)))
        if pos <= retune_at < pos + s_:
            s_ = max(1, retune_at - pos) if retune_at > pos else s_
        if pos == retune_at:
            cur = center + fs
            for c in (eng, orc):
                c.set_frequency_range(cur - fs // 2, cur + fs // 2)
                c.reset()
        eng.process(iq[pos:pos + s_], want=())
        for row in orc.process(iq[pos:pos + s_], want=("psd",))["psd"]:
            L.orc_spectrogram_process(acc_for(cur), row.ctypes.data_as(C.POINTER(C.c_float)))
        pos += s_
        if rng.random() < 0.25 or pos == nframes:
            want8, wantf = np.zeros(size, np.int8), np.zeros(size, np.float32)
            cnt_ref = L.orc_spectrogram_send(acc_for(cur), want8.ctypes.data_as(C.POINTER(C.c_int8)), wantf.ctypes.data_as(C.POINTER(C.c_float)))
            got8, gotf, cnt = eng.spectrogram_read()
            assert cnt == cnt_ref, (seed, pos, cnt, cnt_ref)
            if cnt:
                sends += 1
                assert np.max(np.abs(gotf - wantf)) < 1e-4 * 60
                frac = np.abs(wantf - np.trunc(wantf))
                decided = (frac > 1e-3) & (frac < 1 - 1e-3)
                np.testing.assert_array_equal(got8[decided], want8[decided])
    assert sends >= 1
    for g in accs.values():
        L.orc_spectrogram_destroy(g)
