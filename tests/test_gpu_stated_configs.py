"""Parity on BASELINE.json's own configurations at their stated shapes, against the REFERENCE'S OWN code (oracle/_ref:
psd.cpp / noise_learner.cpp / transmission.cpp / averager.cpp / utils.cpp compiled in place), not only the C restatement:

  config 2   8192 points, 1024 frames in ONE call                                  (CF32)
  config 3   65536 points, 128-frame calls, int8 IQ, candidates only (detect mode)
  config 5   2^20 points, 16-frame calls, 21 x 21 grouping (the fused back end)
  golden     N = 8192 fixture made by the reference (tests/golden/ref_big_n8192_cs8.npz), which also travels to boxes without _ref

Every case prints the achieved error quantiles per plane and the number of candidates inside the +-1e-3 dB band.
Needs an MI355X: run with -m gpu."""
import os

import numpy as np
import pytest

import rtl_sdr_scanner_cpp_amd as pkg
from parity import (BAND, all_bins_vs_fp64, cand_set, check_all, check_plane, dont_care_limit, error_quantiles, excess_vs_fp64, floor_tolerance, format_all_bins, running_sum_drift_at,
                    format_excess, format_quantiles, strict_excess)

pytestmark = pytest.mark.gpu

CENTER = 145_000_000


def _cat(outs, keys):
    res = {k: np.concatenate([o[k] for o in outs]) for k in keys if k in outs[0]}
    counts = np.concatenate([np.diff(o["cand_off"]) for o in outs])
    res["cand_off"] = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    return res


def _ref_result(r):
    off = np.zeros(len(r["cands"]) + 1, np.int32)
    off[1:] = np.cumsum([len(c) for c in r["cands"]])
    idx = np.concatenate(r["cands"]).astype(np.int32) if off[-1] else np.zeros(0, np.int32)
    return {"psd": r["psd"], "rel": r["rel"], "avg": r["avg"], "cand_off": off, "cand_idx": idx}


def _report(name, got, ref, ncand, ndc, iq=None, fs=None):
    print(f"\n[{name}] {ncand} reference candidates, {ndc} inside the {BAND} dB band; |err| dB: {format_quantiles(error_quantiles(got, ref))}")
    # what the bare 1e-4 * max(1, |ref|) does not cover, and — where the input is at hand — that the engine is no farther from an
    # fp64 FFT on those bins than the reference's own fp32 FFT is (asserted inside excess_vs_fp64)
    vs64 = excess_vs_fp64(iq, got["psd"], ref["psd"], fs) if iq is not None and "psd" in got else None
    print(f"[{name}] outside the bare 1e-4 tolerance: {format_excess(strict_excess(got, ref), vs64)}")
    if iq is not None and "psd" in got:
        # ... and over ALL bins: is the engine's transform systematically farther from the truth than the reference's? The rms has a heavy
        # tail — five of half a million bins hold a quarter to a half of the sum of squares (deep nulls: the fp32 floor times the null's
        # depth), so its ratio moves by +-0.2 with the data (scripts/fft8192_accuracy_model.py: 0.79 .. 1.00 over six bands for a transform
        # whose median ratio is 0.96 every time) and stays held to 1.5 x; what is systematic shows in the 99th percentile — the bins the
        # FFT's rounding floor decides — held to 1.15 x (measured 0.98 at 8192 points, 1.03 at 262144). (The MEDIAN, 2-4e-6 dB, is the dB
        # conversion's, not the transform's: hardware log2 times 3.0103 against log10f — 1.16 x at 8192 points, 2.2 x at 262144, both
        # thirty times below the contract's 1e-4.)
        allb = all_bins_vs_fp64(iq, got["psd"], ref["psd"], fs)
        print(f"[{name}] {format_all_bins(allb)}")
        assert allb is None or allb["engine_over_reference_rms"] <= 1.5, allb
        assert allb is None or allb["engine_over_reference_p99"] <= 1.15, allb


def test_config2_8192_points_1024_frames_in_one_call(ref_mod):
    n, fs, nframes = 8192, 2_048_000, 1024
    band = pkg.synth.SyntheticBand(n, seed=21, on_frame=150, off_frame=700, period=900)
    iq = band.frames_cf32(nframes)
    t = (10_000 + 20 * np.arange(nframes)).astype(np.int64)  # 50 frames per second: learning ends after 101 frames
    ref_mod.ref().orc_set_fft_backend(0)
    ref = _ref_result(ref_mod.RefChain(n, fs, CENTER - fs // 2, CENTER + fs // 2).process(iq, t))
    eng = pkg.SpectrumEngine(fs, CENTER, fft_size=n, decim=1, max_batch=nframes)
    got = eng.process(iq, t_ms=t)
    errs, ncand, ndc = check_all(got, ref)
    _report("config 2: 8192 x 1024, one call", got, ref, ncand, ndc, iq, fs)
    assert ncand > 50_000 and ndc <= dont_care_limit(ncand), (ncand, ndc)


def test_config3_65536_points_int8_candidates_only(ref_mod):
    n, fs, nframes, chunk = 65536, 20_000_000, 256, 128
    band = pkg.synth.SyntheticBand(n, seed=22, on_frame=60, off_frame=230)
    iq8 = band.frames_cs8(nframes)
    iq = (iq8[..., 0].astype(np.float32) / np.float32(128.0) + 1j * (iq8[..., 1].astype(np.float32) / np.float32(128.0))).astype(np.complex64)
    t = (10_000 + 50 * np.arange(nframes)).astype(np.int64)
    ref_mod.ref().orc_set_fft_backend(0)
    ref = _ref_result(ref_mod.RefChain(n, fs, CENTER - fs // 2, CENTER + fs // 2).process(iq, t))
    eng = pkg.SpectrumEngine(fs, CENTER, fft_size=n, decim=1, in_format=pkg.abi.SS_FMT_CS8, max_batch=chunk)
    outs = [eng.process(iq8[a:a + chunk], t_ms=t[a:a + chunk], want=()) for a in range(0, nframes, chunk)]  # detect mode: no plane leaves the device
    got = _cat(outs, ("cand_idx", "cand_avg"))
    a, b = cand_set(got["cand_off"], got["cand_idx"]), cand_set(ref["cand_off"], ref["cand_idx"])
    near = np.abs(ref["avg"] - np.float32(8.0)) < BAND
    outside = [(f, i) for (f, i) in a ^ b if not near[f, i]]
    assert not outside, sorted(outside)[:10]
    # the sort key the tracker gets (transmission.cpp:95) is the reference's avg at the candidate
    frames = np.repeat(np.arange(nframes), np.diff(got["cand_off"]))
    check_plane("cand_avg", got["cand_avg"][None], ref["avg"][frames, got["cand_idx"]][None], floor=running_sum_drift_at(n, ref["avg"], frames, got["cand_idx"])[None])
    print(f"\n[config 3: 65536 x 128, CS8, detect] {len(b)} reference candidates, {len(a ^ b)} inside the {BAND} dB band; "
          f"cand_avg max |err| {np.abs(got['cand_avg'] - ref['avg'][frames, got['cand_idx']]).max():.1e} dB")
    assert len(b) > 10_000 and len(a ^ b) <= dont_care_limit(len(b))


def test_config5_one_million_points_fused_back_end(ref_mod):
    n, fs, nframes, chunk = 1 << 20, 61_440_000, 48, 16
    band = pkg.synth.SyntheticBand(n, seed=23, on_frame=27, off_frame=46)
    iq = band.frames_cf32(nframes)
    t = (10_000 + 400 * np.arange(nframes)).astype(np.int64)  # learning ends after 6 frames, the averager is full at frame 26
    ref_mod.ref().orc_set_fft_backend(0)
    ref = _ref_result(ref_mod.RefChain(n, fs, CENTER - fs // 2, CENTER + fs // 2).process(iq, t))
    eng = pkg.SpectrumEngine(fs, CENTER, fft_size=n, decim=1, max_batch=chunk)  # grouping 21 x 21: k_detect_fused
    outs = [eng.process(iq[a:a + chunk], t_ms=t[a:a + chunk]) for a in range(0, nframes, chunk)]
    got = _cat(outs, ("psd", "rel", "avg", "cand_idx", "cand_avg"))
    errs, ncand, ndc = check_all(got, ref)
    _report("config 5: 2^20 x 16-frame calls, 21 x 21", got, ref, ncand, ndc, iq, fs)
    assert ncand > 1000 and ndc <= dont_care_limit(ncand), (ncand, ndc)


def test_headline_size_golden_made_by_the_reference():
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_big_n8192_cs8.npz"))
    n, fs, center, sub = int(g["n"]), int(g["fs"]), int(g["center"]), int(g["sub"])
    eng = pkg.SpectrumEngine(fs, center, fft_size=n, decim=1, in_format=pkg.abi.SS_FMT_CS8, max_batch=40)
    iq8, t = g["iq8"], g["t_ms"]
    outs = [eng.process(iq8[a:a + 40], t_ms=t[a:a + 40]) for a in range(0, iq8.shape[0], 40)]
    got = _cat(outs, ("psd", "rel", "avg", "cand_idx", "cand_avg"))
    floor = floor_tolerance(g["psd_sub"])  # (frame means from every 8th bin: the same to a fraction of a dB)
    quant = {}
    for k in ("psd", "rel", "avg"):
        check_plane(k, got[k][:, ::sub], g[k + "_sub"], floor if k != "avg" else floor.mean() + 2e-4)
        quant.update(error_quantiles({k: got[k][:, ::sub]}, {k: g[k + "_sub"]}, planes=(k,)))
    a, b = cand_set(got["cand_off"], got["cand_idx"]), cand_set(g["cand_off"], g["cand_idx"])
    ref_avg_at = dict(zip(zip(np.repeat(np.arange(len(g["cand_off"]) - 1), np.diff(g["cand_off"])).tolist(), g["cand_idx"].tolist()), g["cand_avg"].tolist()))
    for (f, i) in a ^ b:  # only bins whose avg sits on the threshold may differ
        v = ref_avg_at.get((f, i), float(got["avg"][f, i]))
        assert abs(v - 8.0) < 2 * BAND, (f, i, v)
    print(f"\n[golden N = 8192, int8] {len(b)} reference candidates, {len(a ^ b)} on the threshold; |err| dB: {format_quantiles(quant)}")
    assert len(b) > 1000 and len(a ^ b) <= 2
    check_plane("noise ceiling", eng.read_noise()[0][None], g["thr"][None], floor_tolerance(g["thr"][None]))


def _device_calls(eng, iq_batches, n, planes=True):
    """ss_process_device, one call per batch, NO synchronisation in between (what bench.py times: up to five calls in flight on
    the library's two queues, a fresh output set per call, the PSD plane the only plane handed out)."""
    import torch
    dev = torch.device("cuda", 0)
    d_iq = [torch.from_numpy(b.view(np.float32) if b.dtype == np.complex64 else b).to(dev) for b in iq_batches]
    outs = []
    for b in iq_batches:
        nb = b.shape[0]
        outs.append(dict(psd=torch.empty((nb, n), dtype=torch.float32, device=dev) if planes else None,
                         off=torch.zeros(nb + 1, dtype=torch.int32, device=dev), idx=torch.empty(nb * 1024, dtype=torch.int32, device=dev),
                         avg=torch.empty(nb * 1024, dtype=torch.float32, device=dev)))
    torch.cuda.synchronize()
    for d, o in zip(d_iq, outs):
        eng.process_device(d, d.shape[0], psd=o["psd"], cand_off=o["off"], cand_idx=o["idx"], cand_avg=o["avg"])
    eng.sync()
    res = []
    for o in outs:
        off = o["off"].cpu().numpy()
        res.append({"cand_off": off, "cand_idx": o["idx"].cpu().numpy()[:off[-1]], "cand_avg": o["avg"].cpu().numpy()[:off[-1]],
                    **({"psd": o["psd"].cpu().numpy()} if planes else {})})
    return res


def test_config2_the_timed_path_against_the_reference(ref_mod):
    """What bench.py times — consecutive 1024-frame ss_process_device calls with nothing in between: stages of five calls in
    flight, halo frames re-transformed, tiles culled, lists handed from plan to FFT workgroups inside a launch — compared
    DIRECTLY with the reference's own code, not with the engine's host path."""
    n, fs, nb, ncalls = 8192, 2_048_000, 1024, 7
    band = pkg.synth.SyntheticBand(n, seed=41, on_frame=150, off_frame=560, period=800)
    iq = band.frames_cf32(nb * ncalls)
    t = (10_000 + 20 * np.arange(nb * ncalls)).astype(np.int64)  # learning ends after 101 frames
    ref_mod.ref().orc_set_fft_backend(0)
    ref = _ref_result(ref_mod.RefChain(n, fs, CENTER - fs // 2, CENTER + fs // 2).process(iq, t))
    eng = pkg.SpectrumEngine(fs, CENTER, fft_size=n, decim=1, max_batch=nb, learn_frames=101)
    outs = _device_calls(eng, [iq[k * nb:(k + 1) * nb] for k in range(ncalls)], n)
    got = _cat(outs, ("psd", "cand_idx", "cand_avg"))
    check_plane("psd", got["psd"], ref["psd"], floor_tolerance(ref["psd"]))
    a, b = cand_set(got["cand_off"], got["cand_idx"]), cand_set(ref["cand_off"], ref["cand_idx"])
    near = np.abs(ref["avg"] - np.float32(8.0)) < BAND
    outside = [(f, i) for (f, i) in a ^ b if not near[f, i]]
    assert not outside, sorted(outside)[:10]
    frames = np.repeat(np.arange(nb * ncalls), np.diff(got["cand_off"]))
    check_plane("cand_avg", got["cand_avg"][None], ref["avg"][frames, got["cand_idx"]][None], floor=running_sum_drift_at(n, ref["avg"], frames, got["cand_idx"])[None])
    print(f"\n[config 2, timed path: {ncalls} x 1024 frames in flight] {len(b)} reference candidates, {len(a ^ b)} inside the {BAND} dB band; "
          f"|err| dB: {format_quantiles(error_quantiles({'psd': got['psd']}, {'psd': ref['psd']}, planes=('psd',)))}")
    print(f"[config 2, timed path] outside the bare 1e-4 tolerance: "
          f"{format_excess(strict_excess({'psd': got['psd']}, {'psd': ref['psd']}, planes=('psd',)), excess_vs_fp64(iq, got['psd'], ref['psd'], fs))}")
    assert len(b) > 300_000 and len(a ^ b) <= dont_care_limit(len(b))


@pytest.mark.parametrize("chunk,ncalls", [(128, 3), (300, 2)])
def test_config3_device_calls_against_the_reference(ref_mod, chunk, ncalls):
    """(300-frame calls: the library takes a 65536-point call of more than 256 frames through in chunks of 256 — a chunk's work buffer
    stays in the Infinity Cache between its column and its row half —, the deferred stages riding on the first chunk's launches.)"""
    n, fs = 65536, 20_000_000
    band = pkg.synth.SyntheticBand(n, seed=42, on_frame=60, off_frame=chunk * ncalls - 54)
    iq8 = band.frames_cs8(chunk * ncalls)
    iq = (iq8[..., 0].astype(np.float32) / np.float32(128.0) + 1j * (iq8[..., 1].astype(np.float32) / np.float32(128.0))).astype(np.complex64)
    t = (10_000 + 50 * np.arange(chunk * ncalls)).astype(np.int64)  # learning ends after 41 frames
    ref_mod.ref().orc_set_fft_backend(0)
    ref = _ref_result(ref_mod.RefChain(n, fs, CENTER - fs // 2, CENTER + fs // 2).process(iq, t))
    eng = pkg.SpectrumEngine(fs, CENTER, fft_size=n, decim=1, in_format=pkg.abi.SS_FMT_CS8, max_batch=chunk, learn_frames=41)
    outs = _device_calls(eng, [iq8[k * chunk:(k + 1) * chunk] for k in range(ncalls)], n, planes=False)
    got = _cat(outs, ("cand_idx", "cand_avg"))
    a, b = cand_set(got["cand_off"], got["cand_idx"]), cand_set(ref["cand_off"], ref["cand_idx"])
    near = np.abs(ref["avg"] - np.float32(8.0)) < BAND
    outside = [(f, i) for (f, i) in a ^ b if not near[f, i]]
    assert not outside, sorted(outside)[:10]
    print(f"\n[config 3, device calls: {ncalls} x {chunk} frames of 65536 points, CS8] {len(b)} reference candidates, {len(a ^ b)} inside the {BAND} dB band")
    assert len(b) > 10_000 and len(a ^ b) <= dont_care_limit(len(b))


@pytest.mark.parametrize("n,chunk,ncalls", [(65536, 128, 10), (65536, 64, 12), (65536, 16, 30), (65536, 256, 4), (131072, 64, 8), (131072, 24, 12)])
def test_config3_steady_state_of_the_shipped_form_against_the_reference(ref_mod, n, chunk, ncalls):
    """Config 3 at depth on the PRODUCT library, against the reference's own code: the learning frames as a call of their own (which
    takes the four-step form and leaves its rows in bin order), then `ncalls` int8 detect-mode ss_process_device calls with nothing in
    between — the form bench.py times: one launch per call, every launch carrying the fold of its own call, the plan of the call
    before, the listed tiles of the call before that and the candidate lists of a fourth (csrc/scan_step.h KIND 8), so from the fourth
    call on every launch is a steady-state one. ss_get_stats shows that tiles were culled and that nothing drained the pipeline between
    the first of those calls and the last (reference chain: transmission.cpp:57-68,88-96, averager.cpp:14-25,52-61, utils.cpp:31-53)."""
    import torch
    fs, learn = 20_000_000, 41  # (131072 points: the size getFft picks for this signal — the same fold with radix 16, KIND 9)
    total = learn + chunk * ncalls
    band = pkg.synth.SyntheticBand(n, seed=45, on_frame=learn + 30, off_frame=total - 40, period=total)
    iq8 = band.frames_cs8(total)
    iq = (iq8[..., 0].astype(np.float32) / np.float32(128.0) + 1j * (iq8[..., 1].astype(np.float32) / np.float32(128.0))).astype(np.complex64)
    t = (10_000 + 50 * np.arange(total)).astype(np.int64)  # learning ends after 41 frames
    ref_mod.ref().orc_set_fft_backend(0)
    ref = _ref_result(ref_mod.RefChain(n, fs, CENTER - fs // 2, CENTER + fs // 2).process(iq, t))
    del ref["psd"], ref["rel"]
    eng = pkg.SpectrumEngine(fs, CENTER, fft_size=n, decim=1, in_format=pkg.abi.SS_FMT_CS8, max_batch=max(chunk, learn), learn_frames=learn)
    dev = torch.device("cuda", 0)
    cuts = [(0, learn)] + [(learn + k * chunk, learn + (k + 1) * chunk) for k in range(ncalls)]
    d_iq = [torch.from_numpy(iq8[a:b]).to(dev) for a, b in cuts]
    outs = [dict(off=torch.zeros(b - a + 1, dtype=torch.int32, device=dev), idx=torch.empty((b - a) * 1024, dtype=torch.int32, device=dev)) for a, b in cuts]
    torch.cuda.synchronize()
    eng.process_device(d_iq[0], learn, cand_off=outs[0]["off"], cand_idx=outs[0]["idx"])
    eng.process_device(d_iq[1], chunk, cand_off=outs[1]["off"], cand_idx=outs[1]["idx"])  # (the change of form drains the learning call's stages: counted before the run)
    drains_before = eng.stats()["drains"]
    for d, o in zip(d_iq[2:], outs[2:]):
        eng.process_device(d, chunk, cand_off=o["off"], cand_idx=o["idx"])
    drains_after = eng.stats()["drains"]
    eng.sync()
    st = eng.stats()
    offs = [o["off"].cpu().numpy() for o in outs]
    got_off = np.concatenate([[0], np.cumsum(np.concatenate([np.diff(x) for x in offs]))]).astype(np.int32)
    got_idx = np.concatenate([o["idx"].cpu().numpy()[:x[-1]] for o, x in zip(outs, offs)])
    a, b = cand_set(got_off, got_idx), cand_set(ref["cand_off"], ref["cand_idx"])
    near = np.abs(ref["avg"] - np.float32(8.0)) < BAND
    outside = [(f, i) for (f, i) in a ^ b if not near[f, i]]
    assert not outside, sorted(outside)[:10]
    print(f"\n[config 3 at depth, product library: learning call + {ncalls} x {chunk} frames of {n} points, CS8] {len(b)} reference candidates, {len(a ^ b)} inside the {BAND} dB band; "
          f"tiles {st['tiles_total']}, tested {st['tiles_tested']}, culled {st['tiles_culled']}; drains during the run {drains_after - drains_before}")
    # (calls shorter than the averager's window slide it along the ring's buffer and bring it back to the front when the end is reached —
    # a drain once in a few dozen calls, csrc/ring_place.h; calls of a window and more go round the buffer without one)
    assert drains_after - drains_before <= (0 if chunk >= 35 else 1 + ncalls // 20), (drains_before, drains_after)
    assert st["culling"] and st["tiles_culled"] > 0 and st["tiles_culled"] <= st["tiles_tested"] <= st["tiles_total"], st
    assert len(b) > 10_000 and len(a ^ b) <= dont_care_limit(len(b))
    assert st["tiles_culled"] > 0.5 * st["tiles_tested"], st  # (the band leaves a few per cent of the tiles to evaluate)


@pytest.mark.parametrize("n,fs,fmt,chunk,ncalls", [(1 << 17, 20_000_000, "cs8", 32, 4), (1 << 18, 61_440_000, "cf32", 16, 5), (1 << 18, 61_440_000, "cf32", 32, 4), (1 << 18, 61_440_000, "cs8", 40, 3)])
def test_the_sizes_getfft_would_pick_device_calls_against_the_reference(ref_mod, n, fs, fmt, chunk, ncalls):
    """The transform sizes the reference itself would run the signals of configs 3 and 5 at — getFft(20 MS/s, 250 Hz) = 131072 and
    getFft(61.44 MS/s, 250 Hz) = 262144 (utils/radio_utils.cpp:98-104, tests/test_radio_utils.cpp:4-16) — as detect-mode
    ss_process_device calls with nothing in between, DIRECTLY against the reference's own code (not only its C restatement)."""
    learn = 21
    total = chunk * ncalls
    band = pkg.synth.SyntheticBand(n, seed=46, on_frame=learn + 24, off_frame=total - 12)
    if fmt == "cs8":
        raw = band.frames_cs8(total)
        iq = (raw[..., 0].astype(np.float32) / np.float32(128.0) + 1j * (raw[..., 1].astype(np.float32) / np.float32(128.0))).astype(np.complex64)
    else:
        raw = iq = band.frames_cf32(total)
    t = (10_000 + 100 * np.arange(total)).astype(np.int64)  # learning ends after 21 frames
    ref_mod.ref().orc_set_fft_backend(0)
    ref = _ref_result(ref_mod.RefChain(n, fs, CENTER - fs // 2, CENTER + fs // 2).process(iq, t))
    del ref["psd"], ref["rel"]
    eng = pkg.SpectrumEngine(fs, CENTER, fft_size=n, decim=1, in_format=pkg.abi.SS_FMT_CS8 if fmt == "cs8" else pkg.abi.SS_FMT_CF32, max_batch=chunk, learn_frames=learn)
    outs = _device_calls(eng, [raw[k * chunk:(k + 1) * chunk] for k in range(ncalls)], n, planes=False)
    st = eng.stats()
    got = _cat(outs, ("cand_idx", "cand_avg"))
    a, b = cand_set(got["cand_off"], got["cand_idx"]), cand_set(ref["cand_off"], ref["cand_idx"])
    near = np.abs(ref["avg"] - np.float32(8.0)) < BAND
    outside = [(f, i) for (f, i) in a ^ b if not near[f, i]]
    assert not outside, sorted(outside)[:10]
    frames = np.repeat(np.arange(total), np.diff(got["cand_off"]))
    check_plane("cand_avg", got["cand_avg"][None], ref["avg"][frames, got["cand_idx"]][None], floor=running_sum_drift_at(n, ref["avg"], frames, got["cand_idx"])[None])
    print(f"\n[getFft's own size: {ncalls} x {chunk} frames of {n} points, {fmt}, detect mode] {len(b)} reference candidates, {len(a ^ b)} inside the {BAND} dB band; "
          f"tiles {st['tiles_total']}, tested {st['tiles_tested']}, culled {st['tiles_culled']}")
    assert len(b) > 2000 and len(a ^ b) <= dont_care_limit(len(b))
    # both sizes go through culled chains (131072: the radix-16 fold, round 5; 262144: the 1024-point row tile behind 256-point columns, round 6)
    assert st["culling"] and st["tiles_culled"] > 0 and st["tiles_culled"] <= st["tiles_tested"] <= st["tiles_total"], st


@pytest.mark.parametrize("chunk,ncalls", [(16, 8), (40, 3)])
def test_config5_device_calls_against_the_reference(ref_mod, chunk, ncalls):
    """Config 5 as it ships and as bench.py times it: 2^20-point CF32 frames in 16-frame ss_process_device calls, detect mode (no
    plane handed out), tile culling on, eight calls enqueued back to back with no synchronisation — compared DIRECTLY with the
    reference's own code. ss_get_stats shows that tiles really were culled. (40-frame calls: the library takes a call of more than 16
    frames through in chunks of 16, 16 + 16 + 8 here, the first of them holding the 32 learning frames.)"""
    n, fs, learn = 1 << 20, 61_440_000, 32
    band = pkg.synth.SyntheticBand(n, seed=43, on_frame=70, off_frame=118)
    iq = band.frames_cf32(chunk * ncalls)
    t = (10_000 + 30 * np.arange(chunk * ncalls)).astype(np.int64)
    t[learn - 1:] += 2_000  # the reference's wall clock (noise_learner.cpp:23) ends learning with frame learn - 1: `learn` frames, as ss_process_device counts them
    ref_mod.ref().orc_set_fft_backend(0)
    ref = _ref_result(ref_mod.RefChain(n, fs, CENTER - fs // 2, CENTER + fs // 2).process(iq, t))
    del ref["psd"], ref["rel"]
    eng = pkg.SpectrumEngine(fs, CENTER, fft_size=n, decim=1, max_batch=chunk, learn_frames=learn)
    outs = _device_calls(eng, [iq[k * chunk:(k + 1) * chunk] for k in range(ncalls)], n, planes=False)
    st = eng.stats()
    got = _cat(outs, ("cand_idx", "cand_avg"))
    a, b = cand_set(got["cand_off"], got["cand_idx"]), cand_set(ref["cand_off"], ref["cand_idx"])
    near = np.abs(ref["avg"] - np.float32(8.0)) < BAND
    outside = [(f, i) for (f, i) in a ^ b if not near[f, i]]
    assert not outside, sorted(outside)[:10]
    frames = np.repeat(np.arange(chunk * ncalls), np.diff(got["cand_off"]))
    check_plane("cand_avg", got["cand_avg"][None], ref["avg"][frames, got["cand_idx"]][None], floor=running_sum_drift_at(n, ref["avg"], frames, got["cand_idx"])[None])
    print(f"\n[config 5, device calls: {ncalls} x {chunk} frames of 2^20 points, detect mode, culled] {len(b)} reference candidates, {len(a ^ b)} inside the {BAND} dB band; "
          f"tiles {st['tiles_total']}, tested {st['tiles_tested']}, culled {st['tiles_culled']}")
    assert st["culling"] and st["tiles_culled"] > 0 and st["tiles_culled"] <= st["tiles_tested"] <= st["tiles_total"], st
    assert len(b) > 2000 and len(a ^ b) <= dont_care_limit(len(b))


@pytest.mark.parametrize("chunk,ncalls", [(16, 5), (48, 4), (37, 12)])
def test_config5_calls_that_keep_no_db_plane_serve_the_tracker_all_the_same(chunk, ncalls):
    """A 2^20-point ss_process_device call in detect mode writes no dB plane (its rows go straight to the averager ring's buffer as
    noise-relative values — behind the window for calls shorter than the ring, as a region of their own otherwise, wrapping around
    the buffer's end in the longest case here —, include/specscan.h): ss_read_window(SS_PLANE_REL) — what the signal tracker reads —
    gives the bits a call WITH a plane gives, the ring rows before the batch included; SS_PLANE_PSD says that it is not there; and
    the candidate lists are those of the call with a plane."""
    import torch
    n, fs, learn = 1 << 20, 61_440_000, chunk
    dev = torch.device("cuda", 0)
    band = pkg.synth.SyntheticBand(n, seed=44, on_frame=chunk + 24, off_frame=chunk * ncalls - 10)
    d_iq = [torch.from_numpy(band.frames_cs8(chunk)).to(dev) for k in range(ncalls)]
    res = {}
    for with_plane in (False, True):
        eng = pkg.SpectrumEngine(fs, CENTER, fft_size=n, decim=1, in_format=pkg.abi.SS_FMT_CS8, max_batch=chunk, learn_frames=learn)
        outs = []
        for d in d_iq:
            o = dict(psd=torch.empty((chunk, n), dtype=torch.float32, device=dev) if with_plane else None, off=torch.zeros(chunk + 1, dtype=torch.int32, device=dev),
                     idx=torch.empty(chunk * 1024, dtype=torch.int32, device=dev))
            eng.process_device(d, chunk, psd=o["psd"], cand_off=o["off"], cand_idx=o["idx"])
            outs.append(o)
        eng.sync()
        wins = [eng.read_window(pkg.abi.SS_PLANE_REL, f, lo, lo + 300) for f in (-20, -1, 0, 7, chunk - 1) for lo in (0, 524_000, n - 300)]
        if with_plane:
            eng.read_window(pkg.abi.SS_PLANE_PSD, 3, 0, 64)
        else:
            with pytest.raises(pkg.abi.SpecscanError, match="kept no dB"):
                eng.read_window(pkg.abi.SS_PLANE_PSD, 3, 0, 64)
        offs = [o["off"].cpu().numpy() for o in outs]
        res[with_plane] = (wins, offs, [o["idx"].cpu().numpy()[:x[-1]] for o, x in zip(outs, offs)])
        eng.close()
    for a, b in zip(res[False][0], res[True][0]):
        np.testing.assert_array_equal(a, b)
    for k in range(ncalls):
        np.testing.assert_array_equal(res[False][1][k], res[True][1][k])
        np.testing.assert_array_equal(res[False][2][k], res[True][2][k])
    assert sum(int(x[-1]) for x in res[True][1]) > 1000


@pytest.mark.parametrize("n,fs,chunk,ncalls", [(65536, 20_000_000, 48, 3), (65536, 20_000_000, 20, 4), (1 << 20, 61_440_000, 40, 2), (1 << 18, 61_440_000, 40, 3), (1 << 18, 61_440_000, 16, 4)])
def test_read_window_after_a_retune_does_not_subtract_the_ceiling_twice(n, fs, chunk, ncalls):
    """ss_set_frequency_range settles the ring window's dB rows in place (settle_ring_db: the newest <= 35 rows of the last detect-mode
    call become noise-relative). ss_read_window(SS_PLANE_REL) of that last call must give the same values before and after the retune:
    rows the settle pass has rewritten are read as they are, the earlier frames of a call longer than the window still have the
    ceiling subtracted on the way out (round 5's advisor: they were subtracted twice, with SS_OK)."""
    import torch
    dev = torch.device("cuda", 0)
    band = pkg.synth.SyntheticBand(n, seed=45, on_frame=chunk + 10, off_frame=chunk * ncalls + 100)
    eng = pkg.SpectrumEngine(fs, CENTER, fft_size=n, decim=1, in_format=pkg.abi.SS_FMT_CS8, max_batch=chunk, learn_frames=chunk)
    keep = []
    for k in range(ncalls):
        d = torch.from_numpy(band.frames_cs8(chunk)).to(dev)
        o = dict(off=torch.zeros(chunk + 1, dtype=torch.int32, device=dev), idx=torch.empty(chunk * 1024, dtype=torch.int32, device=dev))
        eng.process_device(d, chunk, cand_off=o["off"], cand_idx=o["idx"])
        keep.append((d, o))
    eng.sync()
    frames = sorted({-20, -1, 0, 3, chunk - 36, chunk - 35, chunk - 34, chunk - 1} & set(range(-20, chunk)))
    spots = [(f, lo) for f in frames for lo in (0, n // 2 - 150, n - 300)]
    before = [eng.read_window(pkg.abi.SS_PLANE_REL, f, lo, lo + 300) for f, lo in spots]
    eng.set_frequency_range(CENTER - fs // 4, CENTER + fs // 4)
    after = [eng.read_window(pkg.abi.SS_PLANE_REL, f, lo, lo + 300) for f, lo in spots]
    for (f, lo), a, b in zip(spots, before, after):
        np.testing.assert_array_equal(a, b, err_msg=f"frame {f}, bins from {lo}")
    # (noise-relative values of a noisy band — or the -100 of a learning frame's row —, nowhere near a ceiling subtracted twice)
    assert all(float(np.abs(a[a != np.float32(-100.0)]).max(initial=0.0)) < 60.0 for a in before)
    eng.close()
