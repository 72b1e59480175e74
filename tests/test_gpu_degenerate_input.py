"""Degenerate input: an all-zero frame. PSD::work gives -inf for every bin (log10f(0), psd.cpp:19), exactly like the engine.
Downstream the reference never recovers on its own: Averager::subtract turns -inf - (-inf) into NaN when the row leaves the
21-frame window (averager.cpp:40-50), average() drags NaN along the rest of each row (utils.cpp:39-48), and both stay NaN —
no detection at all — until the next Transmission::resetBuffers (a retune). The engine restarts its sliding sums every 16
frames / 16 bins by design, so it is blind for exactly as long as the -inf row is inside the 21-frame window
(the reference is blind there too) and detects again from the first 16-frame tile whose window is clean. This file pins both
behaviours and where they part; include/specscan.h documents it. Needs an MI355X: run with -m gpu."""
import numpy as np
import pytest

import rtl_sdr_scanner_cpp_amd as pkg
from parity import check_all, floor_tolerance

pytestmark = pytest.mark.gpu


def test_all_zero_frame_reference_stays_blind_engine_recovers(ref_mod):
    n, fs, center = 2048, 512_000, 145_000_000
    nframes, z = 200, 90
    band = pkg.synth.SyntheticBand(n, seed=31, on_frame=50, off_frame=10_000)
    iq = band.frames_cf32(nframes)
    iq[z] = 0
    t = (10_000 + 100 * np.arange(nframes)).astype(np.int64)  # learning: 2 s = the first 21 frames
    ref_mod.ref().orc_set_fft_backend(0)
    ref_chain = ref_mod.RefChain(n, fs, center - fs // 2, center + fs // 2)
    r = ref_chain.process(iq, t)
    eng = pkg.SpectrumEngine(fs, center, fft_size=n, decim=1, max_batch=64)
    outs = [eng.process(iq[a:a + 50], t_ms=t[a:a + 50]) for a in range(0, nframes, 50)]
    psd = np.concatenate([o["psd"] for o in outs])
    avg = np.concatenate([o["avg"] for o in outs])
    counts = np.concatenate([np.diff(o["cand_off"]) for o in outs])
    ref_counts = np.array([len(c) for c in r["cands"]])

    # up to the zero frame: the ordinary contract
    head = {k: np.concatenate([o[k] for o in outs])[:z] for k in ("psd", "rel", "avg")}
    off = np.concatenate([[0], np.cumsum(counts[:z])]).astype(np.int32)
    idx = np.concatenate([o["cand_idx"] for o in outs])[: off[-1]]
    ref_off = np.concatenate([[0], np.cumsum(ref_counts[:z])]).astype(np.int32)
    ref_idx = np.concatenate(r["cands"][:z]).astype(np.int32)
    check_all({**head, "cand_off": off, "cand_idx": idx, "cand_avg": np.concatenate([o["cand_avg"] for o in outs])[: off[-1]]},
              {"psd": r["psd"][:z], "rel": r["rel"][:z], "avg": r["avg"][:z], "cand_off": ref_off, "cand_idx": ref_idx})
    assert ref_counts[50 + 21:z].min() > 50 and counts[50 + 21:z].min() > 50  # both were detecting the transmission

    # the zero frame itself: -inf everywhere, both
    assert np.isneginf(psd[z]).all() and np.isneginf(r["psd"][z]).all()
    # while the -inf row is inside the 21-frame window nobody detects anything
    assert counts[z:z + 21].sum() == 0 and ref_counts[z:z + 21].sum() == 0
    assert not np.isfinite(avg[z:z + 21]).any() and not np.isfinite(r["avg"][z:z + 21]).any()
    # the reference stays blind (NaN) to the end of the stream ...
    assert ref_counts[z:].sum() == 0 and np.isnan(r["avg"][z + 21:]).all()
    # ... the engine sees again from the first tile whose window is clean: at most 15 frames after the row has left
    first_clean = z + 21
    assert np.isfinite(avg[first_clean + 15:]).all()
    assert counts[first_clean + 15:].min() > 50
    # after Transmission::resetBuffers both agree again (warm-up -100 first, then detections)
    more = band.frames_cf32(60)
    t2 = (t[-1] + 100 + 100 * np.arange(60)).astype(np.int64)
    ref_chain.reset()
    eng.reset()
    r2 = ref_chain.process(more, t2)
    g2 = eng.process(more, t_ms=t2)
    off2 = np.zeros(61, np.int32)
    off2[1:] = np.cumsum([len(c) for c in r2["cands"]])
    check_all(g2, {"psd": r2["psd"], "rel": r2["rel"], "avg": r2["avg"], "cand_off": off2, "cand_idx": np.concatenate(r2["cands"]).astype(np.int32)})
    assert off2[-1] > 1000


# ---- SS_FLAG_REFERENCE_NAN: the reference's behaviour reproduced instead of repaired (csrc/reference_nan.h) --------------------
def _same_nonfinite_pattern(name, got, ref, tol_floor=2e-3):
    """NaN where the reference has NaN, -inf / +inf where it has them, the contract's tolerance on every other bin (tol_floor: a
    number, or the per-bin fp32-FFT floor of parity.floor_tolerance for the planes that hold single bins)."""
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    assert (np.isnan(got) == np.isnan(ref)).all(), f"{name}: NaN pattern differs at {np.argwhere(np.isnan(got) != np.isnan(ref))[:5].tolist()}"
    assert (np.isneginf(got) == np.isneginf(ref)).all(), f"{name}: -inf pattern differs at {np.argwhere(np.isneginf(got) != np.isneginf(ref))[:5].tolist()}"
    assert (np.isposinf(got) == np.isposinf(ref)).all(), f"{name}: +inf pattern differs"
    fin = np.isfinite(ref)
    err = np.abs(got[fin].astype(np.float64) - ref[fin].astype(np.float64))
    tol = 1e-4 * np.maximum(1.0, np.abs(ref[fin])) + (tol_floor[fin] if isinstance(tol_floor, np.ndarray) else tol_floor)
    assert (err <= tol).all(), f"{name}: worst {err.max():.3e} dB"


def _run_both(ref_mod, n, fs, iq, t, chunk, device_calls=False, **eng_kw):
    center = 145_000_000
    ref_mod.ref().orc_set_fft_backend(0)
    ref_chain = ref_mod.RefChain(n, fs, center - fs // 2, center + fs // 2)
    r = ref_chain.process(iq, t)
    eng = pkg.SpectrumEngine(fs, center, fft_size=n, decim=1, max_batch=chunk, flags=pkg.abi.SS_FLAG_REFERENCE_NAN, **eng_kw)
    if device_calls:
        import torch
        dev = torch.device("cuda", 0)
        outs = []
        for a in range(0, iq.shape[0], chunk):
            b = iq[a:a + chunk]
            d = torch.from_numpy(b.view(np.float32)).to(dev)
            o = dict(psd=torch.empty((b.shape[0], n), dtype=torch.float32, device=dev), avg=torch.empty((b.shape[0], n), dtype=torch.float32, device=dev),
                     off=torch.zeros(b.shape[0] + 1, dtype=torch.int32, device=dev), idx=torch.empty(b.shape[0] * n, dtype=torch.int32, device=dev))
            eng.process_device(d, b.shape[0], psd=o["psd"], avg=o["avg"], cand_off=o["off"], cand_idx=o["idx"])
            outs.append((d, o))  # (no synchronisation between the calls)
        eng.sync()
        res = []
        for _, o in outs:
            off = o["off"].cpu().numpy()
            res.append({"psd": o["psd"].cpu().numpy(), "avg": o["avg"].cpu().numpy(), "cand_off": off, "cand_idx": o["idx"].cpu().numpy()[:off[-1]]})
    else:
        res = [eng.process(iq[a:a + chunk], t_ms=t[a:a + chunk]) for a in range(0, iq.shape[0], chunk)]
    return r, res, ref_chain, eng


@pytest.mark.parametrize("kind", ["zero frame", "NaN sample", "zero frame during the warm-up", "zero frame, then a NaN sample"])
def test_reference_nan_flag_follows_the_reference_to_the_end_of_the_stream(ref_mod, kind):
    n, fs = 2048, 512_000
    nframes, chunk = 200, 50
    band = pkg.synth.SyntheticBand(n, seed=32, on_frame=40, off_frame=10_000)
    iq = band.frames_cf32(nframes)
    if kind.startswith("zero frame during"):
        iq[27] = 0  # learning ends with frame 20; the Averager hands out -100 until it has seen 21 frames (frame 41)
    elif kind == "NaN sample":
        iq[90, 7] = np.nan
    else:
        iq[90] = 0
        if kind.endswith("NaN sample"):
            iq[140, 1000] = np.nan
    t = (10_000 + 100 * np.arange(nframes)).astype(np.int64)  # learning: 2 s = the first 21 frames
    r, res, ref_chain, eng = _run_both(ref_mod, n, fs, iq, t, chunk)
    ref_counts = np.array([len(c) for c in r["cands"]])
    counts = np.concatenate([np.diff(o["cand_off"]) for o in res])
    assert (counts == ref_counts).all(), np.argwhere(counts != ref_counts)[:10].tolist()
    got_idx = np.concatenate([o["cand_idx"] for o in res])
    ref_idx = np.concatenate(r["cands"]).astype(np.int32) if ref_counts.sum() else np.zeros(0, np.int32)
    assert (got_idx == ref_idx).all()
    floor = floor_tolerance(np.where(np.isfinite(r["psd"]), r["psd"], np.float32(-60.0)))  # (the degenerate rows have no frame mean to measure depth against)
    for k in ("psd", "rel", "avg"):
        _same_nonfinite_pattern(k, np.concatenate([o[k] for o in res]), r[k], 2e-3 if k == "avg" else floor + 1e-3)
    first_bad = 27 if kind.startswith("zero frame during") else 90
    assert ref_counts[:first_bad].sum() > 1000 or first_bad < 60  # the reference was detecting before ...
    assert ref_counts[first_bad + 21:].sum() == 0 and np.isnan(r["avg"][max(first_bad + 21, 41):, 11:]).all()  # ... and is blind for good afterwards
    # Transmission::resetBuffers -> Averager::reset: both start afresh and agree on the ordinary contract
    more = band.frames_cf32(60)
    t2 = (t[-1] + 100 + 100 * np.arange(60)).astype(np.int64)
    ref_chain.reset()
    eng.reset()
    r2 = ref_chain.process(more, t2)
    g2s = [eng.process(more[a:a + 30], t_ms=t2[a:a + 30]) for a in (0, 30)]
    g2 = {k: np.concatenate([o[k] for o in g2s]) for k in ("psd", "rel", "avg", "cand_idx", "cand_avg")}
    g2["cand_off"] = np.concatenate([[0], np.cumsum(np.concatenate([np.diff(o["cand_off"]) for o in g2s]))]).astype(np.int32)
    off2 = np.zeros(61, np.int32)
    off2[1:] = np.cumsum([len(c) for c in r2["cands"]])
    check_all(g2, {"psd": r2["psd"], "rel": r2["rel"], "avg": r2["avg"], "cand_off": off2, "cand_idx": np.concatenate(r2["cands"]).astype(np.int32)})
    assert off2[-1] > 1000


@pytest.mark.parametrize("n,fs,chunk,nframes,z", [(8192, 2_048_000, 64, 256, 150), (65536, 20_000_000, 32, 128, 70)])
def test_reference_nan_flag_on_the_device_path(ref_mod, n, fs, chunk, nframes, z):
    """The same through ss_process_device (the step kernel's launches, stages in order on the chain's stream, nothing synchronised
    between the calls): tile culling included at 8192 points."""
    band = pkg.synth.SyntheticBand(n, seed=33, on_frame=40, off_frame=10_000)
    iq = band.frames_cf32(nframes)
    iq[z] = 0
    t = (10_000 + 100 * np.arange(nframes)).astype(np.int64)  # the reference's wall clock ends learning with frame 20; ss_process_device counts frames
    r, res, _, eng = _run_both(ref_mod, n, fs, iq, t, chunk, device_calls=True, learn_frames=21)
    ref_counts = np.array([len(c) for c in r["cands"]])
    counts = np.concatenate([np.diff(o["cand_off"]) for o in res])
    assert (counts == ref_counts).all(), np.argwhere(counts != ref_counts)[:10].tolist()
    assert (np.concatenate([o["cand_idx"] for o in res]) == np.concatenate(r["cands"]).astype(np.int32)).all()
    _same_nonfinite_pattern("avg", np.concatenate([o["avg"] for o in res]), r["avg"])
    assert ref_counts[:z].sum() > 1000 and ref_counts[z:].sum() == 0
