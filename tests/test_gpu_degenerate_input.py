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
from parity import check_all

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
