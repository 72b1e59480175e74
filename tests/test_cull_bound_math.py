"""The inequality behind the tile culling (csrc/detect_fused.h), checked on the CPU against the oracle's own planes — no GPU:

    every rel value of frame g within reach of a tile (its 256 bins and 32 more on either side) is <= M_g - m, with M_g the
    largest dB value of the frame over those bins and m the smallest noise ceiling over them; so the 21 x 21 mean that ends
    at frame f is <= mean(M_{f-20} .. M_f) - m, and a tile whose bound stays below start_level - 0.0625 dB for each of its
    16 frames holds no candidate.

The oracle is the reference's arithmetic (running sums included; tests/test_oracle_vs_reference.py), so this pins the bound
against what the reference would have found — for signals from well below to well above the threshold, int8 input, and with a
fair share of the tiles really dropped (the check is not vacuous). The stream here is one sequence of frames: where a caller's batches
begin makes no difference to the bound — which is why the tiles at a batch's start may be tested with the maxima of the frames before it
(8192 points, deep pipelining: the halo frames the launch transforms once more leave theirs, round 5). The GPU tests (tests/test_gpu_cull.py) then show that the
kernels implement this decision: culled == unculled, list by list."""
import numpy as np
import pytest

import rtl_sdr_scanner_cpp_amd as pkg

G, TF, TB, MARGIN = 21, 16, 256, 0.0625


def _bound_check(oracle_mod, n, fs, nframes, learn, rel_db, seed, fmt="cf32", centres=None):
    kw = dict(on_frame=learn + 30, off_frame=nframes - 40, rel_db=rel_db)
    if centres:
        kw["centres"] = centres
    band = pkg.synth.SyntheticBand(n, seed=seed, **kw)
    iq = band.frames_cf32(nframes) if fmt == "cf32" else band.frames_cs8(nframes)
    in_format = pkg.abi.SS_FMT_CF32 if fmt == "cf32" else pkg.abi.SS_FMT_CS8
    ch = oracle_mod.oracle_chain(fs, 145_000_000, fft_size=n, decim=1, in_format=in_format, learn_frames=learn, max_batch=nframes)
    r = ch.process(iq)
    psd = r["psd"].astype(np.float64)
    thr = ch.read_noise()[0].astype(np.float64)
    hits = np.zeros((nframes, n), bool)
    frames = np.repeat(np.arange(nframes), np.diff(r["cand_off"]))
    hits[frames, r["cand_idx"]] = True
    cols = n // TB
    pad_p = np.pad(psd, ((0, 0), (32, 32)), constant_values=-np.inf)
    pad_t = np.pad(thr, (32, 32), constant_values=np.inf)
    culled = evaluated = 0
    for bt in range(cols):
        m_col = pad_p[:, TB * bt:TB * bt + TB + 64].max(axis=1)   # M_g over bins [256 bt - 32, 256 bt + 288)
        tm = pad_t[TB * bt:TB * bt + TB + 64].min()
        for f0 in range(0, nframes - TF + 1, TF):
            if f0 - (G - 1) < learn:  # rows that are learning frames (or before the stream): never tested
                continue
            sums = np.array([m_col[f - (G - 1):f + 1].sum() for f in range(f0, f0 + TF)])
            bound = sums.max() / G - tm
            if bound < 8.0 - MARGIN:
                culled += 1
                assert not hits[f0:f0 + TF, TB * bt:TB * bt + TB].any(), (n, bt, f0, bound)
            else:
                evaluated += 1
    return culled, evaluated, int(hits.sum())


@pytest.mark.parametrize("rel_db", [14.0, 17.0, 18.5, 20.0, 25.0])
def test_a_tile_below_the_bound_holds_no_candidate(oracle_mod, rel_db):
    culled, evaluated, ncand = _bound_check(oracle_mod, 4096, 2_048_000, 400, 100, rel_db, seed=int(rel_db * 10), centres=(0.05, -0.31, 0.43))
    assert culled > 0 and evaluated > 0, (culled, evaluated)
    if rel_db >= 20.0:
        assert ncand > 500, ncand
    assert culled > evaluated // 4, (culled, evaluated)  # a fair share of the tiles is really dropped


def test_the_bound_on_int8_input_and_a_short_learning_phase(oracle_mod):
    # a ceiling learnt over 16 frames sits low: the bound drops fewer tiles but never one that holds a candidate
    c16, e16, _ = _bound_check(oracle_mod, 4096, 2_048_000, 300, 16, 25.0, seed=3, fmt="cs8", centres=(0.05, -0.31))
    c100, e100, ncand = _bound_check(oracle_mod, 4096, 2_048_000, 300, 100, 25.0, seed=3, fmt="cs8", centres=(0.05, -0.31))
    assert ncand > 200 and c100 > 0 and c16 + e16 > 0
    assert c100 / (c100 + e100) >= c16 / (c16 + e16) - 0.05, (c16, e16, c100, e100)


def test_the_drift_allowance_at_chosen_bins_is_the_planes_allowance_there():
    """tests/parity.py: running_sum_drift_at (what cand_avg is held to at the candidates of a 65536-point frame and longer) is
    running_sum_drift's value at those (frame, bin) pairs, sentinel frames included."""
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from parity import running_sum_drift, running_sum_drift_at
    rng = np.random.default_rng(5)
    n, frames = 4096, 12
    avg = rng.normal(0.0, 6.0, (frames, n)).astype(np.float32)
    avg[3] = -100.0  # a frame that still holds sentinels: its running sum is an order of magnitude larger
    avg[7, 100:200] = np.inf
    plane = running_sum_drift(n, avg)
    f = rng.integers(0, frames, 500)
    b = rng.integers(0, n, 500)
    np.testing.assert_allclose(running_sum_drift_at(n, avg, f, b), plane[f, b], rtol=1e-12)
    assert plane[3].max() > 5 * plane[0].max()
