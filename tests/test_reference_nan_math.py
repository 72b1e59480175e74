"""SS_FLAG_REFERENCE_NAN, the arithmetic behind it, checked on the CPU against the reference's own code (oracle/_ref): the rule of
csrc/reference_nan.h — from the first NaN / -inf / +inf bin of every frame alone, bins >= bad_from(f) of the reference's avg row
are NaN and the bins below are not — for spectra with exact zeros, NaNs and overflowing values at random bins of random frames
(what real input cannot place at single bins: a zero frame is zero everywhere), through retunes' resets and the Averager's warm-up.
`nan_plan` below restates k_nan_plan line by line."""
import numpy as np
import pytest


def nan_plan(psd, n_learn_before, pushed_before, state, n):
    """k_nonfinite_scan + k_nan_plan of csrc/reference_nan.h for one batch: psd [nframes, n] -> bad_from [nframes]."""
    bad_from = np.full(psd.shape[0], n, np.int64)
    for f in range(psd.shape[0]):
        row = psd[f]
        first = [n, n, n]
        if f >= n_learn_before:
            for k, m in enumerate((np.isnan(row), np.isneginf(row), np.isposinf(row))):
                w = np.flatnonzero(m)
                if w.size:
                    first[k] = int(w[0])
        pos = state["pos"]
        state["cum_old"] = min(state["cum_old"], min(state["ring"][pos]))
        state["ring"][pos] = [first[1], first[2]]
        state["pos"] = 0 if pos == 20 else pos + 1
        state["cum_nan"] = min(state["cum_nan"], first[0])
        wm = min(r[0] for r in state["ring"])
        wp = min(r[1] for r in state["ring"])
        first_inf, other = min(wm, wp), max(wm, wp)
        first_nan = min(state["cum_nan"], state["cum_old"])
        bad = n
        if first_nan < n:
            bad = min(bad, first_nan - 10)
        if first_inf < n:
            bad = min(bad, first_inf + 11)
        if other < n and other <= first_inf + 20:
            bad = min(bad, other - 10)
        if pushed_before + f + 1 < 21:
            bad = n
        bad_from[f] = max(bad, 0)
    return bad_from


def fresh_state(n):
    return {"cum_nan": n, "cum_old": n, "ring": [[n, n] for _ in range(21)], "pos": 0}


@pytest.mark.parametrize("seed", range(24))
def test_first_bins_alone_give_the_reference_nan_region(ref_mod, seed):
    rng = np.random.default_rng(seed)
    n, fs, center = 512, 128_000, 145_000_000
    nframes = 140
    spec = (rng.standard_normal((nframes, n)) + 1j * rng.standard_normal((nframes, n))).astype(np.complex64) * 30
    spec[60:, 200:230] *= 40  # a transmission, so that candidates exist to be lost
    kinds = rng.choice(["zero", "nan", "huge", "zero+nan", "two zeros", "zero+huge"])
    f1 = int(rng.integers(5, 100))
    b1 = int(rng.integers(0, n))
    if kinds in ("zero", "zero+nan", "two zeros", "zero+huge"):
        spec[f1, b1] = 0
    if kinds == "nan":
        spec[f1, b1] = np.nan
    if kinds == "huge":
        spec[f1, b1] = 3e19 + 0j
    if kinds == "zero+nan":
        spec[min(f1 + int(rng.integers(1, 40)), nframes - 1), int(rng.integers(0, n))] = np.nan
    if kinds == "two zeros":
        spec[min(f1 + int(rng.integers(0, 30)), nframes - 1), int(rng.integers(0, n))] = 0
    if kinds == "zero+huge":
        spec[min(f1 + int(rng.integers(0, 30)), nframes - 1), min(n - 1, b1 + int(rng.integers(1, 60)))] = 3e19 + 0j
    t = (10_000 + 1_100 * np.arange(nframes)).astype(np.int64)  # learning ends with frame 2
    ref_mod.ref().orc_set_fft_backend(0)
    chain = ref_mod.RefChain(n, fs, center - fs // 2, center + fs // 2)
    r = chain.process(spec, t, spectrum=True)
    state = fresh_state(n)
    bad = nan_plan(r["psd"], 3, 0, state, n)
    isn = np.isnan(r["avg"])
    for f in range(nframes):
        want = np.arange(n) >= bad[f]
        assert (isn[f] == want).all(), (kinds, f, f1, b1, int(bad[f]), np.flatnonzero(isn[f] != want)[:6].tolist())
        assert all(c < bad[f] for c in r["cands"][f])
    assert isn.any() or f1 + 21 >= nframes or kinds == "huge"
    # a retune: Transmission::resetBuffers -> Averager::reset; the state of the rule starts afresh with it
    chain.reset()
    more = (rng.standard_normal((40, n)) + 1j * rng.standard_normal((40, n))).astype(np.complex64) * 30
    more[25, 100] = 0
    r2 = chain.process(more, t[-1] + 1_100 * (1 + np.arange(40)), spectrum=True)
    bad2 = nan_plan(r2["psd"], 0, 0, fresh_state(n), n)
    isn2 = np.isnan(r2["avg"])
    for f in range(40):
        assert (isn2[f] == (np.arange(n) >= bad2[f])).all(), (f, int(bad2[f]))
