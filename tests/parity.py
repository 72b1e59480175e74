"""Shared parity checks: HIP engine vs CPU oracle (or vs golden planes made by the reference's own code).

Contract (BASELINE.json north_star):
  * per-bin dB planes:  |got - ref| <= 1e-4 * max(1, |ref|);  -100 sentinels and -inf must match exactly;
  * candidate indices:  identical per frame, except bins whose reference avg lies within BAND dB of
    start_level — there the fp32 rounding of two different FFTs decides, and they are counted and
    reported, not compared.
"""
import numpy as np

TOL = 1e-4
BAND = 1e-3  # dB


def check_plane(name, got, ref):
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    exact = got == ref  # covers -100 and +-inf
    special = ~np.isfinite(ref) | (ref == -100.0)
    assert exact[special].all(), f"{name}: sentinel / non-finite bins differ"
    err = np.abs(got - ref)
    tol = TOL * np.maximum(1.0, np.abs(ref))
    bad = ~exact & ~(err <= tol)
    assert not bad.any(), f"{name}: {int(bad.sum())} bins outside tolerance, worst {np.nanmax(np.where(bad, err, 0)):.3e} dB"
    fin = np.isfinite(ref) & (ref != -100.0)
    return float(err[fin].max()) if fin.any() else 0.0


def cand_set(off, idx):
    frames = np.repeat(np.arange(len(off) - 1), np.diff(off))
    return set(zip(frames.tolist(), np.asarray(idx).tolist()))


def check_candidates(got_off, got_idx, ref_off, ref_idx, ref_avg, start_level, got_avg_plane=None):
    """Returns (n_reference_candidates, n_dont_care)."""
    a, b = cand_set(got_off, got_idx), cand_set(ref_off, ref_idx)
    near = np.abs(ref_avg - np.float32(start_level)) < BAND
    diff = [(f, i) for (f, i) in a ^ b if not near[f, i]]
    assert not diff, f"candidate sets differ outside the {BAND} dB band: {sorted(diff)[:10]}"
    # inside each frame the engine lists candidates in ascending bin order
    for f in range(len(got_off) - 1):
        seg = np.asarray(got_idx[got_off[f]:got_off[f + 1]])
        assert (np.diff(seg) > 0).all()
    return len(b), len(a ^ b)


def check_all(got, ref, start_level=8.0):
    errs = {k: check_plane(k, got[k], ref[k]) for k in ("psd", "rel", "avg") if k in got and k in ref}
    ncand, ndc = check_candidates(got["cand_off"], got["cand_idx"], ref["cand_off"], ref["cand_idx"], ref["avg"], start_level)
    # cand_avg = avg plane at the candidates
    frames = np.repeat(np.arange(len(got["cand_off"]) - 1), np.diff(got["cand_off"]))
    if "avg" in got and len(frames):
        np.testing.assert_array_equal(got["cand_avg"], got["avg"][frames, got["cand_idx"]])
    return errs, ncand, ndc
