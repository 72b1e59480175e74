"""Pin the C oracle (oracle/specscan_oracle.c) on the whole chain:
  * against committed golden vectors produced by the reference's own compiled sources
    (tests/golden/make_golden.py -> oracle/_ref), bit for bit;
  * live against oracle/_ref where it is built, on further seeds / sizes, bit for bit.
Both sides use the same restated fft_v (the FFT itself lives in GNU Radio/FFTW, un-vendored), so every
difference would come from the reference-owned stages: PSD, NoiseLearner, Averager, average(),
candidate predicate."""
import glob
import os

import numpy as np
import pytest

import rtl_sdr_scanner_cpp_amd as pkg

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "ref_chain_*.npz")))


def _run_oracle(O, iq, t, n, fs, center, ignored, retune_at, chunk):
    ch = O.oracle_chain(fs, center, fft_size=n, decim=1, max_batch=64, ignored=ignored)
    O.lib().orc_set_fft_backend(0)
    outs = []
    lo, hi = center - fs // 2, center + fs // 2
    pos = 0
    nframes = iq.shape[0]
    while pos < nframes:
        if retune_at is not None and pos == retune_at:
            ch.set_frequency_range(lo + fs, hi + fs)
            ch.reset()
        end = min(nframes, pos + chunk)
        if retune_at is not None and pos < retune_at:
            end = min(end, retune_at)
        outs.append(ch.process(iq[pos:end], t_ms=t[pos:end]))
        pos = end
    psd = np.concatenate([o["psd"] for o in outs])
    rel = np.concatenate([o["rel"] for o in outs])
    avg = np.concatenate([o["avg"] for o in outs])
    idx = np.concatenate([o["cand_idx"] for o in outs])
    counts = np.concatenate([np.diff(o["cand_off"]) for o in outs])
    off = np.zeros(nframes + 1, np.int32)
    off[1:] = np.cumsum(counts)
    # cand_avg is avg at the candidate
    cav = np.concatenate([o["cand_avg"] for o in outs])
    return psd, rel, avg, off, idx, cav


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
@pytest.mark.parametrize("chunk", [1, 7, 64])
def test_oracle_matches_golden(oracle_mod, path, chunk):
    g = np.load(path)
    n, fs, center = int(g["n"]), int(g["fs"]), int(g["center"])
    retune_at = int(g["retune_at"])
    psd, rel, avg, off, idx, cav = _run_oracle(oracle_mod, g["iq"], g["t_ms"], n, fs, center, g["ignored"],
                                               None if retune_at < 0 else retune_at, chunk)
    np.testing.assert_array_equal(psd, g["psd"])
    np.testing.assert_array_equal(rel, g["rel"])
    np.testing.assert_array_equal(avg, g["avg"])
    np.testing.assert_array_equal(off, g["cand_off"])
    np.testing.assert_array_equal(idx, g["cand_idx"])
    frames = np.repeat(np.arange(len(off) - 1), np.diff(off))
    np.testing.assert_array_equal(cav, g["avg"][frames, idx])
    assert off[-1] > 0 and (g["rel"] == -100).any() and (g["avg"] == -100).any()  # the fixture exercises all phases


@pytest.mark.parametrize("n,fs,seed", [(64, 16000, 11), (512, 128000, 12), (2048, 2048000 // 4, 13)])
def test_oracle_matches_live_reference(ref_mod, n, fs, seed):
    O = ref_mod
    center = 433_000_000
    nframes = 150
    band = pkg.synth.SyntheticBand(n, seed=seed, on_frame=70, off_frame=130, comb_width=16)
    iq = band.frames_cf32(nframes)
    t = (5_000 + 40 * np.arange(nframes)).astype(np.int64)
    O.ref().orc_set_fft_backend(0)
    ref = O.RefChain(n, fs, center - fs // 2, center + fs // 2)
    r = ref.process(iq, t)
    psd, rel, avg, off, idx, _ = _run_oracle(O, iq, t, n, fs, center, (), None, 32)
    np.testing.assert_array_equal(psd, r["psd"])
    np.testing.assert_array_equal(rel, r["rel"])
    np.testing.assert_array_equal(avg, r["avg"])
    for f in range(nframes):
        np.testing.assert_array_equal(idx[off[f]:off[f + 1]], r["cands"][f])


@pytest.mark.parametrize("seed", range(int(os.environ.get("SS_ORACLE_SEEDS", "8"))))
def test_oracle_matches_live_reference_on_random_runs(ref_mod, seed):
    """Random size, frame period, ignored ranges, call sizes and a retune with reset at a random frame: the restated chain
    against the reference's own compiled PSD / NoiseLearner / Transmission, bit for bit, frame by frame."""
    O = ref_mod
    rng = np.random.default_rng(300 + seed)
    n = int(rng.choice([64, 128, 256, 512, 1024]))
    fs = n * int(rng.choice([125, 250, 500]))
    center = 433_000_000
    nframes = int(rng.integers(140, 260))
    dt = int(rng.choice([15, 40, 70]))
    band = pkg.synth.SyntheticBand(n, seed=700 + seed, on_frame=int(2000 / dt) + 12, off_frame=nframes - 5, comb_width=max(8, n // 24))
    iq = band.frames_cf32(nframes)
    t = (9_000 + dt * np.arange(nframes)).astype(np.int64)
    ignored = ()
    if rng.random() < 0.6:
        lo = center + int(rng.integers(-fs // 3, fs // 4))
        ignored = (lo, lo + fs // 16)
    retune_at = int(rng.integers(nframes // 2, nframes - 40)) if rng.random() < 0.5 else None
    chunk = int(rng.choice([1, 5, 33, 64]))
    O.ref().orc_set_fft_backend(0)
    ref = O.RefChain(n, fs, center - fs // 2, center + fs // 2, ignored=ignored)
    parts = []
    if retune_at is None:
        parts.append(ref.process(iq, t))
    else:
        parts.append(ref.process(iq[:retune_at], t[:retune_at]))
        ref.set_range(center + fs - fs // 2, center + fs + fs // 2)  # SdrDevice::setFrequencyRange + Transmission::resetBuffers
        ref.reset()
        parts.append(ref.process(iq[retune_at:], t[retune_at:]))
    r = {k: np.concatenate([p_[k] for p_ in parts]) for k in ("psd", "rel", "avg")}
    cands = [c for p_ in parts for c in p_["cands"]]
    psd, rel, avg, off, idx, _ = _run_oracle(O, iq, t, n, fs, center, ignored, retune_at, chunk)
    np.testing.assert_array_equal(psd, r["psd"])
    np.testing.assert_array_equal(rel, r["rel"])
    np.testing.assert_array_equal(avg, r["avg"])
    for f in range(nframes):
        np.testing.assert_array_equal(idx[off[f]:off[f + 1]], cands[f], err_msg=f"seed {seed} frame {f}")


def test_frame_count_learning_equals_timestamp_learning(oracle_mod):
    """learn_frames = k is the same as timestamps that complete NOISE_LEARNING_TIME on frame k."""
    O = oracle_mod
    n, fs, center = 128, 32000, 100_000_000
    band = pkg.synth.SyntheticBand(n, seed=21, on_frame=40, off_frame=70, comb_width=8)
    iq = band.frames_cf32(90)
    a = O.oracle_chain(fs, center, fft_size=n, decim=1, max_batch=128, learn_frames=26)
    b = O.oracle_chain(fs, center, fft_size=n, decim=1, max_batch=128, learn_ms=2000)
    t = (80 * np.arange(90)).astype(np.int64)  # 2000 ms are over on frame index 25 -> 26 frames learned
    ra, rb = a.process(iq), b.process(iq, t_ms=t)
    for k in ("psd", "rel", "avg", "cand_idx", "cand_off"):
        np.testing.assert_array_equal(ra[k], rb[k])
    assert (ra["rel"][25] == -100).all() and (ra["rel"][26] != -100).all()


def test_decimation_keeps_first_n_of_each_item(oracle_mod):
    """Decimator::decimate (decimator.h:15-22)."""
    O = oracle_mod
    n, fs, center, d = 128, 32000, 100_000_000, 3
    band = pkg.synth.SyntheticBand(n, decim=d, seed=22, on_frame=10, off_frame=20, comb_width=8)
    iq = band.frames_cf32(12)
    a = O.oracle_chain(fs, center, fft_size=n, decim=d, max_batch=16, learn_frames=4)
    b = O.oracle_chain(fs, center, fft_size=n, decim=1, max_batch=16, learn_frames=4)
    ra, rb = a.process(iq), b.process(np.ascontiguousarray(iq[:, :n]))
    np.testing.assert_array_equal(ra["psd"], rb["psd"])


def test_empty_batch_and_overflow(oracle_mod):
    O = oracle_mod
    n, fs, center = 128, 32000, 100_000_000
    ch = O.oracle_chain(fs, center, fft_size=n, decim=1, max_batch=64, learn_frames=5)
    r = ch.process(np.zeros((0, n), np.complex64))
    assert r["cand_off"].tolist() == [0] and r["psd"].shape == (0, n)
    band = pkg.synth.SyntheticBand(n, seed=23, on_frame=5, off_frame=60, comb_width=8)
    iq = band.frames_cf32(60)
    full = O.oracle_chain(fs, center, fft_size=n, decim=1, max_batch=64, learn_frames=5).process(iq)
    assert full["cand_off"][-1] > 10
    small = ch.process(iq, cand_cap=10)
    assert small["status"] == pkg.abi.SS_ERR_CAND_OVERFLOW
    np.testing.assert_array_equal(small["cand_off"], full["cand_off"])  # offsets stay exact
    np.testing.assert_array_equal(small["cand_idx"], full["cand_idx"][:10])
    with pytest.raises(pkg.abi.SpecscanError):
        ch.process(np.zeros((65, n), np.complex64))


def test_oracle_matches_the_headline_size_golden(oracle_mod):
    """N = 8192 at 2.048 MS/s on int8 IQ (tests/golden/ref_big_n8192_cs8.npz: the reference's candidate lists in full, its
    planes at every 8th bin, its noise ceiling), wall-clock learning included."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_big_n8192_cs8.npz"))
    n, fs, center, sub = int(g["n"]), int(g["fs"]), int(g["center"]), int(g["sub"])
    oracle_mod.lib().orc_set_fft_backend(0)
    ch = oracle_mod.oracle_chain(fs, center, fft_size=n, decim=1, in_format=pkg.abi.SS_FMT_CS8, max_batch=32)
    iq8, t = g["iq8"], g["t_ms"]
    outs = [ch.process(iq8[a:a + 32], t_ms=t[a:a + 32]) for a in range(0, iq8.shape[0], 32)]
    for k in ("psd", "rel", "avg"):
        np.testing.assert_array_equal(np.concatenate([o[k] for o in outs])[:, ::sub], g[k + "_sub"])
    counts = np.concatenate([np.diff(o["cand_off"]) for o in outs])
    np.testing.assert_array_equal(np.concatenate([[0], np.cumsum(counts)]), g["cand_off"])
    np.testing.assert_array_equal(np.concatenate([o["cand_idx"] for o in outs]), g["cand_idx"])
    np.testing.assert_array_equal(np.concatenate([o["cand_avg"] for o in outs]), g["cand_avg"])
    np.testing.assert_array_equal(ch.read_noise()[0], g["thr"])
    assert g["cand_off"][-1] > 1000
