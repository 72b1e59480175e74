"""Host-side signal tracker (host/signal_tracker.cpp) against the reference's own Transmission / Signal code
(oracle/_ref, compiled in place from /root/reference/sources): the list handed to Notification::notify
(transmission.cpp:67) — tuned frequency shifts and flush flags, strongest first — and the tracked signal keys
must be identical, frame by frame. Integer work: bit-exact. Runs on the CPU: the tracker takes planes and
candidates from whoever produced them (here the oracle, whose planes equal the reference's bit for bit)."""
import glob
import os

import numpy as np
import pytest

import rtl_sdr_scanner_cpp_amd as pkg

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "ref_tracker_*.npz")))


def _scenario(n, fs, seed, nframes, on, off, dt):
    band = pkg.synth.SyntheticBand(n, seed=seed, on_frame=on, off_frame=off, comb_width=max(8, n // 32))
    iq = band.frames_cf32(nframes)
    t = (1_000 + dt * np.arange(nframes)).astype(np.int64)
    return iq, t


def _track(O, iq, t, n, fs, center, **tk):
    ch = O.oracle_chain(fs, center, fft_size=n, decim=1, max_batch=iq.shape[0])
    r = ch.process(iq, t_ms=t)
    tr = pkg.tracker.SignalTracker(n, fs, **tk)
    return tr.process_batch(t, r["avg"], r["rel"], r["cand_off"], r["cand_idx"])


@pytest.mark.parametrize("n,fs,seed,min_ms,timeout_ms", [(256, 64_000, 31, 2000, 2000), (1024, 256_000, 32, 400, 600), (512, 128_000, 33, 0, 40)])
def test_tracker_matches_reference_live(ref_mod, n, fs, seed, min_ms, timeout_ms):
    O = ref_mod
    center, nframes, dt = 145_000_000, 330, 40
    iq, t = _scenario(n, fs, seed, nframes, on=70, off=190, dt=dt)
    O.ref().orc_set_fft_backend(0)
    O.lib().orc_set_fft_backend(0)
    ref = O.RefChain(n, fs, center - fs // 2, center + fs // 2, min_time_ms=min_ms, timeout_ms=timeout_ms)
    want = ref.process(iq, t)
    got = _track(O, iq, t, n, fs, center, min_time_ms=min_ms, timeout_ms=timeout_ms)
    seen_tx = seen_flush = 0
    for f in range(nframes):
        tx, sig = got[f]
        np.testing.assert_array_equal(tx, want["tx"][f], err_msg=f"frame {f}")
        np.testing.assert_array_equal(sig, want["signals"][f], err_msg=f"frame {f}")
        seen_tx += len(tx)
        seen_flush += int(tx[:, 1].sum()) if len(tx) else 0
    assert seen_tx > 100 and len(want["tx"][-1]) == 0  # transmissions appeared and timed out again
    if min_ms > 0:
        assert seen_flush > 0


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_tracker_matches_golden(oracle_mod, path):
    g = np.load(path)
    n, fs, center = int(g["n"]), int(g["fs"]), int(g["center"])
    oracle_mod.lib().orc_set_fft_backend(0)
    got = _track(oracle_mod, g["iq"], g["t_ms"], n, fs, center, min_time_ms=int(g["min_ms"]), timeout_ms=int(g["timeout_ms"]))
    tx_off, tx, sig_off, sig = g["tx_off"], g["tx"], g["sig_off"], g["sig"]
    for f in range(len(got)):
        np.testing.assert_array_equal(got[f][0], tx[tx_off[f]:tx_off[f + 1]], err_msg=f"frame {f}")
        np.testing.assert_array_equal(got[f][1], sig[sig_off[f]:sig_off[f + 1]], err_msg=f"frame {f}")
    assert tx_off[-1] > 50


def test_reset_clears_signals_and_ring(oracle_mod):
    n, fs = 256, 64_000
    tr = pkg.tracker.SignalTracker(n, fs)
    avg = np.full(n, 20.0, np.float32)
    rel = np.full(n, 20.0, np.float32)
    tx, sig = tr.process_frame(1000, avg, rel, np.arange(100, 110, dtype=np.int32))
    assert len(sig) >= 1
    tr.reset()
    tx, sig = tr.process_frame(1040, np.full(n, -100.0, np.float32), np.full(n, -100.0, np.float32), np.zeros(0, np.int32))
    assert len(sig) == 0 and len(tx) == 0
