"""Generate tests/golden/*.npz from oracle/_ref — the reference's own PSD / NoiseLearner / Transmission /
Averager / average() code compiled in place from /root/reference/sources (oracle/Makefile), fed through
the restated fft_v (built-in fp32 FFT back end). Run in the build container:

    python tests/golden/make_golden.py

The fixtures are small (N = 256, and one at the headline N = 8192 that keeps int8 IQ, full candidate lists and every 8th
bin of the planes) so they can be committed; they pin the C oracle on machines where /root/reference — and therefore
oracle/_ref — does not exist."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
import rtl_sdr_scanner_cpp_amd as pkg  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def chain_case(name, n, fs, seed, nframes, dt_ms, ignored=(), retune_at=None):
    center = 145_000_000
    lo, hi = center - fs // 2, center + fs // 2
    band = pkg.synth.SyntheticBand(n, seed=seed, on_frame=60, off_frame=110, comb_width=12)
    iq = band.frames_cf32(nframes)
    t = (1_000_000 + dt_ms * np.arange(nframes)).astype(np.int64)
    O.ref().orc_set_fft_backend(0)
    ref = O.RefChain(n, fs, lo, hi, ignored=ignored)
    if retune_at is None:
        r = ref.process(iq, t)
        segs = [r]
    else:  # retune away and back: new centre learns its own noise, the averager is reset (sdr_device.cpp:54-80)
        a = ref.process(iq[:retune_at], t[:retune_at])
        ref.set_range(lo + fs, hi + fs)
        ref.reset()
        b = ref.process(iq[retune_at:], t[retune_at:])
        segs = [a, b]
    psd = np.concatenate([s["psd"] for s in segs])
    rel = np.concatenate([s["rel"] for s in segs])
    avg = np.concatenate([s["avg"] for s in segs])
    cands = [c for s in segs for c in s["cands"]]
    off = np.zeros(nframes + 1, np.int32)
    off[1:] = np.cumsum([len(c) for c in cands])
    idx = np.concatenate(cands).astype(np.int32) if off[-1] else np.zeros(0, np.int32)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), iq=iq, t_ms=t, psd=psd, rel=rel, avg=avg, cand_off=off,
                        cand_idx=idx, n=n, fs=fs, center=center, ignored=np.array(ignored, np.int32).reshape(-1),
                        retune_at=-1 if retune_at is None else retune_at)
    print(name, "frames", nframes, "candidates", int(off[-1]))


def chain_case_cs8(name, n, fs, seed, nframes, dt_ms, on_frame, sub=8):
    """A fixture at the headline size (N = 8192) that stays committable: int8 IQ (HackRF-shaped, converted exactly like the
    engine's load stage: int8 / 128), the reference's candidate lists in full, its PSD / rel / avg planes at every `sub`-th bin,
    and the learned noise ceiling."""
    center = 145_000_000
    band = pkg.synth.SyntheticBand(n, seed=seed, on_frame=on_frame, off_frame=nframes - 6)
    iq8 = band.frames_cs8(nframes)
    iq = (iq8[..., 0].astype(np.float32) / np.float32(128.0) + 1j * (iq8[..., 1].astype(np.float32) / np.float32(128.0))).astype(np.complex64)
    t = (1_000_000 + dt_ms * np.arange(nframes)).astype(np.int64)
    O.ref().orc_set_fft_backend(0)
    ref = O.RefChain(n, fs, center - fs // 2, center + fs // 2)
    r = ref.process(iq, t)
    off = np.zeros(nframes + 1, np.int32)
    off[1:] = np.cumsum([len(c) for c in r["cands"]])
    idx = np.concatenate(r["cands"]).astype(np.int32)
    thr, ready = ref.noise()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), iq8=iq8, t_ms=t, psd_sub=r["psd"][:, ::sub], rel_sub=r["rel"][:, ::sub],
                        avg_sub=r["avg"][:, ::sub], cand_off=off, cand_idx=idx, cand_avg=r["avg"][np.repeat(np.arange(nframes), np.diff(off)), idx],
                        thr=thr, n=n, fs=fs, center=center, sub=sub)
    print(name, "frames", nframes, "candidates", int(off[-1]), "noise ready", ready)


def tracker_case(name, n, fs, seed, nframes, dt_ms, min_ms, timeout_ms):
    """Per-frame output of the reference's Transmission/Signal bookkeeping: the notified (shift Hz, flush) list and
    the keys of m_signals."""
    center = 145_000_000
    band = pkg.synth.SyntheticBand(n, seed=seed, on_frame=70, off_frame=190, comb_width=max(8, n // 32))
    iq = band.frames_cf32(nframes)
    t = (1_000 + dt_ms * np.arange(nframes)).astype(np.int64)
    O.ref().orc_set_fft_backend(0)
    ref = O.RefChain(n, fs, center - fs // 2, center + fs // 2, min_time_ms=min_ms, timeout_ms=timeout_ms)
    r = ref.process(iq, t)
    tx_off = np.zeros(nframes + 1, np.int32)
    tx_off[1:] = np.cumsum([len(x) for x in r["tx"]])
    sig_off = np.zeros(nframes + 1, np.int32)
    sig_off[1:] = np.cumsum([len(x) for x in r["signals"]])
    tx = np.concatenate([x.reshape(-1, 2) for x in r["tx"]]).astype(np.int32)
    sig = np.concatenate(r["signals"]).astype(np.int32)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), iq=iq, t_ms=t, tx_off=tx_off, tx=tx, sig_off=sig_off, sig=sig, n=n, fs=fs,
                        center=center, min_ms=min_ms, timeout_ms=timeout_ms)
    print(name, "frames", nframes, "transmission entries", int(tx_off[-1]))


if __name__ == "__main__":
    if not O.have_ref():
        sys.exit("oracle/_ref is not built (needs /root/reference): run make -C oracle")
    # fs/N = 250 Hz bins like the reference's regime; 2 s of learning at dt = 40 ms -> 51 frames
    chain_case("ref_chain_n256", 256, 64000, seed=3, nframes=140, dt_ms=40)
    chain_case("ref_chain_n256_ignored", 256, 64000, seed=4, nframes=140, dt_ms=40,
               ignored=[145_000_000 + 9000, 145_000_000 + 13000])
    chain_case("ref_chain_n256_retune", 256, 64000, seed=5, nframes=220, dt_ms=40, retune_at=100)
    tracker_case("ref_tracker_n256", 256, 64000, seed=6, nframes=300, dt_ms=40, min_ms=800, timeout_ms=1200)
    # the headline size: 2.048 MS/s, N = 8192; dt = 100 ms -> learning ends after 21 frames, the averager is full 20 frames on
    chain_case_cs8("ref_big_n8192_cs8", 8192, 2_048_000, seed=7, nframes=72, dt_ms=100, on_frame=46)
