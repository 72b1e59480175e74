"""BASELINE.json full sizes (config 2: 1024 frames x 8192 points on one MI355X) through
size-independent properties: the oracle would need minutes here, the properties do not need it.
Run with -m gpu."""
import os

import numpy as np
import pytest

import rtl_sdr_scanner_cpp_amd as pkg

pytestmark = pytest.mark.gpu

N, FS, CENTER, B = 8192, 2_048_000, 145_000_000, 1024


@pytest.fixture(scope="module")
def batch():
    band = pkg.synth.SyntheticBand(N, seed=31, on_frame=150, off_frame=900)
    return band.frames_cf32(B)


def test_batch_split_invariance(batch):
    """One 1024-frame call == four 256-frame calls == ragged calls, bit for bit: the state carried across
    work() calls (noise ceiling, averager ring, frame counter) is exactly what a bigger batch sees."""
    kw = dict(fft_size=N, decim=1, learn_frames=100, max_batch=B)
    whole = pkg.SpectrumEngine(FS, CENTER, **kw).process(batch)
    for sizes in ([256] * 4, [1, 99, 100, 7, 13, 804], [3, 5, 2, 30, 40, 17, 36, 35, 856]):
        eng = pkg.SpectrumEngine(FS, CENTER, **kw)
        pos, outs = 0, []
        for s in sizes:
            outs.append(eng.process(batch[pos:pos + s]))
            pos += s
        for k in ("psd", "rel", "avg", "cand_idx"):
            np.testing.assert_array_equal(np.concatenate([o[k] for o in outs]), whole[k], err_msg=f"{k} {sizes}")
    assert whole["cand_off"][-1] > 100_000


def test_parseval_per_frame(batch):
    """sum_k |X[k]|^2 = N * sum_n |x[n] w[n]|^2 for every one of the 1024 frames."""
    eng = pkg.SpectrumEngine(FS, CENTER, fft_size=N, decim=1, learn_frames=100, max_batch=B)
    psd = eng.process(batch, want=("psd",))["psd"].astype(np.float64)
    k = np.arange(N)
    w = (0.54 - 0.46 * np.cos(2 * np.pi * k / (N - 1))).astype(np.float32).astype(np.float64)
    lhs = (10.0 ** (psd / 10.0)).sum(axis=1) * FS
    rhs = N * (np.abs(batch.astype(np.complex128) * w) ** 2).sum(axis=1)
    assert np.max(np.abs(lhs / rhs - 1.0)) < 2e-5


def test_gain_shifts_db_by_constant(batch):
    """Scaling the IQ by 2 moves every PSD bin by 20*log10(2) dB and leaves rel/avg/candidates unchanged
    once the ceiling is learned on equally scaled noise."""
    kw = dict(fft_size=N, decim=1, learn_frames=100, max_batch=B)
    a = pkg.SpectrumEngine(FS, CENTER, **kw).process(batch[:256])
    b = pkg.SpectrumEngine(FS, CENTER, **kw).process((batch[:256] * np.float32(2.0)).astype(np.complex64))
    assert np.max(np.abs(b["psd"] - a["psd"] - 20 * np.log10(2.0))) < 2e-5 * 60
    fin = a["rel"] != -100
    assert np.max(np.abs(b["rel"][fin] - a["rel"][fin])) < 5e-5
    np.testing.assert_array_equal(a["cand_off"], b["cand_off"])


def test_tone_lands_on_shifted_bin():
    """A complex exponential at +k bins shows up at index N/2 + k (fft_v shift=true), for a frame in every
    position of a full batch."""
    eng = pkg.SpectrumEngine(FS, CENTER, fft_size=N, decim=1, learn_frames=1, max_batch=B)
    ks = (np.arange(B) * 37 - 4000) % N - N // 2  # -4096 .. 4095
    n = np.arange(N)
    iq = np.exp(2j * np.pi * ks[:, None] * n[None, :] / N).astype(np.complex64)
    psd = eng.process(iq, want=("psd",))["psd"]
    np.testing.assert_array_equal(np.argmax(psd, axis=1), (N // 2 + ks) % N)


def test_frame_range_sharding_equals_single_rank():
    """BASELINE config 5's sharding: ranks scan contiguous frame ranges of one band with a re-read halo
    and no exchange. Simulated here rank by rank on one GPU: the union of the ranks' candidate lists is the single-rank
    scan, bit for bit."""
    n, fs, center = 8192, 2_048_000, 145_000_000
    nframes, learn, world = 1500, 100, 4
    band = pkg.synth.SyntheticBand(n, seed=33, on_frame=150, off_frame=1400)
    frames = band.frames_cf32(nframes)
    kw = dict(fft_size=n, decim=1, learn_frames=learn, max_batch=256)
    single = pkg.dist.scan_frame_range(pkg.SpectrumEngine(fs, center, **kw), frames, 0, nframes, learn, 256)
    assert len(single) == nframes and sum(len(c) for c in single) > 100_000
    got = []
    for rank in range(world):
        _, lo, hi = pkg.dist.frame_ranges(nframes, rank, world, 20)
        part = pkg.dist.scan_frame_range(pkg.SpectrumEngine(fs, center, **kw), frames, lo, hi, learn, 256)
        assert len(part) == hi - lo
        got.extend(part)
    assert len(got) == nframes
    for f in range(nframes):
        np.testing.assert_array_equal(got[f], single[f], err_msg=f"frame {f}")


def test_full_size_calls_in_flight_equal_call_by_call():
    """The headline shape through the device entry point, the way bench.py drives it: six 1024-frame calls without a
    synchronisation in between (deep pipelining: launches of consecutive calls overlap on two queues, five calls in flight)
    against the same frames call by call through the host entry point. Same bits: PSD plane, offsets, candidate lists."""
    import torch
    dev = torch.device("cuda:0")
    ncalls, learn = 6, 100
    band = pkg.synth.SyntheticBand(N, seed=35, on_frame=300, off_frame=5000, period=5600)
    kw = dict(fft_size=N, decim=1, learn_frames=learn, max_batch=B)
    ref, eng = pkg.SpectrumEngine(FS, CENTER, **kw), pkg.SpectrumEngine(FS, CENTER, **kw)
    d_iq, outs, want = [], [], []
    for k in range(ncalls):
        chunk = band.frames_cf32(B)
        want.append(ref.process(chunk, want=("psd",)))
        d_iq.append(torch.from_numpy(chunk.view(np.float32)).to(dev))
        outs.append(dict(psd=torch.empty((B, N), dtype=torch.float32, device=dev), off=torch.full((B + 1,), -1, dtype=torch.int32, device=dev),
                         idx=torch.empty(B * 512, dtype=torch.int32, device=dev), cav=torch.empty(B * 512, dtype=torch.float32, device=dev)))
    torch.cuda.synchronize()
    for k in range(ncalls):
        o = outs[k]
        eng.process_device(d_iq[k], B, psd=o["psd"], cand_off=o["off"], cand_idx=o["idx"], cand_avg=o["cav"])
    if os.environ.get("SS_PIPELINE") != "0":
        assert int(outs[ncalls - 1]["off"][-1]) == -1  # (nothing of the last call's candidate stage has run yet)
    eng.sync()
    total = 0
    for k in range(ncalls):
        o, w = outs[k], want[k]
        np.testing.assert_array_equal(o["off"].cpu().numpy(), w["cand_off"], err_msg=f"call {k}")
        t = int(w["cand_off"][-1])
        np.testing.assert_array_equal(o["idx"][:t].cpu().numpy(), w["cand_idx"], err_msg=f"call {k}")
        np.testing.assert_array_equal(o["cav"][:t].cpu().numpy(), w["cand_avg"], err_msg=f"call {k}")
        np.testing.assert_array_equal(o["psd"].cpu().numpy(), w["psd"], err_msg=f"call {k}")
        total += t
    assert total > 100_000
