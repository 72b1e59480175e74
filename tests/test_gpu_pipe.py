"""ss_pipe_*: one band, several lanes taking its calls in turn (include/specscan.h). Every frame past the averager's warm-up must
come out exactly as from one context fed the same calls — candidates and PSD bit for bit — whatever the call sizes are."""
import numpy as np
import pytest

import rtl_sdr_scanner_cpp_amd as pkg

pytestmark = pytest.mark.gpu


def _scan(obj, d_iq_all, calls, n, item, dev, torch):
    """Feed `calls` (list of frame counts) and return per-call (psd, cand_off, cand_idx, cand_avg) as numpy."""
    outs, pos = [], 0
    bufs = []
    for k, nf in enumerate(calls):
        psd = torch.empty((nf, n), dtype=torch.float32, device=dev)
        off = torch.zeros(nf + 1, dtype=torch.int32, device=dev)
        idx = torch.empty(nf * 512, dtype=torch.int32, device=dev)
        avg = torch.empty(nf * 512, dtype=torch.float32, device=dev)
        obj.process_device(d_iq_all[pos:pos + nf], nf, psd=psd, cand_off=off, cand_idx=idx, cand_avg=avg)
        bufs.append((psd, off, idx, avg))
        pos += nf
    obj.sync()
    for psd, off, idx, avg in bufs:
        o = off.cpu().numpy()
        outs.append((psd.cpu().numpy(), o, idx.cpu().numpy()[:o[-1]], avg.cpu().numpy()[:o[-1]]))
    return outs


@pytest.mark.parametrize("lanes,n,decim,fmt,calls", [
    (2, 8192, 1, "cf32", [100, 128, 128, 128, 96, 128]),                       # learning (100) then big calls in turn
    (3, 2048, 1, "cf32", [64, 70, 200, 8, 130, 129, 64, 3, 1, 250, 100, 77]),  # ragged: small calls go to every lane, odd alignments
    (2, 1024, 3, "cs8", [50, 300, 300, 65, 300]),                             # frame decimation: items of 3 N samples, int8
    (4, 4096, 1, "cf32", [120, 64, 64, 64, 64, 64, 64, 64, 64]),
    (1, 2048, 1, "cf32", [90, 128, 40]),                                       # one lane: the plain sequence
])
def test_pipe_equals_one_context(lanes, n, decim, fmt, calls):
    import torch
    dev = torch.device("cuda:0")
    fs, center, learn = 250 * n, 145_000_000, 40
    total = sum(calls)
    band = pkg.synth.SyntheticBand(n, decim=decim, seed=5 + lanes, on_frame=learn + 10, off_frame=total - 20)
    if fmt == "cf32":
        iq, in_format = band.frames_cf32(total), pkg.abi.SS_FMT_CF32
        d_iq = torch.from_numpy(iq.view(np.float32).reshape(total, -1)).to(dev)
    else:
        iq, in_format = band.frames_cs8(total), pkg.abi.SS_FMT_CS8
        d_iq = torch.from_numpy(iq.reshape(total, -1)).to(dev)
    kw = dict(fft_size=n, decim=decim, in_format=in_format, learn_frames=learn, max_batch=max(max(calls), 64))
    one = pkg.SpectrumEngine(fs, center, **kw)
    ref = _scan(one, d_iq, calls, n, None, dev, torch)
    pipe = pkg.engine.Pipe(fs, center, lanes=lanes, **kw)
    got = _scan(pipe, d_iq, calls, n, None, dev, torch)
    ncand = 0
    for k, ((p1, o1, i1, a1), (p2, o2, i2, a2)) in enumerate(zip(ref, got)):
        np.testing.assert_array_equal(p1, p2, err_msg=f"psd of call {k}")
        np.testing.assert_array_equal(o1, o2, err_msg=f"offsets of call {k}")
        np.testing.assert_array_equal(i1, i2, err_msg=f"candidate bins of call {k}")
        np.testing.assert_array_equal(a1, a2, err_msg=f"candidate power of call {k}")
        ncand += int(o1[-1])
    assert ncand > 200
    pipe.close()


def test_pipe_retune_and_reset():
    """SdrDevice::setFrequencyRange on a pipe: every lane retunes and restarts; the new centre learns its own ceiling."""
    import torch
    dev = torch.device("cuda:0")
    n, fs, learn = 2048, 512_000, 30
    calls = [64, 128, 128, 128]
    total = sum(calls)
    band = pkg.synth.SyntheticBand(n, seed=77, on_frame=learn + 10, off_frame=total - 10)
    d_iq = torch.from_numpy(band.frames_cf32(total).view(np.float32).reshape(total, -1)).to(dev)
    kw = dict(fft_size=n, decim=1, learn_frames=learn, max_batch=128)
    one, pipe = pkg.SpectrumEngine(fs, 100_000_000, **kw), pkg.engine.Pipe(fs, 100_000_000, lanes=2, **kw)
    a = _scan(one, d_iq, calls, n, None, dev, torch)
    b = _scan(pipe, d_iq, calls, n, None, dev, torch)
    for obj in (one, pipe):
        obj.set_frequency_range(150_000_000 - fs // 2, 150_000_000 + fs // 2)
        obj.reset()
    a += _scan(one, d_iq, calls, n, None, dev, torch)
    b += _scan(pipe, d_iq, calls, n, None, dev, torch)
    for (p1, o1, i1, a1), (p2, o2, i2, a2) in zip(a, b):
        np.testing.assert_array_equal(o1, o2)
        np.testing.assert_array_equal(i1, i2)
        np.testing.assert_array_equal(a1, a2)
    assert sum(int(o[-1]) for _, o, _, _ in a) > 200


def test_pipe_argument_errors():
    with pytest.raises(pkg.abi.SpecscanError):
        pkg.engine.Pipe(512_000, 100_000_000, lanes=5, fft_size=2048, max_batch=128)
    with pytest.raises(pkg.abi.SpecscanError):
        pkg.engine.Pipe(512_000, 100_000_000, lanes=2, fft_size=2048, max_batch=32)
    with pytest.raises(pkg.abi.SpecscanError):
        pkg.engine.Pipe(512_000, 100_000_000, lanes=2, fft_size=2048, max_batch=128, flags=pkg.abi.SS_FLAG_SPECTROGRAM)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("SS_PIPE_SEEDS", "6"))))
def test_random_call_sequences(seed):
    """Random lane counts, frame sizes, formats, call sizes (1 .. 300 frames: long calls taken in turn, short ones by every lane,
    every alignment of the halo), with a retune + reset somewhere: a pipe and one context must agree bit for bit."""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(9000 + seed)
    lanes = int(rng.integers(2, 5))
    n = int(rng.choice([1024, 2048, 8192, 16384]))
    decim = int(rng.choice([1, 1, 2]))
    fmt = str(rng.choice(["cf32", "cs8"]))
    learn = int(rng.integers(5, 60))
    fs, center = 250 * n, 145_000_000
    calls = []
    while sum(calls) < 1200:
        calls.append(int(rng.integers(64, 301)) if rng.random() < 0.7 else int(rng.integers(1, 64)))
    total = sum(calls)
    band = pkg.synth.SyntheticBand(n, decim=decim, seed=seed, on_frame=learn + 10, off_frame=total - 30)
    if fmt == "cf32":
        in_format = pkg.abi.SS_FMT_CF32
        d_iq = torch.from_numpy(band.frames_cf32(total).view(np.float32).reshape(total, -1)).to(dev)
    else:
        in_format = pkg.abi.SS_FMT_CS8
        d_iq = torch.from_numpy(band.frames_cs8(total).reshape(total, -1)).to(dev)
    kw = dict(fft_size=n, decim=decim, in_format=in_format, learn_frames=learn, max_batch=300)
    one, pipe = pkg.SpectrumEngine(fs, center, **kw), pkg.engine.Pipe(fs, center, lanes=lanes, **kw)
    cut = int(rng.integers(2, len(calls) - 2))
    a = _scan(one, d_iq, calls[:cut], n, None, dev, torch)
    b = _scan(pipe, d_iq, calls[:cut], n, None, dev, torch)
    if rng.random() < 0.7:  # SdrDevice::setFrequencyRange: retune + Transmission::resetBuffers
        for obj in (one, pipe):
            obj.set_frequency_range(center + fs - fs // 2, center + fs + fs // 2)
            obj.reset()
    off = sum(calls[:cut])
    a += _scan(one, d_iq[off:], calls[cut:], n, None, dev, torch)
    b += _scan(pipe, d_iq[off:], calls[cut:], n, None, dev, torch)
    for k, ((p1, o1, i1, a1), (p2, o2, i2, a2)) in enumerate(zip(a, b)):
        np.testing.assert_array_equal(p1, p2, err_msg=f"seed {seed}: psd of call {k}")
        np.testing.assert_array_equal(o1, o2, err_msg=f"seed {seed}: offsets of call {k}")
        np.testing.assert_array_equal(i1, i2, err_msg=f"seed {seed}: bins of call {k}")
        np.testing.assert_array_equal(a1, a2, err_msg=f"seed {seed}: powers of call {k}")
    pipe.close()


def test_pipe_without_a_psd_plane():
    """Detect mode: the caller takes candidates only; the lanes keep the PSD rows in their own planes."""
    import torch
    dev = torch.device("cuda:0")
    n, fs, learn, calls = 8192, 2_048_000, 30, [64, 128, 128, 100, 128]
    total = sum(calls)
    band = pkg.synth.SyntheticBand(n, seed=3, on_frame=learn + 10, off_frame=total - 10)
    d_iq = torch.from_numpy(band.frames_cf32(total).view(np.float32).reshape(total, -1)).to(dev)
    kw = dict(fft_size=n, decim=1, learn_frames=learn, max_batch=128)
    one, pipe = pkg.SpectrumEngine(fs, 145_000_000, **kw), pkg.engine.Pipe(fs, 145_000_000, lanes=3, **kw)
    res = []
    for obj in (one, pipe):
        pos, bufs = 0, []
        for nf in calls:
            off = torch.zeros(nf + 1, dtype=torch.int32, device=dev)
            idx = torch.empty(nf * 512, dtype=torch.int32, device=dev)
            avg = torch.empty(nf * 512, dtype=torch.float32, device=dev)
            obj.process_device(d_iq[pos:pos + nf], nf, psd=None, cand_off=off, cand_idx=idx, cand_avg=avg)
            bufs.append((off, idx, avg))
            pos += nf
        obj.sync()
        res.append([(o.cpu().numpy(), i.cpu().numpy()[:int(o[-1])], a.cpu().numpy()[:int(o[-1])]) for o, i, a in bufs])
    for (o1, i1, a1), (o2, i2, a2) in zip(*res):
        np.testing.assert_array_equal(o1, o2)
        np.testing.assert_array_equal(i1, i2)
        np.testing.assert_array_equal(a1, a2)
    assert sum(int(o[-1]) for o, _, _ in res[0]) > 200
