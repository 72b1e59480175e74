"""Recorder channeliser on the GPU against the CPU oracle (oracle/channelizer_oracle.c), through the
C ABI (include/specscan_channelizer.h). Run with -m gpu.

What "parity" means here: the resampler cascade is compared sample by sample (2e-5 of the signal's
scale; summation order differs). The rotator of the reference is an fp32 recurrence whose phase creeps by a few 1e-8
rad per sample relative to the rotation it stands for (tests/test_channelizer_oracle.py); the engine evaluates that
rotation in closed form, so outputs are compared modulo a slow linear phase creep per slot, bounded at 1e-7 rad per
input sample, and exactly (tight tolerance, int8 included) where the creep vanishes: shift 0, short runs."""
import numpy as np
import pytest

import rtl_sdr_scanner_cpp_amd as pkg
from rtl_sdr_scanner_cpp_amd.channelizer import Channelizer
from oracle import oracle

pytestmark = pytest.mark.gpu


def _stream(n, fs, carriers, seed):
    """Noise plus a few modulated carriers at the given offsets (Hz): something for every slot to pull out."""
    rng = np.random.default_rng(seed)
    t = np.arange(n) / fs
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.02
    for k, f in enumerate(carriers):
        am = 0.15 * (1.0 + 0.5 * np.sin(2 * np.pi * (700 + 300 * k) * t))
        x += am * np.exp(2j * np.pi * (f * t + 0.3 * np.sin(2 * np.pi * (1100 + 200 * k) * t)))
    return x.astype(np.complex64)


def _decreep(got, ref, in_per_out):
    """Remove the best-fit linear phase ramp between got and ref; returns (residual max |error| / max |ref|, slope per input sample)."""
    if len(ref) < 8:
        return float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-9)), 0.0
    w = np.abs(ref) ** 2
    d = np.angle(got * np.conj(ref))
    k = np.arange(len(ref), dtype=np.float64)
    slope = float((w * k) @ d / ((w * k) @ k))
    fixed = got * np.exp(-1j * slope * k)
    return float(np.abs(fixed - ref).max() / np.abs(ref).max()), slope / in_per_out


def test_stage_design_equals_the_oracle():
    for fs, bw in ((2_048_000, 32_000), (2_048_000, 16_000), (1_024_000, 20_000), (1_000_000, 16_000), (20_000_000, 32_000)):
        ch = Channelizer(fs, bw, channels=1, max_samples=4096)
        o = oracle.ChannelizerOracle(fs, bw)
        assert ch.stages == o.stages, (fs, bw)
        for s, (i, d, _n) in enumerate(ch.stages):
            np.testing.assert_array_equal(ch.stage_taps(s), oracle.design_taps(i, d))
        assert [(i, d) for i, d, _ in ch.stages] == oracle.resampler_factors(fs, bw)
        ch.close()


@pytest.mark.parametrize("fs,bw,shifts,n", [
    (2_048_000, 32_000, [250_000, -613_500, 12_500], 300_000),  # (1,64): the reference's default recording bandwidth
    (2_048_000, 16_000, [100_000, -400_000], 200_000),            # (1,8),(1,16)
    (1_024_000, 20_000, [-200_000, 33_000], 150_000),             # (1,16),(5,16): interpolation > 1 in the second stage
    (1_000_000, 16_000, [150_000], 120_000),                      # (2,125): interpolation > 1 at the input rate
])
def test_channels_match_the_oracle(fs, bw, shifts, n):
    x = _stream(n, fs, shifts, seed=len(shifts) + bw)
    ch = Channelizer(fs, bw, channels=len(shifts) + 1, max_samples=1 << 17)
    oracles = []
    for k, sh in enumerate(shifts):
        ch.start(k, sh)
        o = oracle.ChannelizerOracle(fs, bw)
        o.set_shift(sh)
        oracles.append(o)
    sizes, pos = [], 0
    rng = np.random.default_rng(5)
    while pos < n:
        s = int(min(n - pos, rng.integers(1, 1 << 17)))
        sizes.append(s)
        pos += s
    got = {k: ([], []) for k in range(len(shifts))}
    ref = {k: ([], []) for k in range(len(shifts))}
    pos = 0
    for s in sizes:
        out = ch.process(x[pos:pos + s])
        assert sorted(out) == list(range(len(shifts)))  # the idle slot reports nothing
        for k in range(len(shifts)):
            got[k][0].append(out[k][0])
            got[k][1].append(out[k][1])
            y, i8 = oracles[k].process(x[pos:pos + s])
            ref[k][0].append(i8)
            ref[k][1].append(y)
        pos += s
    in_per_out = fs / bw
    for k in range(len(shifts)):
        g8, gy = np.concatenate(got[k][0]), np.concatenate(got[k][1])
        r8, ry = np.concatenate(ref[k][0]), np.concatenate(ref[k][1])
        assert len(gy) == len(ry) and len(ry) >= n * bw // fs - 2
        resid, slope = _decreep(gy, ry, in_per_out)
        assert resid < 1.5e-4 and abs(slope) < 1e-7, (k, resid, slope)
        assert np.abs(ry).max() > 0.05  # the slot really pulled its carrier out
        # the engine's own int8 is exactly the conversion of its own float output
        r = gy.view(np.float32).reshape(-1, 2) * np.float32(127.0)
        np.testing.assert_array_equal(g8, np.clip(np.rint(r), -128, 127).astype(np.int8))
        # against the oracle's int8 while the creep is still below a tenth of an LSB (the first 20k input samples)
        m = int(20_000 / in_per_out)
        diff = np.abs(g8[:m].astype(np.int32) - r8[:m].astype(np.int32))
        assert diff.max() <= 1 and (diff != 0).mean() < 0.03, (k, diff.max(), (diff != 0).mean())
    ch.close()


def test_no_creep_cases_match_tightly_int8_included():
    """Shift 0 (identity rotator) and half the sample rate (increment exactly -1): no creep, so the whole stream compares
    directly — floats to 2e-5 of full scale, int8 equal except at exact rounding ties of the scaled value."""
    fs, bw, n = 2_048_000, 32_000, 400_000
    x = _stream(n, fs, [0.0, 3_000.0], seed=9)
    for shift in (0, fs // 2):
        ch = Channelizer(fs, bw, channels=2, max_samples=1 << 18)
        ch.start(1, shift)
        o = oracle.ChannelizerOracle(fs, bw)
        o.set_shift(shift)
        gy, g8, ry, r8 = [], [], [], []
        for a in range(0, n, 1 << 18):
            out = ch.process(x[a:a + (1 << 18)])
            assert list(out) == [1]
            g8.append(out[1][0])
            gy.append(out[1][1])
            y, i8 = o.process(x[a:a + (1 << 18)])
            ry.append(y)
            r8.append(i8)
        gy, g8, ry, r8 = map(np.concatenate, (gy, g8, ry, r8))
        scale = np.abs(ry).max()
        assert np.abs(gy - ry).max() < 2e-5 * scale + 1e-6, shift  # + the rounding floor set by the input's scale (0.4)
        diff = g8.astype(np.int32) - r8.astype(np.int32)
        near_tie = np.abs((ry.view(np.float32).reshape(-1, 2) * 127.0) % 1.0 - 0.5) < 0.01
        assert np.abs(diff).max() <= 1 and not (diff != 0)[~near_tie].any()
        ch.close()


def test_result_does_not_depend_on_the_call_sizes():
    fs, bw, n = 2_048_000, 16_000, 150_000
    x = _stream(n, fs, [-300_000.0], seed=3)
    whole = Channelizer(fs, bw, channels=1, max_samples=n)
    whole.start(0, -300_000)
    w8, wy = whole.process(x)[0]
    cut = Channelizer(fs, bw, channels=1, max_samples=n)
    cut.start(0, -300_000)
    parts, pos = [], 0
    for s in (1, 7, 127, 128, 129, 5000, 64, 100_000, 44_544):
        parts.append(cut.process(x[pos:pos + s])[0])
        pos += s
    assert pos == n
    cy, c8 = np.concatenate([p[1] for p in parts]), np.concatenate([p[0] for p in parts])
    assert len(cy) == len(wy)
    assert np.abs(cy - wy).max() < 1e-6 * np.abs(wy).max()  # closed-form phase: only the fp64 phase bookkeeping differs
    assert (c8 != w8).mean() < 1e-3


@pytest.mark.parametrize("fs,bw", [(2_048_000, 32_000), (2_000_000, 20_000), (2_048_000, 16_000), (250_000, 25_000)])
def test_branch_kernel_equals_the_generic_stage_kernel(fs, bw, monkeypatch, diag_lib):
    """The polyphase-by-branch first stage (full-tile and edge code) against the one-output-per-lane kernel (SC_GENERIC=1)
    on the same stream cut into the same ragged calls: two summation orders of the same fp32 products."""
    n = 140_000
    x = _stream(n, fs, [fs / 7.0, -fs / 5.0], seed=9)
    outs = []
    for generic in ("0", "1"):
        monkeypatch.setenv("SC_GENERIC", generic)
        ch = Channelizer(fs, bw, channels=2, max_samples=1 << 16)
        ch.start(0, int(fs / 7))
        ch.start(1, int(-fs / 5))
        parts, pos = [[], []], 0
        for s_ in (3, 40_000, 65_536, 1, 20_000, 14_460):
            r = ch.process(x[pos:pos + s_])
            for k in (0, 1):
                parts[k].append(r[k][1])
            pos += s_
        assert pos == n
        outs.append([np.concatenate(p) for p in parts])
        ch.close()
    for k in (0, 1):
        a, b = outs[0][k], outs[1][k]
        assert len(a) == len(b) > 500
        assert np.abs(a - b).max() < 3e-6 * np.abs(b).max()


def test_start_stop_keeps_state_like_the_reference():
    """An idle slot sees no samples (Blocker drops them): its rotator phase and filter histories stay as they were and
    the next recording starts from them (recorder.cpp:58-87 never resets the blocks)."""
    fs, bw = 1_024_000, 16_000  # (1,64)
    x = _stream(120_000, fs, [50_000.0, -120_000.0], seed=4)
    ch = Channelizer(fs, bw, channels=2, max_samples=1 << 16)
    o = oracle.ChannelizerOracle(fs, bw)
    ch.start(0, 50_000)
    o.set_shift(50_000)
    a = ch.process(x[:40_000])[0]
    ra = o.process(x[:40_000])
    ch.stop(0)
    assert not ch.is_recording(0) and ch.process(x[40_000:60_000]) == {}  # dropped while idle; the oracle is simply not fed
    ch.start(0, -120_000)
    o.set_shift(-120_000)
    b = ch.process(x[60_000:])[0]
    rb = o.process(x[60_000:])
    for (g8, gy), (ry, r8) in ((a, ra), (b, rb)):
        resid, slope = _decreep(gy, ry, fs / bw)
        assert len(gy) == len(ry) and resid < 1.5e-4 and abs(slope) < 1e-7


def test_device_entry_point_and_counts():
    import torch
    fs, bw, n = 2_048_000, 32_000, 1 << 18
    x = _stream(n, fs, [200_000.0], seed=6)
    ch = Channelizer(fs, bw, channels=3, max_samples=n)
    ch.start(2, 200_000)
    cap = ch.output_capacity(n)
    dev = torch.device("cuda:0")
    d_iq = torch.from_numpy(x.view(np.float32).copy()).to(dev)
    d_i8 = torch.zeros((3, cap, 2), dtype=torch.int8, device=dev)
    d_cf = torch.zeros((3, cap, 2), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()  # (torch's fills run on torch's stream, the library writes from its own)
    counts = ch.process_device(d_iq, n, d_i8, d_cf, cap)
    ch.sync()
    assert list(counts) == [0, 0, n // 64]
    host = Channelizer(fs, bw, channels=3, max_samples=n)
    host.start(2, 200_000)
    h8, hy = host.process(x)[2]
    np.testing.assert_array_equal(d_i8[2, :counts[2]].cpu().numpy(), h8)
    np.testing.assert_array_equal(d_cf[2, :counts[2]].cpu().numpy().view(np.complex64).reshape(-1), hy)
    assert not d_i8[:2].any().item()


def test_argument_errors():
    ch = Channelizer(2_048_000, 32_000, channels=2, max_samples=4096)
    with pytest.raises(pkg.abi.SpecscanError):
        ch.start(2, 1000)
    with pytest.raises(pkg.abi.SpecscanError):
        ch.process(np.zeros(4097, np.complex64))
    with pytest.raises(pkg.abi.SpecscanError):
        Channelizer(2_048_000, 32_000, channels=17)
    assert ch.process(np.zeros(100, np.complex64)) == {}


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("SC_FUZZ_SEEDS", "8"))))
def test_random_recording_sessions(seed):
    """Random sample rates / bandwidths (single and multi stage, interpolating stages included), random slots starting,
    stopping and restarting on other shifts at random times, random call sizes: every slot against its own oracle chain
    fed exactly the samples it was recording for."""
    rng = np.random.default_rng(4000 + seed)
    fs, bw = [(2_048_000, 32_000), (2_048_000, 16_000), (1_024_000, 20_000), (1_000_000, 16_000), (2_400_000, 32_000),
              (250_000, 25_000), (1_024_000, 16_000), (2_000_000, 20_000)][int(rng.integers(0, 8))]
    nslots = int(rng.integers(1, 5))
    n = int(rng.integers(60_000, 160_000))
    carriers = [float(rng.integers(-fs // 3, fs // 3)) for _ in range(3)]
    x = _stream(n, fs, carriers, seed=seed)
    ch = Channelizer(fs, bw, channels=nslots, max_samples=1 << 16)
    oracles = [oracle.ChannelizerOracle(fs, bw) for _ in range(nslots)]
    active = [False] * nslots
    segs = [[] for _ in range(nslots)]  # per slot: list of (got_cf32, ref_cf32) per recording session
    cur = [None] * nslots
    pos = 0
    while pos < n:
        for k in range(nslots):  # random control events between calls
            r = rng.random()
            if not active[k] and r < 0.35:
                sh = int(rng.choice(carriers) + rng.integers(-2000, 2000))
                ch.start(k, sh)
                oracles[k].set_shift(sh)
                active[k] = True
                cur[k] = ([], [])
            elif active[k] and r < 0.08:
                ch.stop(k)
                active[k] = False
                segs[k].append(cur[k])
                cur[k] = None
        size = int(min(n - pos, rng.integers(1, 1 << 16)))
        out = ch.process(x[pos:pos + size])
        assert sorted(out) == [k for k in range(nslots) if active[k]]
        for k in out:
            y, _i8 = oracles[k].process(x[pos:pos + size])
            assert len(out[k][1]) == len(y), (seed, k, len(out[k][1]), len(y))
            cur[k][0].append(out[k][1])
            cur[k][1].append(y)
        pos += size
    for k in range(nslots):
        if cur[k] is not None:
            segs[k].append(cur[k])
        for got, ref in segs[k]:
            if not got:
                continue
            g, r = np.concatenate(got), np.concatenate(ref)
            if len(r) < 40 or np.abs(r).max() < 1e-3:
                continue
            # every session restarts from the state the previous one left (phase, histories): compare modulo the creep
            resid, slope = _decreep(g, r, fs / bw)
            assert resid < 3e-4 and abs(slope) < 1e-7, (seed, k, len(r), resid, slope)
    ch.close()
