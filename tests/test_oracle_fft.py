"""Anchor the PARITY-UNPINNED front end of the oracle (window, forward FFT, half rotation): it lives in
GNU Radio / VOLK / FFTW, which the reference does not vendor and no reference test covers. The restated
fft_v is checked against fp64 numpy, its fp32 back ends against each other, and against MKL's FFTW3
interface (the API gr::fft drives) where libmkl_rt is present."""
import ctypes as C

import numpy as np
import pytest

c_float_p = C.POINTER(C.c_float)


def fp(a):
    return a.ctypes.data_as(c_float_p)


def _fft_v(L, n, win, x):
    out = np.empty(n, np.complex64)
    L.orc_fft_v(n, fp(win), fp(x.view(np.float32)), fp(out.view(np.float32)))
    return out


@pytest.mark.parametrize("n", [64, 1024, 8192, 65536])
def test_hamming_and_fft_v_vs_fp64(oracle_mod, n):
    L = oracle_mod.lib()
    win = np.empty(n, np.float32)
    L.orc_hamming(n, fp(win))
    k = np.arange(n, dtype=np.float64)
    np.testing.assert_array_equal(win, (0.54 - 0.46 * np.cos(2 * np.pi * k / (n - 1))).astype(np.float32))
    assert win[0] == np.float32(0.08) and abs(win[n // 2] - 1.0) < 1e-3 and abs(win[1] - win[n - 2]) < 1e-6  # symmetric

    rng = np.random.default_rng(n)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    want = np.fft.fftshift(np.fft.fft(x.astype(np.complex128) * win.astype(np.float64)))
    scale = np.sqrt(np.mean(np.abs(want) ** 2))
    for backend in (0, 1):
        assert L.orc_set_fft_backend(backend) == 0
        got = _fft_v(L, n, win, x)
        err = np.max(np.abs(got - want)) / scale
        assert err < (5e-7 if backend == 1 else 3e-6), (backend, err)
    L.orc_set_fft_backend(0)


def test_shift_puts_dc_at_n_over_2(oracle_mod):
    L = oracle_mod.lib()
    n = 256
    ones = np.ones(n, np.float32)
    L.orc_set_fft_backend(0)
    out = _fft_v(L, n, ones, np.ones(n, np.complex64))
    assert abs(out[n // 2] - n) < 1e-3 and np.abs(np.delete(out, n // 2)).max() < 1e-3
    # a tone at +k bins lands at n/2 + k (asymmetric input: catches a mirrored spectrum)
    k = 5
    tone = np.exp(2j * np.pi * k * np.arange(n) / n).astype(np.complex64)
    out = _fft_v(L, n, ones, tone)
    assert np.argmax(np.abs(out)) == n // 2 + k


@pytest.mark.parametrize("n", [1024, 8192])
def test_fftw_interface_backend_agrees(oracle_mod, n):
    L = oracle_mod.lib()
    if L.orc_set_fft_backend(2) != 0:
        pytest.skip("no FFTW3-interface library (libmkl_rt / libfftw3f) on this host")
    try:
        rng = np.random.default_rng(7)
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
        win = np.empty(n, np.float32)
        L.orc_hamming(n, fp(win))
        a = _fft_v(L, n, win, x)
        L.orc_set_fft_backend(0)
        b = _fft_v(L, n, win, x)
        scale = np.sqrt(np.mean(np.abs(b) ** 2))
        assert np.max(np.abs(a - b)) / scale < 3e-6
    finally:
        L.orc_set_fft_backend(0)


def test_psd_formula(oracle_mod):
    """psd.cpp:19 = 10*log10f(hypotf(re,im)^2 / float(fs)); scripts/converter.py:17-21 states the same
    formula in numpy (abs(X**2)/fs -> 10*log10)."""
    L = oracle_mod.lib()
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(4096) + 1j * rng.standard_normal(4096)).astype(np.complex64) * 30
    out = np.empty(4096, np.float32)
    L.orc_psd(fp(x.view(np.float32)), fp(out), 4096, 2048000)
    want = 10 * np.log10(np.abs(x.astype(np.complex128)) ** 2 / 2048000.0)
    assert np.max(np.abs(out - want)) < 2e-5
    if oracle_mod.have_ref():
        out2 = np.empty(4096, np.float32)
        oracle_mod.ref().ref_psd(fp(x.view(np.float32)), fp(out2), 4096, 2048000)
        np.testing.assert_array_equal(out, out2)
    z = np.zeros(4, np.complex64)
    o = np.empty(4, np.float32)
    L.orc_psd(fp(z.view(np.float32)), fp(o), 4, 1000)
    assert np.isneginf(o).all()  # log10f(0) = -inf, as in the reference
