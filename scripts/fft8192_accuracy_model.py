#!/usr/bin/env python
"""Where the 8192-point transform's distance to an fp64 FFT comes from (round 5's parity blocks: engine / reference rms 1.24): the
transform of csrc/fft8192_v2.h restated in numpy with fp32 arithmetic (every fmaf emulated through fp64) — Stockham passes of radix 16,
16 and 32 with the kernel's own butterflies (dft4 / dft16 / the W_32 rotations, the radix-32 tail as two 16-point DFTs of lane pairs) —
against the oracle's radix-2 fp32 transform (oracle/specscan_oracle.c:207-236) on the same windowed frames, with the pass-3 twiddle
W_8192^(j r) formed in different ways:
    ship     as the kernel forms it: (lane part x wave part) for W_8192^(j a), a = r & 3, and for W_2048^(j q2), q2 = r >> 2 — four rounded
             table entries and three complex products per twiddle
    tw0      the first-generation tables: W_8192^(j a) and W_2048^(j q2) rounded once each, one product
    split16  W_8192^(j r) = W_8192^(j (r & 15)) * W_512^(j (r >> 4)): the second factor is 1 for half of the inputs
    exact    W_8192^(j r) rounded once from fp64 (no product at all: what a 64 KiB table would give)
No GPU needed:  python scripts/fft8192_accuracy_model.py [frames]"""
import sys

import numpy as np

f32 = np.float32


def fma(a, b, c):
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(f32)


def cmul(a, b):
    return (fma(a[0], b[0], -(a[1] * b[1])), fma(a[0], b[1], a[1] * b[0]))


def cadd(a, b):
    return (a[0] + b[0], a[1] + b[1])


def csub(a, b):
    return (a[0] - b[0], a[1] - b[1])


def cmul_mi(a):  # times -i
    return (a[1], -a[0])


KC32 = [f32(1.0), f32(0.98078528040323043), f32(0.92387953251128674), f32(0.83146961230254524), f32(0.70710678118654757)]
KS32 = [f32(0.0), f32(0.19509032201612825), f32(0.38268343236508978), f32(0.55557023301960218), f32(0.70710678118654757)]


def mulw32(x, K):
    k = K % 32
    if k == 0:
        return x
    if k == 8:
        return (x[1], -x[0])
    if k == 16:
        return (-x[0], -x[1])
    if k == 24:
        return (-x[1], x[0])
    q, r = k // 8, k % 8
    c0 = KC32[r] if r <= 4 else KS32[8 - r]
    s0 = KS32[r] if r <= 4 else KC32[8 - r]
    c = [c0, -s0, -c0, s0][q]
    s = [-s0, -c0, s0, c0][q]
    return (fma(x[0], c, -(x[1] * s)), fma(x[0], s, x[1] * c))


def dft4(v, i0, i1, i2, i3):
    a0, a1, a2, a3 = v[i0], v[i1], v[i2], v[i3]
    s0, s1, s2, s3 = cadd(a0, a2), csub(a0, a2), cadd(a1, a3), cmul_mi(csub(a1, a3))
    v[i0], v[i1], v[i2], v[i3] = cadd(s0, s2), cadd(s1, s3), csub(s0, s2), csub(s1, s3)


def slot16(k):
    return 4 * (k & 3) + (k >> 2)


def dft16(v):
    for n2 in range(4):
        dft4(v, n2, n2 + 4, n2 + 8, n2 + 12)
    for idx, K in ((5, 2), (6, 4), (7, 6), (9, 4), (10, 8), (11, 12), (13, 6), (14, 12), (15, 18)):
        v[idx] = mulw32(v[idx], K)
    for k1 in range(4):
        dft4(v, 4 * k1, 4 * k1 + 1, 4 * k1 + 2, 4 * k1 + 3)


def W(num, den):
    ang = -2.0 * np.pi * np.asarray(num, np.float64) / den
    return (np.cos(ang).astype(f32), np.sin(ang).astype(f32))


def engine_fft(x, mode):
    """x: (re, im) of shape [F, 8192] fp32, windowed. Returns X in natural order."""
    F = x[0].shape[0]
    t = np.arange(512)
    # pass 1: radix 16, Ns = 1
    v = [(x[0][:, t + 512 * r], x[1][:, t + 512 * r]) for r in range(16)]
    dft16(v)
    y = (np.empty((F, 8192), f32), np.empty((F, 8192), f32))
    for k in range(16):
        y[0][:, 16 * t + k], y[1][:, 16 * t + k] = v[slot16(k)]
    # pass 2: radix 16, Ns = 16
    m = t & 15
    v = [(y[0][:, t + 512 * r], y[1][:, t + 512 * r]) for r in range(16)]
    for r in range(1, 16):
        v[r] = cmul(v[r], W(m * r, 256.0))
    dft16(v)
    z = (np.empty((F, 8192), f32), np.empty((F, 8192), f32))
    zbase = ((t >> 4) << 8) + (t & 15)
    for k in range(16):
        z[0][:, zbase + 16 * k], z[1][:, zbase + 16 * k] = v[slot16(k)]
    # pass 3: radix 32, Ns = 256, lane pairs (h = 0, 1) share butterfly j
    X = (np.empty((F, 8192), f32), np.empty((F, 8192), f32))
    j = np.arange(256)
    w, lam = j >> 5, j & 31
    A = []
    for h in range(2):
        a = [(z[0][:, j + 256 * (2 * q + h)], z[1][:, j + 256 * (2 * q + h)]) for q in range(16)]
        for q in range(16):
            r = 2 * q + h
            if mode == "exact":
                tw = W(j * r, 8192.0)
            elif mode == "split16":
                tw = W(j * (r & 15), 8192.0)
                if r >> 4:
                    tw = cmul(tw, W(j * (r >> 4), 512.0))
            else:
                a_idx, q2 = r & 3, r >> 2
                if mode == "ship":
                    wa = cmul(W(lam * a_idx, 8192.0), W(w * a_idx, 256.0))
                    wb = cmul(W(lam * q2, 2048.0), W(w * q2, 64.0))
                else:  # tw0
                    wa = W(j * a_idx, 8192.0)
                    wb = W(j * q2, 2048.0)
                tw = wa if q2 == 0 else cmul(wa, wb)
            a[q] = cmul(a[q], tw)
        dft16(a)
        A.append(a)
    for k in range(16):
        e = A[0][slot16(k)]
        o = mulw32(A[1][slot16(k)], k)
        Xe, Xo = cadd(e, o), csub(e, o)
        X[0][:, j + 256 * k], X[1][:, j + 256 * k] = Xe
        X[0][:, j + 256 * (k + 16)], X[1][:, j + 256 * (k + 16)] = Xo
    return X


def radix2_fft(x):
    """The oracle's textbook radix-2 DIT in fp32 (no contraction: -ffp-contract=off), twiddles rounded from fp64."""
    n = x[0].shape[1]
    bits = n.bit_length() - 1
    idx = np.arange(n)
    rev = np.zeros(n, np.int64)
    for b in range(bits):
        rev |= ((idx >> b) & 1) << (bits - 1 - b)
    re, im = x[0][:, rev].copy(), x[1][:, rev].copy()  # x[rev[i]] = in[i]  <=>  out position j takes in[rev[j]] (an involution)
    tw = W(np.arange(n // 2), float(n))
    length = 2
    while length <= n:
        half, tstep = length // 2, n // length
        k = np.arange(half)
        wr, wi = tw[0][k * tstep], tw[1][k * tstep]
        R = re.reshape(-1, n // length, length)
        I = im.reshape(-1, n // length, length)
        a_r, a_i, b_r, b_i = R[:, :, :half], I[:, :, :half], R[:, :, half:], I[:, :, half:]
        tr = b_r * wr - b_i * wi
        ti = b_r * wi + b_i * wr
        nb_r, nb_i, na_r, na_i = a_r - tr, a_i - ti, a_r + tr, a_i + ti
        R[:, :, half:], I[:, :, half:], R[:, :, :half], I[:, :, :half] = nb_r, nb_i, na_r, na_i
        length *= 2
    return (re, im)


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    rng = np.random.default_rng(3)
    n = 8192
    iq = (rng.standard_normal((frames, n)) + 1j * rng.standard_normal((frames, n))).astype(np.complex64)
    # (a comb of strong tones on the noise, like the benchmark's band: deep nulls between them)
    k = np.arange(n)
    for c in (0.11, -0.23, 0.37):
        iq += (30.0 * np.exp(2j * np.pi * c * k)).astype(np.complex64)
    win = (0.54 - 0.46 * np.cos(2 * np.pi * k / (n - 1))).astype(f32)
    x = ((iq.real.astype(f32) * win), (iq.imag.astype(f32) * win))
    exact = np.fft.fft(x[0].astype(np.float64) + 1j * x[1].astype(np.float64), axis=1)
    rms = np.sqrt((np.abs(exact) ** 2).mean())
    ref = radix2_fft(x)
    e_ref = np.sqrt((np.abs((ref[0].astype(np.float64) + 1j * ref[1]) - exact) ** 2).mean()) / rms
    print(f"oracle radix-2 fp32        : rms |X - X64| / rms |X64| = {e_ref:.3e}")
    for mode in ("ship", "tw0", "split16", "exact"):
        X = engine_fft(x, mode)
        e = np.sqrt((np.abs((X[0].astype(np.float64) + 1j * X[1]) - exact) ** 2).mean()) / rms
        print(f"engine 16.16.32, {mode:8s}: {e:.3e}   engine / oracle = {e / e_ref:.2f}")


if __name__ == "__main__":
    main()
