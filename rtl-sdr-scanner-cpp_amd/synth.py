"""Seeded synthetic IQ for the scan path.

Complex Gaussian noise plus K wide multitone combs that are gated on and off, so that the noise
learning phase, the -100 warm-up of the averager, signal start and signal stop are all exercised.
The detector averages dB values over 21 bins x 21 frames (reference sources/utils/utils.cpp:31-53,
sources/radio/averager.cpp:52-61), so transmissions must be wide (a 1-bin tone never reaches
start_level); each comb is W adjacent bin-centre exponentials, i.e. a ~W*step wide NFM-like block.
"""
from __future__ import annotations

import numpy as np

COMB_CENTRES = (0.18, 0.31, -0.27, -0.42)  # fractions of fs


def comb_amplitude(n: int, sigma: float, rel_db: float = 25.0, ceiling: float = 8.0) -> float:
    """Per-tone amplitude that puts the comb about `rel_db` above the LEARNED noise ceiling (the max over
    the learning frames, about `ceiling` x the mean noise power). 0.54 / 0.3974 are the Hamming coherent /
    power gains."""
    return float(10.0 ** (rel_db / 20.0) * np.sqrt(ceiling * 2.0 * sigma * sigma * 0.3974 * n) / (0.54 * n))


class SyntheticBand:
    """Deterministic generator of `fft_size*decim`-sample items for one band."""

    def __init__(self, fft_size: int, decim: int = 1, seed: int = 0, sigma: float = 0.05, comb_width: int = 48,
                 rel_db: float = 25.0, on_frame: int = 130, off_frame: int = 330, centres=COMB_CENTRES, period: int = 0,
                 start_frame: int = 0):
        self.n = int(fft_size)
        self.decim = int(decim)
        self.sigma = float(sigma)
        self.rng = np.random.default_rng(seed)
        self.on_frame = int(on_frame)
        self.off_frame = int(off_frame)
        self.period = int(period)  # > 0: the on/off gate repeats every `period` frames (a long stream for the benchmark)
        self.frame = int(start_frame)
        n = self.n
        w = min(comb_width, max(2, n // 32))
        amp = comb_amplitude(n, sigma, rel_db)
        self.comb_bins = []  # shifted-spectrum bin indexes (DC at n/2) covered by each comb
        self.combs = []
        for c in centres:
            centre = int(round(c * n))  # FFT bin relative to DC
            ks = (centre + np.arange(w) - w // 2) % n
            spec = np.zeros(n, dtype=np.complex128)
            spec[ks] = amp * np.exp(2j * np.pi * self.rng.random(w))
            wave = np.fft.ifft(spec) * n  # periodic in n samples -> identical in every item
            self.combs.append(np.tile(wave, self.decim).astype(np.complex64))
            self.comb_bins.append(np.sort((ks + n // 2) % n))

    def active(self, frame: int) -> bool:
        if self.period > 0:
            frame %= self.period
        return self.on_frame <= frame < self.off_frame

    def frames_cf32(self, nframes: int) -> np.ndarray:
        """[nframes, fft_size*decim] complex64."""
        m = self.n * self.decim
        out = np.empty((nframes, m), dtype=np.complex64)
        for f in range(nframes):
            z = self.rng.standard_normal((m, 2), dtype=np.float32) * np.float32(self.sigma)
            x = z[:, 0] + 1j * z[:, 1]
            if self.active(self.frame):
                for c in self.combs:
                    x = x + c
            out[f] = x
            self.frame += 1
        return out

    def frames_cs8(self, nframes: int, full_scale: float = 0.5) -> np.ndarray:
        """[nframes, fft_size*decim, 2] int8 (HackRF-shaped): cf32 * 128/full_scale rounded and clipped."""
        x = self.frames_cf32(nframes)
        y = np.stack([x.real, x.imag], axis=-1) * (128.0 / full_scale)
        return np.clip(np.rint(y), -128, 127).astype(np.int8)

    def frames_cu8(self, nframes: int, full_scale: float = 0.5) -> np.ndarray:
        x = self.frames_cf32(nframes)
        y = np.stack([x.real, x.imag], axis=-1) * (127.5 / full_scale) + 127.5
        return np.clip(np.rint(y), 0, 255).astype(np.uint8)
