import sys; sys.path.insert(0,'.')
import numpy as np
import rtl_sdr_scanner_cpp_amd as pkg
from oracle import oracle as O
for n, fs, nf in ((8192, 2048000, 16), (65536, 20000000, 8), (1<<20, 61440000, 2), (1024, 256000, 32)):
    band = pkg.synth.SyntheticBand(n, seed=41, on_frame=0, off_frame=100)
    iq = band.frames_cf32(nf)
    kw = dict(fft_size=n, decim=1, learn_frames=1, max_batch=nf)
    got = pkg.SpectrumEngine(fs, 145_000_000, **kw).process(iq, want=("psd",))["psd"].astype(np.float64)
    ref = O.oracle_chain(fs, 145_000_000, **kw).process(iq, want=("psd",))["psd"].astype(np.float64)
    k = np.arange(n)
    w = (0.54 - 0.46 * np.cos(2 * np.pi * k / (n - 1))).astype(np.float32).astype(np.float64)
    exact = np.abs(np.fft.fftshift(np.fft.fft(iq.astype(np.complex128) * w, axis=1), axes=1)) ** 2 / fs
    rms = np.sqrt(exact.mean(axis=1, keepdims=True))
    ae = lambda db: np.abs(np.sqrt(10.0 ** (db / 10.0)) - np.sqrt(exact)) / rms
    eg, eo = ae(got), ae(ref)
    weak = exact < exact.mean(axis=1, keepdims=True)  # bins below the frame mean
    print(n, "gpu rms %.2e max %.2e weakmax %.2e | oracle rms %.2e max %.2e weakmax %.2e" % (np.sqrt((eg**2).mean()), eg.max(), eg[weak].max(), np.sqrt((eo**2).mean()), eo.max(), eo[weak].max()))
