"""ctypes loader for the CPU oracle (oracle/liboracle.so) and for oracle/_ref/libref_specscan.so (the
reference's own .cpp files compiled in place). TEST INFRASTRUCTURE: imported only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg — never by the product package."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle.so")
REF_LIB = os.path.join(HERE, "_ref", "libref_specscan.so")

c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)


def build(quiet: bool = True) -> None:
    """make -C oracle (liboracle.so always; _ref only where /root/reference exists)."""
    subprocess.run(["make", "-C", HERE], check=True, stdout=subprocess.DEVNULL if quiet else None)


_lib = None
_ref = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = C.CDLL(LIB)
        L = _lib
        L.orc_hamming.argtypes = [C.c_int, c_float_p]
        L.orc_get_fft.argtypes = [C.c_int32, C.c_int32]
        L.orc_get_tuned_frequency.argtypes = [C.c_int32, C.c_int32]
        L.orc_get_tuned_frequency.restype = C.c_int32
        L.orc_index_to_shift.argtypes = [C.c_int32, C.c_int, C.c_int]
        L.orc_index_to_shift.restype = C.c_int32
        L.orc_average.argtypes = [c_float_p, c_float_p, C.c_int, C.c_int]
        L.orc_get_max_index.argtypes = [c_float_p, C.c_int, C.c_int, C.c_int]
        L.orc_contains_with_margin.argtypes = [c_int32_p, C.c_int, C.c_int, C.c_int, c_int32_p]
        L.orc_most_frequent_value.argtypes = [c_int32_p, C.c_int]
        L.orc_psd.argtypes = [c_float_p, c_float_p, C.c_int, C.c_int32]
        L.orc_set_fft_backend.argtypes = [C.c_int]
        L.orc_fft_forward.argtypes = [C.c_int, c_float_p, c_float_p]
        L.orc_fft_v.argtypes = [C.c_int, c_float_p, c_float_p, c_float_p]
        L.orc_averager_create.argtypes = [C.c_int, C.c_int]
        L.orc_averager_create.restype = C.c_void_p
        L.orc_averager_destroy.argtypes = [C.c_void_p]
        L.orc_averager_push.argtypes = [C.c_void_p, c_float_p]
        L.orc_averager_reset.argtypes = [C.c_void_p]
        L.orc_averager_average.argtypes = [C.c_void_p]
        L.orc_averager_average.restype = c_float_p
        L.orc_averager_row.argtypes = [C.c_void_p, C.c_int]
        L.orc_averager_row.restype = c_float_p
        L.orc_stage_seconds.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.orc_spectrogram_create.argtypes = [C.c_int, C.c_int32]
        L.orc_spectrogram_create.restype = C.c_void_p
        L.orc_spectrogram_destroy.argtypes = [C.c_void_p]
        L.orc_spectrogram_size.argtypes = [C.c_void_p]
        L.orc_spectrogram_process.argtypes = [C.c_void_p, c_float_p]
        L.orc_spectrogram_send.argtypes = [C.c_void_p, C.POINTER(C.c_int8), c_float_p]
        L.orc_spectrogram_payload.argtypes = [C.c_uint64, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
        # channeliser oracle (channelizer_oracle.h)
        c_int_p = C.POINTER(C.c_int)
        L.cho_resampler_factors.argtypes = [C.c_int32, C.c_int32, C.c_int, c_int_p, c_int_p, C.c_int]
        L.cho_design_taps.argtypes = [C.c_int, C.c_int, c_float_p, C.c_int]
        L.cho_create.argtypes = [C.c_int32, C.c_int32, C.c_int]
        L.cho_create.restype = C.c_void_p
        L.cho_destroy.argtypes = [C.c_void_p]
        L.cho_stage_count.argtypes = [C.c_void_p]
        L.cho_stage_info.argtypes = [C.c_void_p, C.c_int, c_int_p, c_int_p, c_int_p]
        L.cho_set_shift.argtypes = [C.c_void_p, C.c_int32]
        L.cho_process.argtypes = [C.c_void_p, c_float_p, C.c_int, c_float_p, C.POINTER(C.c_int8), C.c_int]
        L.cho_transmission_payload.argtypes = [C.c_uint64, C.c_int32, C.c_int32, C.POINTER(C.c_int8), C.c_int, C.POINTER(C.c_uint8), C.c_int]
    return _lib


def resampler_factors(sample_rate: int, bandwidth: int, threshold: int = 125, which: str = "oracle"):
    """getResamplersFactors as a list of (interpolation, decimation); which = "oracle" (restated) or "ref" (the reference's own code)."""
    a, b = (C.c_int * 16)(), (C.c_int * 16)()
    fn = lib().cho_resampler_factors if which == "oracle" else ref().ref_get_resamplers_factors
    n = fn(sample_rate, bandwidth, threshold, a, b, 16)
    return [(a[i], b[i]) for i in range(n)]


def design_taps(interp: int, decim: int) -> np.ndarray:
    n = lib().cho_design_taps(interp, decim, None, 0)
    t = np.zeros(n, np.float32)
    lib().cho_design_taps(interp, decim, fp(t), n)
    return t


class ChannelizerOracle:
    """One recording slot of the reference's Recorder, restated (channelizer_oracle.c)."""

    def __init__(self, sample_rate: int, bandwidth: int, threshold: int = 125):
        self._h = lib().cho_create(sample_rate, bandwidth, threshold)
        self.sample_rate, self.bandwidth = sample_rate, bandwidth
        n = lib().cho_stage_count(self._h)
        self.stages = []
        for s in range(n):
            i, d, t = C.c_int(), C.c_int(), C.c_int()
            lib().cho_stage_info(self._h, s, C.byref(i), C.byref(d), C.byref(t))
            self.stages.append((i.value, d.value, t.value))

    def set_shift(self, shift_hz: int):
        lib().cho_set_shift(self._h, int(shift_hz))

    def process(self, iq: np.ndarray):
        """iq: complex64 [n]. Returns (cf32 output, int8 [m, 2] output)."""
        x = np.ascontiguousarray(iq, dtype=np.complex64)
        cap = x.size + 16
        for i, d, _ in self.stages:
            cap = cap * i // d + 16
        out = np.zeros(cap, np.complex64)
        i8 = np.zeros((cap, 2), np.int8)
        m = lib().cho_process(self._h, x.view(np.float32).ctypes.data_as(c_float_p), x.size, out.view(np.float32).ctypes.data_as(c_float_p),
                              i8.ctypes.data_as(C.POINTER(C.c_int8)), cap)
        return out[:m], i8[:m]

    def close(self):
        if self._h:
            lib().cho_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def have_ref() -> bool:
    return os.path.exists(REF_LIB)


def ref() -> C.CDLL:
    global _ref
    if _ref is None:
        _ref = C.CDLL(REF_LIB)
        R = _ref
        R.ref_set_time.argtypes = [C.c_int64]
        R.ref_create.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, c_int32_p, C.c_int,
                                 C.c_int, C.c_int, C.c_int]
        R.ref_create.restype = C.c_void_p
        R.ref_destroy.argtypes = [C.c_void_p]
        R.ref_set_window.argtypes = [C.c_void_p, c_float_p]
        R.ref_set_range.argtypes = [C.c_void_p, C.c_int, C.c_int]
        R.ref_reset.argtypes = [C.c_void_p]
        R.ref_reset_noise.argtypes = [C.c_void_p]
        sig = [C.c_void_p, c_float_p, c_float_p, c_float_p, c_float_p, c_int32_p, c_int32_p, c_int32_p, c_int32_p,
               c_int32_p, c_int32_p]
        R.ref_process_spectrum.argtypes = sig
        R.ref_process_iq.argtypes = sig
        R.ref_ring_row.argtypes = [C.c_void_p, C.c_int, c_float_p]
        R.ref_noise.argtypes = [C.c_void_p, c_float_p]
        R.ref_average.argtypes = [c_float_p, c_float_p, C.c_int, C.c_int]
        R.ref_get_max_index.argtypes = [c_float_p, C.c_int, C.c_int, C.c_int]
        R.ref_get_fft.argtypes = [C.c_int, C.c_int]
        R.ref_get_tuned_frequency.argtypes = [C.c_int, C.c_int]
        R.ref_get_raw_file_name.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_int]
        R.ref_get_raw_file_name.restype = C.c_int
        R.ref_get_resamplers_factors.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
        R.ref_contains_with_margin.argtypes = [c_int32_p, C.c_int, C.c_int, C.c_int, c_int32_p]
        R.ref_most_frequent_value.argtypes = [c_int32_p, C.c_int]
        R.ref_psd.argtypes = [c_float_p, c_float_p, C.c_int, C.c_int]
        R.ref_averager_create.argtypes = [C.c_int, C.c_int]
        R.ref_averager_create.restype = C.c_void_p
        R.ref_averager_destroy.argtypes = [C.c_void_p]
        R.ref_averager_push.argtypes = [C.c_void_p, c_float_p]
        R.ref_averager_reset.argtypes = [C.c_void_p]
        R.ref_averager_average.argtypes = [C.c_void_p, c_float_p]
        R.ref_averager_row.argtypes = [C.c_void_p, C.c_int, c_float_p]
        R.orc_set_fft_backend.argtypes = [C.c_int]
        R.ref_spectrogram_create.argtypes = [C.c_int, C.c_int, C.c_int]
        R.ref_spectrogram_create.restype = C.c_void_p
        R.ref_spectrogram_destroy.argtypes = [C.c_void_p]
        R.ref_spectrogram_size.argtypes = [C.c_void_p]
        R.ref_spectrogram_set_frequency.argtypes = [C.c_void_p, C.c_int]
        R.ref_spectrogram_work.argtypes = [C.c_void_p, c_float_p, C.c_int]
        R.ref_spectrogram_container.argtypes = [C.c_void_p, c_float_p]
        R.ref_spectrogram_pop.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        R.ref_transmission_payload.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    return _ref


def fp(a):
    return a.ctypes.data_as(c_float_p)


def ip(a):
    return a.ctypes.data_as(c_int32_p)


def oracle_chain(sample_rate: int, center_hz: int, **overrides):
    """The oracle behind the same numpy wrapper the engine uses (prefix orc_)."""
    import rtl_sdr_scanner_cpp_amd as pkg
    return pkg.abi.Chain(lib(), "orc_", sample_rate, center_hz, **overrides)


class RefSpectrogram:
    """The reference's own Spectrogram block + DataController (oracle/_ref: radio/blocks/spectrogram.cpp and
    network/data_controller.cpp compiled in place; Mqtt is a stub that keeps what is published)."""

    def __init__(self, item_size: int, sample_rate: int, frequency: int):
        self._h = ref().ref_spectrogram_create(item_size, sample_rate, frequency)
        self.size = ref().ref_spectrogram_size(self._h)

    def close(self):
        if self._h:
            ref().ref_spectrogram_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_frequency(self, frequency: int):
        ref().ref_spectrogram_set_frequency(self._h, int(frequency))

    def work(self, psd_rows: np.ndarray, now_ms: int) -> int:
        """Spectrogram::work over the rows with getTime() == now_ms; returns the number of payloads waiting."""
        rows = np.ascontiguousarray(psd_rows, dtype=np.float32)
        ref().ref_set_time(int(now_ms))
        return ref().ref_spectrogram_work(self._h, fp(rows), rows.shape[0])

    def container(self):
        """(m_sum, m_counter) of the current centre frequency's container."""
        out = np.empty(self.size, np.float32)
        cnt = ref().ref_spectrogram_container(self._h, fp(out))
        return out, cnt

    def pop(self):
        """Oldest published payload as bytes, or None."""
        buf = np.empty(64 + self.size, np.uint8)
        n = ref().ref_spectrogram_pop(self._h, buf.ctypes.data, buf.size)
        return None if n <= 0 else buf[:n].tobytes()


def ref_transmission_payload(time_ms: int, frequency: int, sample_rate: int, iq_i8: np.ndarray) -> bytes:
    """DataController::pushTransmission's bytes for [n, 2] int8 samples (the reference's own code)."""
    a = np.ascontiguousarray(iq_i8, dtype=np.int8).reshape(-1, 2)
    out = np.empty(32 + a.size, np.uint8)
    n = ref().ref_transmission_payload(int(time_ms), int(frequency), int(sample_rate), a.ctypes.data, a.shape[0], out.ctypes.data, out.size)
    assert n > 0
    return out[:n].tobytes()


class RefChain:
    """The reference's own PSD / NoiseLearner / Transmission objects (oracle/_ref), one frame at a time."""

    def __init__(self, fft_size, sample_rate, range_lo, range_hi, start_level=8.0, stop_level=5.0, ignored=(),
                 group_size=None, min_time_ms=2000, timeout_ms=2000, tuning_step=2500, bandwidth=32000, window=None):
        R = ref()
        self.n = fft_size
        if group_size is None:  # indexStep, sdr_device.cpp:151
            group_size = int(np.ceil(bandwidth / (sample_rate / fft_size)))
        ig = np.ascontiguousarray(ignored, dtype=np.int32).reshape(-1)
        self._h = R.ref_create(fft_size, sample_rate, start_level, stop_level, range_lo, range_hi, ig.size // 2,
                               ip(ig) if ig.size else None, group_size, min_time_ms, timeout_ms, tuning_step)
        if window is not None:
            w = np.ascontiguousarray(window, dtype=np.float32)
            R.ref_set_window(self._h, fp(w))

    def close(self):
        if self._h:
            ref().ref_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def process(self, iq_cf32: np.ndarray, t_ms, spectrum: bool = False):
        """iq_cf32 [nframes, N] complex64 (already decimated to the first N of each item), t_ms per frame.
        Returns planes + per-frame candidate lists, notified transmissions and signal keys."""
        R = ref()
        n = self.n
        x = np.ascontiguousarray(iq_cf32, dtype=np.complex64)
        nf = x.shape[0]
        psd = np.empty((nf, n), np.float32)
        rel = np.empty((nf, n), np.float32)
        avg = np.empty((nf, n), np.float32)
        cand = np.empty(n, np.int32)
        tx = np.empty(2 * n, np.int32)
        sig = np.empty(n, np.int32)
        nc, nt, ns = C.c_int32(), C.c_int32(), C.c_int32()
        cands, txs, sigs = [], [], []
        fn = R.ref_process_spectrum if spectrum else R.ref_process_iq
        for f in range(nf):
            R.ref_set_time(int(t_ms[f]))
            fn(self._h, x[f].view(np.float32).ctypes.data_as(c_float_p), fp(psd[f]), fp(rel[f]), fp(avg[f]), ip(cand),
               C.byref(nc), ip(tx), C.byref(nt), ip(sig), C.byref(ns))
            cands.append(cand[:nc.value].copy())
            txs.append(tx[:2 * nt.value].reshape(-1, 2).copy())
            sigs.append(sig[:ns.value].copy())
        return {"psd": psd, "rel": rel, "avg": avg, "cands": cands, "tx": txs, "signals": sigs}

    def set_range(self, lo, hi):
        ref().ref_set_range(self._h, lo, hi)

    def reset(self):
        ref().ref_reset(self._h)

    def noise(self):
        thr = np.empty(self.n, np.float32)
        r = ref().ref_noise(self._h, fp(thr))
        return thr, r
