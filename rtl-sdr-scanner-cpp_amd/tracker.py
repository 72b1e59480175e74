"""ctypes binding of host/libspecscan_host.so — the host-side signal tracker (host/signal_tracker.h), the part of
the reference's Transmission block that turns per-frame candidates into the Scanner's (shift Hz, flush) list."""
from __future__ import annotations

import ctypes as C
import math
import os

import numpy as np

from .build import HOST_LIB, build_host_lib

_lib = None
c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)


def load_host_library() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(HOST_LIB):
            build_host_lib()
        lib = C.CDLL(HOST_LIB)
        lib.sst_create.argtypes = [C.c_int, C.c_int32, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int32]
        lib.sst_create.restype = C.c_void_p
        lib.sst_destroy.argtypes = [C.c_void_p]
        lib.sst_reset.argtypes = [C.c_void_p]
        lib.sst_process_frame.argtypes = [C.c_void_p, C.c_int64, c_float_p, c_float_p, c_int32_p, C.c_int, c_int32_p, C.c_int, c_int32_p,
                                          C.c_int, C.POINTER(C.c_int)]
        lib.sst_process_frame.restype = C.c_int
        _lib = lib
    return _lib


def index_step(bandwidth_hz: int, sample_rate: int, fft_size: int) -> int:
    """indexStep of SdrDevice::setupChains (reference sources/radio/sdr_device.cpp:151)."""
    return int(math.ceil(bandwidth_hz / (sample_rate / fft_size)))


class SignalTracker:
    def __init__(self, fft_size, sample_rate, start_level=8.0, stop_level=5.0, group_size=None, grouping_y=21, min_time_ms=2000,
                 timeout_ms=2000, tuning_step=2500, bandwidth=32000):
        self._lib = load_host_library()
        if group_size is None:
            group_size = index_step(bandwidth, sample_rate, fft_size)
        self.n = fft_size
        self._h = self._lib.sst_create(fft_size, sample_rate, start_level, stop_level, group_size, grouping_y, min_time_ms, timeout_ms, tuning_step)
        if not self._h:
            raise ValueError("bad tracker configuration")
        self._tx = np.empty(2 * fft_size, np.int32)
        self._sig = np.empty(fft_size, np.int32)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.sst_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        self._lib.sst_reset(self._h)

    def process_frame(self, now_ms: int, avg_row: np.ndarray, rel_row: np.ndarray, candidates: np.ndarray):
        """Returns (transmissions [k, 2] = (shift Hz, flush), tracked signal keys)."""
        a = np.ascontiguousarray(avg_row, np.float32)
        r = np.ascontiguousarray(rel_row, np.float32)
        c = np.ascontiguousarray(candidates, np.int32)
        nsig = C.c_int()
        ntx = self._lib.sst_process_frame(self._h, int(now_ms), a.ctypes.data_as(c_float_p), r.ctypes.data_as(c_float_p),
                                          c.ctypes.data_as(c_int32_p), c.size, self._tx.ctypes.data_as(c_int32_p), self.n,
                                          self._sig.ctypes.data_as(c_int32_p), self.n, C.byref(nsig))
        return self._tx[:2 * ntx].reshape(-1, 2).copy(), self._sig[:nsig.value].copy()

    def process_batch(self, t_ms, avg, rel, cand_off, cand_idx):
        out = []
        for f in range(avg.shape[0]):
            out.append(self.process_frame(t_ms[f], avg[f], rel[f], cand_idx[cand_off[f]:cand_off[f + 1]]))
        return out
