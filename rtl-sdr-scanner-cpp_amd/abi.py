"""ctypes view of include/specscan.h: the ss_config struct, status codes and a generic binder that
works for any library exporting the boundary's function set under a prefix (``ss_`` for the HIP
engine; the CPU oracle under oracle/ exports the same set as ``orc_`` and is bound by the tests)."""
from __future__ import annotations

import ctypes as C

import numpy as np

SS_ABI_VERSION = 3
SS_OK, SS_ERR_INVALID, SS_ERR_NO_DEVICE, SS_ERR_HIP, SS_ERR_BATCH, SS_ERR_CAND_OVERFLOW, SS_ERR_NOMEM = 0, -1, -2, -3, -4, -5, -6
SS_FMT_CF32, SS_FMT_CS8, SS_FMT_CU8 = 0, 1, 2
SS_PLANE_PSD, SS_PLANE_REL, SS_PLANE_AVG = 0, 1, 2
SS_FLAG_KEEP_PLANES = 1
SS_FLAG_SPECTROGRAM = 2
SS_FLAG_NO_CULL = 4
SS_FLAG_STREAM_ORDERED = 8
SS_FLAG_REFERENCE_NAN = 16
SS_STATE_CULLING, SS_STATE_OVERLAP, SS_STATE_DEMOTED, SS_STATE_EAGER = 1, 2, 4, 8
SS_NO_DATA = np.float32(-100.0)

c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)
c_int64_p = C.POINTER(C.c_int64)


class SsConfig(C.Structure):
    """struct ss_config (include/specscan.h)."""
    _fields_ = [
        ("abi_version", C.c_int32), ("fft_size", C.c_int32), ("sample_rate", C.c_int32), ("decim", C.c_int32),
        ("in_format", C.c_int32), ("int_scale", C.c_float), ("window", c_float_p),
        ("grouping_x", C.c_int32), ("grouping_y", C.c_int32), ("start_level", C.c_float),
        ("range_lo", C.c_int32), ("range_hi", C.c_int32), ("n_ignored", C.c_int32), ("ignored", c_int32_p),
        ("learn_frames", C.c_int32), ("learn_ms", C.c_int32), ("max_batch", C.c_int32), ("device_id", C.c_int32),
        ("flags", C.c_uint32),
    ]


class SsStats(C.Structure):
    """struct ss_stats (include/specscan.h)."""
    _fields_ = [("size", C.c_uint32), ("state", C.c_uint32)] + [(k, C.c_uint64) for k in (
        "calls", "calls_overlapped", "calls_in_order", "drains", "demotions", "tiles_total", "tiles_tested", "tiles_culled", "wait_fallbacks")]


class SsFeedResult(C.Structure):  # ss_feed_result, include/specscan.h
    _fields_ = [("nframes", C.c_int32), ("status", C.c_int32), ("user_tag", C.c_int64), ("cand_off", C.POINTER(C.c_int32)),
                ("cand_idx", C.POINTER(C.c_int32)), ("cand_avg", C.POINTER(C.c_float)), ("psd_db", C.POINTER(C.c_float))]


class SpecscanError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"specscan status {status}: {message}")
        self.status = status


def _fp(a):
    return None if a is None else a.ctypes.data_as(c_float_p)


def _ip(a):
    return None if a is None else a.ctypes.data_as(c_int32_p)


def bind(lib: C.CDLL, prefix: str) -> None:
    """Declare argtypes/restype of the boundary's function set on `lib`."""
    f = lambda name: getattr(lib, prefix + name)  # noqa: E731
    f("default_config").argtypes = [C.POINTER(SsConfig), C.c_int32, C.c_int32]
    f("default_config").restype = None
    f("create").argtypes = [C.POINTER(SsConfig), C.POINTER(C.c_void_p)]
    f("create").restype = C.c_int
    f("destroy").argtypes = [C.c_void_p]
    f("destroy").restype = None
    f("last_error").argtypes = [C.c_void_p]
    f("last_error").restype = C.c_char_p
    f("process").argtypes = [C.c_void_p, C.c_void_p, C.c_int32, c_int64_p, c_float_p, c_float_p, c_float_p,
                             c_int32_p, c_int32_p, c_float_p, C.c_int32]
    f("process").restype = C.c_int
    f("set_frequency_range").argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    f("set_frequency_range").restype = C.c_int
    f("reset").argtypes = [C.c_void_p]
    f("reset").restype = C.c_int
    f("reset_noise").argtypes = [C.c_void_p]
    f("reset_noise").restype = C.c_int
    f("read_window").argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, c_float_p]
    f("read_window").restype = C.c_int
    f("read_noise").argtypes = [C.c_void_p, c_float_p]
    f("read_noise").restype = C.c_int


def apply_overrides(cfg, overrides: dict) -> list:
    """Set ss_config fields from keyword overrides; returns the arrays the config points into (keep them alive)."""
    keep = []
    for k, v in overrides.items():
        if k == "window" and v is not None:
            w = np.ascontiguousarray(v, dtype=np.float32)
            keep.append(w)
            cfg.window = _fp(w)
        elif k == "ignored":
            ig = np.ascontiguousarray(v, dtype=np.int32).reshape(-1)
            keep.append(ig)
            cfg.ignored = _ip(ig)
            cfg.n_ignored = ig.size // 2
        else:
            if not hasattr(cfg, k):
                raise TypeError(f"unknown ss_config field {k}")
            setattr(cfg, k, v)
    return keep


class Chain:
    """One scan chain behind the C ABI (host-buffer entry points), numpy in / numpy out.

    Mirrors the reference's block interface for this path: construct once per device
    (SdrDevice::setupChains, sources/radio/sdr_device.cpp:148-168), call ``process`` as the scheduler
    calls ``work()``, ``set_frequency_range``/``reset`` as the Scanner thread does on a retune
    (sources/radio/sdr_device.cpp:54-80)."""

    def __init__(self, lib: C.CDLL, prefix: str, sample_rate: int, center_hz: int, **overrides):
        self._lib, self._p = lib, prefix
        bind(lib, prefix)
        cfg = SsConfig()
        self._f("default_config")(C.byref(cfg), int(sample_rate), int(center_hz))
        self._keep = apply_overrides(cfg, overrides)
        self.cfg = cfg
        h = C.c_void_p()
        st = self._f("create")(C.byref(cfg), C.byref(h))
        if st != SS_OK:
            raise SpecscanError(st, (self._f("last_error")(None) or b"").decode())
        self._h = h
        self.n = cfg.fft_size
        self.item = cfg.fft_size * cfg.decim

    def _f(self, name):
        return getattr(self._lib, self._p + name)

    def _check(self, st, allow=()):
        if st != SS_OK and st not in allow:
            raise SpecscanError(st, (self._f("last_error")(self._h) or b"").decode())
        return st

    def close(self):
        if getattr(self, "_h", None):
            self._f("destroy")(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _iq_bytes_per_item(self):
        return self.item * (8 if self.cfg.in_format == SS_FMT_CF32 else 2)

    def process(self, iq: np.ndarray, t_ms=None, want=("psd", "rel", "avg"), cand_cap=None):
        """iq: [nframes, N*D] complex64, or [nframes, N*D, 2] int8/uint8. Returns a dict with the
        requested planes [nframes, N], ``cand_off`` [nframes+1], ``cand_idx``, ``cand_avg``."""
        iq = np.ascontiguousarray(iq)
        nframes = 0 if iq.size == 0 else iq.shape[0]
        if nframes and iq.nbytes != nframes * self._iq_bytes_per_item():
            raise ValueError("iq has the wrong shape for this chain")
        n = self.n
        out = {}
        planes = {}
        for name in ("psd", "rel", "avg"):
            planes[name] = np.empty((nframes, n), dtype=np.float32) if name in want else None
        if cand_cap is None:
            cand_cap = nframes * n
        off = np.zeros(nframes + 1, dtype=np.int32)
        # the candidate arrays are kept from call to call (grow-only): with the default capacity — every bin of every frame — they are
        # two 64 MiB arrays per 16-frame call of 2^20 points, and mapping and unmapping those around every call made the runtime's copy
        # of the NEXT call's input take 19-29 ms instead of 2.4 (profiles/r06/s10_summary.txt)
        if getattr(self, "_cand_bufs", None) is None or self._cand_bufs[0].size < max(cand_cap, 1):
            self._cand_bufs = (np.empty(max(cand_cap, 1), dtype=np.int32), np.empty(max(cand_cap, 1), dtype=np.float32))
        idx, cav = self._cand_bufs
        t = None
        if t_ms is not None:
            t = np.ascontiguousarray(t_ms, dtype=np.int64)
            if t.size != nframes:
                raise ValueError("t_ms must have one entry per frame")
        st = self._f("process")(self._h, iq.ctypes.data_as(C.c_void_p) if nframes else None, nframes,
                                None if t is None else t.ctypes.data_as(c_int64_p),
                                _fp(planes["psd"]), _fp(planes["rel"]), _fp(planes["avg"]),
                                _ip(off), _ip(idx), _fp(cav), int(cand_cap))
        self._check(st, allow=(SS_ERR_CAND_OVERFLOW,))
        out.update({k: v for k, v in planes.items() if v is not None})
        total = min(int(off[-1]), cand_cap)
        out["status"] = st
        out["cand_off"] = off
        out["cand_idx"] = idx[:total].copy()
        out["cand_avg"] = cav[:total].copy()
        return out

    def set_frequency_range(self, lo: int, hi: int):
        self._check(self._f("set_frequency_range")(self._h, int(lo), int(hi)))

    def reset(self):
        self._check(self._f("reset")(self._h))

    def reset_noise(self):
        self._check(self._f("reset_noise")(self._h))

    def read_window(self, plane: int, frame: int, lo: int, hi: int) -> np.ndarray:
        out = np.empty(hi - lo, dtype=np.float32)
        self._check(self._f("read_window")(self._h, plane, frame, lo, hi, _fp(out)))
        return out

    def read_noise(self):
        thr = np.empty(self.n, dtype=np.float32)
        r = self._f("read_noise")(self._h, _fp(thr))
        if r < 0:
            self._check(r)
        return thr, bool(r)
