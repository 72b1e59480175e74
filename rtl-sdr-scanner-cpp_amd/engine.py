"""Host-side mirror of the reference's block interface for the scan path, over libspecscan.so.

``SpectrumEngine`` is one chain (what SdrDevice::setupChains builds per device, reference
sources/radio/sdr_device.cpp:148-168). Host-buffer calls go through ``process`` (numpy, like
``work()`` on scheduler-owned buffers); device-resident calls go through ``process_device`` with torch
tensors used purely as HBM allocations. There is no CPU fallback: if the HIP library is missing or no
GPU is present, construction raises."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import abi
from .build import LIB, LIB_DIAG

_libs = {}
_use_diag = False  # tests / measurement scripts: route new engines to libspecscan_diag.so (use_diag_library)


def use_diag_library(on=True) -> None:
    """Engines created from now on load csrc/libspecscan_diag.so — the same sources built with -DSS_DIAG, the only build
    that lets SS_* / SC_* environment variables pick one implementation of a step over another (A/B tests). A string is the
    path of an A/B build of the diagnostics library (build.build_variant) to load instead."""
    global _use_diag
    _use_diag = on if isinstance(on, str) else bool(on)


def load_library(diag: bool | None = None) -> C.CDLL:
    """dlopen csrc/libspecscan.so (built by build.build_lib / __graft_entry__.build). Raises if absent."""
    diag = _use_diag if diag is None else diag
    path = diag if isinstance(diag, str) else (LIB_DIAG if diag else LIB)
    if path not in _libs:
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `python __graft_entry__.py build` (hipcc, gfx950). "
                               "The spectral-scan engine has no CPU fallback.")
        lib = C.CDLL(path)
        abi.bind(lib, "ss_")
        lib.ss_device_count.restype = C.c_int
        lib.ss_process_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
        lib.ss_process_device.restype = C.c_int
        lib.ss_sync.argtypes = [C.c_void_p]
        lib.ss_sync.restype = C.c_int
        lib.ss_flush.argtypes = [C.c_void_p]
        lib.ss_flush.restype = C.c_int
        lib.ss_stream.argtypes = [C.c_void_p]
        lib.ss_stream.restype = C.c_void_p
        if hasattr(lib, "ss_get_stats"):  # (A/B builds of older trees, scripts/ab: measurement runs only)
            lib.ss_get_stats.argtypes = [C.c_void_p, C.POINTER(abi.SsStats)]
            lib.ss_get_stats.restype = C.c_int
            lib.ss_input_wait.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
            lib.ss_input_wait.restype = C.c_int
        lib.ss_kernel_timing.argtypes = [C.c_void_p, C.c_int]
        lib.ss_kernel_timing.restype = C.c_int
        lib.ss_kernel_timing_read.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32)]
        lib.ss_kernel_timing_read.restype = C.c_int
        lib.ss_kernel_timing_read_slots.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32)]
        lib.ss_kernel_timing_read_slots.restype = C.c_int
        lib.ss_kernel_timing_read_frames.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
        lib.ss_kernel_timing_read_frames.restype = C.c_int
        lib.ss_spectrogram_size.argtypes = [C.c_void_p]
        lib.ss_spectrogram_size.restype = C.c_int
        lib.ss_spectrogram_read.argtypes = [C.c_void_p, C.POINTER(C.c_int8), C.POINTER(C.c_float)]
        lib.ss_spectrogram_read.restype = C.c_int
        lib.ss_spectrogram_payload.argtypes = [C.c_uint64, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
        lib.ss_spectrogram_payload.restype = C.c_int
        lib.ss_selftest.argtypes = [C.c_int, C.c_int]
        lib.ss_selftest.restype = C.c_longlong
        lib.ss_feed_create.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
        lib.ss_feed_create.restype = C.c_int
        lib.ss_feed_destroy.argtypes = [C.c_void_p]
        lib.ss_feed_destroy.restype = None
        lib.ss_feed_acquire.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        lib.ss_feed_acquire.restype = C.c_int
        lib.ss_feed_submit.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.c_int64]
        lib.ss_feed_submit.restype = C.c_int
        lib.ss_feed_collect.argtypes = [C.c_void_p, C.POINTER(abi.SsFeedResult)]
        lib.ss_feed_collect.restype = C.c_int
        lib.ss_feed_pending.argtypes = [C.c_void_p]
        lib.ss_feed_pending.restype = C.c_int
        _libs[path] = lib
    return _libs[path]


EXPORTS = ("ss_default_config", "ss_device_count", "ss_create", "ss_destroy", "ss_last_error", "ss_process",
           "ss_process_device", "ss_flush", "ss_sync", "ss_stream", "ss_input_wait", "ss_get_stats", "ss_set_frequency_range", "ss_reset", "ss_reset_noise",
           "ss_read_window", "ss_read_noise", "ss_kernel_timing", "ss_kernel_timing_read", "ss_kernel_timing_read_slots", "ss_kernel_timing_read_frames", "ss_selftest", "ss_spectrogram_size", "ss_spectrogram_read",
           "ss_spectrogram_payload", "ss_feed_create", "ss_feed_destroy", "ss_feed_acquire", "ss_feed_submit", "ss_feed_collect", "ss_feed_pending")


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def spectrogram_payload(time_ms: int, frequency: int, sample_rate: int, row: np.ndarray) -> bytes:
    """DataController::pushSpectrogram's MQTT payload (reference sources/network/data_controller.cpp:44-57) for one int8 row."""
    lib = load_library()
    a = np.ascontiguousarray(row, dtype=np.int8).reshape(-1)
    n = lib.ss_spectrogram_payload(time_ms, frequency, sample_rate, None, a.size, None, 0)
    if n < 0:
        raise ValueError("spectrogram_payload: empty row")
    out = np.zeros(n, np.uint8)
    if lib.ss_spectrogram_payload(time_ms, frequency, sample_rate, a.ctypes.data, a.size, out.ctypes.data, n) != n:
        raise ValueError("spectrogram_payload")
    return out.tobytes()


class SpectrumEngine(abi.Chain):
    def __init__(self, sample_rate: int, center_hz: int, **overrides):
        lib = load_library()
        if lib.ss_device_count() <= 0:
            raise RuntimeError("no HIP device: the spectral-scan engine runs on an MI355X only (no CPU fallback)")
        super().__init__(lib, "ss_", sample_rate, center_hz, **overrides)

    @property
    def stream_handle(self) -> int:
        return int(self._lib.ss_stream(self._h) or 0)

    def process_device(self, iq, nframes: int, psd=None, rel=None, avg=None, cand_off=None, cand_idx=None, cand_avg=None):
        """All arguments are torch tensors resident on this chain's device (or None). Asynchronous: consecutive calls
        overlap on the library's own queues (include/specscan.h), so ``iq`` and every output must stay untouched — not
        refilled, not read — until ``sync()``; the next call's launch reads this call's last frames once more.
        ``flags=SS_FLAG_STREAM_ORDERED`` at construction gives the classic contract instead (all work on the chain's stream).
        iq must hold nframes items of N*D samples."""
        cap = 0 if cand_idx is None else int(cand_idx.numel())
        st = self._lib.ss_process_device(self._h, _ptr(iq), int(nframes), _ptr(psd), _ptr(rel), _ptr(avg), _ptr(cand_off),
                                         _ptr(cand_idx), _ptr(cand_avg), cap)
        self._check(st)

    def sync(self):
        """Drain the deferred stages of earlier process_device calls and wait for the chain's stream."""
        self._check(self._lib.ss_sync(self._h))

    def flush(self):
        """Enqueue the deferred stages of earlier process_device calls without waiting."""
        self._check(self._lib.ss_flush(self._h))

    def stats(self) -> dict:
        """ss_get_stats: what the library did so far (counters from creation; the device-side ones — tiles_tested, tiles_culled,
        wait_fallbacks — as far as the device has got: sync() first for exact figures). `state` is decoded into booleans."""
        st = abi.SsStats()
        st.size = C.sizeof(abi.SsStats)
        if hasattr(self._lib, "ss_get_stats"):
            self._check(self._lib.ss_get_stats(self._h, C.byref(st)))
        out = {k: int(getattr(st, k)) for k, _ in abi.SsStats._fields_ if k not in ("size", "state")}
        out.update(culling=bool(st.state & abi.SS_STATE_CULLING), overlap=bool(st.state & abi.SS_STATE_OVERLAP),
                   demoted=bool(st.state & abi.SS_STATE_DEMOTED), eager=bool(st.state & abi.SS_STATE_EAGER))
        return out

    def input_wait(self, stream_handle: int | None, calls_back: int):
        """ss_input_wait: `stream_handle` (a hipStream_t as an integer; None = the chain's own stream) waits until the input frames of
        every process_device call up to the one `calls_back` (>= 1) before the latest have been read for the last time."""
        self._check(self._lib.ss_input_wait(self._h, C.c_void_p(stream_handle) if stream_handle else None, int(calls_back)))

    def spectrogram_read(self):
        """(int8 spectrogram row, float means, frames accumulated) for the current centre; clears the accumulator."""
        size = self._lib.ss_spectrogram_size(self._h)
        out = np.zeros(size, np.int8)
        mean = np.zeros(size, np.float32)
        cnt = self._lib.ss_spectrogram_read(self._h, out.ctypes.data_as(C.POINTER(C.c_int8)), mean.ctypes.data_as(C.POINTER(C.c_float)))
        if cnt < 0:
            self._check(cnt)
        return out, mean, cnt

    def kernel_timing(self, every: int):
        """0 = off, 1 = time every launch of the FFT+PSD kernel, k > 1 = every k-th launch."""
        self._check(self._lib.ss_kernel_timing(self._h, int(every)))

    def kernel_timing_read(self):
        """(total device ms, launches) of the timed FFT+PSD launches since the last read."""
        ms, cnt = C.c_double(), C.c_int32()
        self._check(self._lib.ss_kernel_timing_read(self._h, C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    KERNEL_SLOTS = ("step", "rows", "sub", "plan")  # SS_KSLOT_* of include/specscan.h

    def kernel_timing_read_slots(self):
        """{slot: (total device ms, launches)} of the sampled calls since the last read, kernel by kernel."""
        ms, cnt = (C.c_double * 4)(), (C.c_int32 * 4)()
        self._check(self._lib.ss_kernel_timing_read_slots(self._h, ms, cnt))
        return {name: (ms[k], cnt[k]) for k, name in enumerate(self.KERNEL_SLOTS)}

    def kernel_timing_read_frames(self):
        """{slot: (total device ms, launches, frames those launches covered)} of the sampled launches since the last read — a call
        the library takes through in chunks has several launches per slot, each over its chunk's frames."""
        ms, cnt, fr = (C.c_double * 4)(), (C.c_int32 * 4)(), (C.c_int64 * 4)()
        self._check(self._lib.ss_kernel_timing_read_frames(self._h, ms, cnt, fr))
        return {name: (ms[k], cnt[k], fr[k]) for k, name in enumerate(self.KERNEL_SLOTS)}

    def feed(self, depth: int = 3, cand_cap: int = 1 << 20, want_psd: bool = False) -> "Feed":
        """Pipelined host feeding (ss_feed_*): pinned staging slots, H2D overlapped with the chain."""
        return Feed(self, depth, cand_cap, want_psd)


class Feed:
    """ss_feed_* of include/specscan.h: ``acquire()`` a pinned numpy view, fill it, ``submit(nframes)``,
    ``collect()`` results in submission order. Views returned by ``collect`` live in the feed's pinned memory and
    are valid until the slot is acquired again (copy what must outlive that)."""

    def __init__(self, engine: SpectrumEngine, depth: int, cand_cap: int, want_psd: bool):
        self._e, self._lib = engine, engine._lib
        self.cand_cap = int(cand_cap)
        h = C.c_void_p()
        engine._check(self._lib.ss_feed_create(engine._h, int(depth), int(cand_cap), int(bool(want_psd)), C.byref(h)))
        self._h = h
        cfg = engine.cfg
        self._shape = (cfg.max_batch, cfg.fft_size) if cfg.in_format == abi.SS_FMT_CF32 else (cfg.max_batch, cfg.fft_size, 2)
        self._dtype = {abi.SS_FMT_CF32: np.complex64, abi.SS_FMT_CS8: np.int8, abi.SS_FMT_CU8: np.uint8}[cfg.in_format]

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ss_feed_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def pending(self) -> int:
        return int(self._lib.ss_feed_pending(self._h))

    def acquire(self) -> np.ndarray:
        """[max_batch, N] complex64 (or [max_batch, N, 2] int8/uint8) view of the next free pinned slot: frames
        already decimated (the first N samples of each N*D item)."""
        p = C.c_void_p()
        self._e._check(self._lib.ss_feed_acquire(self._h, C.byref(p)))
        nbytes = int(np.prod(self._shape)) * np.dtype(self._dtype).itemsize
        buf = (C.c_char * nbytes).from_address(p.value)
        return np.frombuffer(buf, dtype=self._dtype).reshape(self._shape)

    def submit(self, nframes: int, t_ms=None, tag: int = 0):
        tp = None
        if t_ms is not None:
            t = np.ascontiguousarray(t_ms, dtype=np.int64)
            tp = t.ctypes.data_as(C.POINTER(C.c_int64))
        self._e._check(self._lib.ss_feed_submit(self._h, int(nframes), tp, int(tag)))

    def collect(self) -> dict:
        r = abi.SsFeedResult()
        self._e._check(self._lib.ss_feed_collect(self._h, C.byref(r)))
        nf = r.nframes
        off = np.ctypeslib.as_array(r.cand_off, shape=(nf + 1,))
        total = min(int(off[nf]), self.cand_cap)
        out = {"nframes": nf, "status": r.status, "tag": r.user_tag, "cand_off": off,
               "cand_idx": np.ctypeslib.as_array(r.cand_idx, shape=(max(total, 1),))[:total],
               "cand_avg": np.ctypeslib.as_array(r.cand_avg, shape=(max(total, 1),))[:total]}
        if r.psd_db:
            out["psd"] = np.ctypeslib.as_array(r.psd_db, shape=(nf, self._e.n))
        return out
