"""ctypes view of include/specscan_channelizer.h — the recorder channeliser: for every recording slot
of a device, rotator -> cascaded rational resamplers -> int8, what the reference's Recorder builds per slot out of GNU
Radio blocks (reference sources/radio/recorder.cpp:14-46). Test and tooling plumbing; the compute is csrc/channelizer.hip."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .abi import SpecscanError
from .engine import load_library

SC_ABI_VERSION = 1
SC_MAX_CHANNELS = 16

EXPORTS = ("sc_default_config", "sc_create", "sc_destroy", "sc_last_error", "sc_stage_count", "sc_stage_info", "sc_stage_taps",
           "sc_output_capacity", "sc_start", "sc_stop", "sc_is_recording", "sc_process", "sc_process_device", "sc_sync",
           "sc_transmission_payload")


class ScConfig(C.Structure):  # sc_config
    _fields_ = [("abi_version", C.c_uint32), ("sample_rate", C.c_int32), ("bandwidth", C.c_int32), ("threshold", C.c_int32),
                ("channels", C.c_int32), ("max_samples", C.c_int32), ("pack_scale", C.c_float), ("device_id", C.c_int32)]


def _bind(lib):
    if getattr(lib, "_sc_bound", False):
        return lib
    i32p = C.POINTER(C.c_int32)
    lib.sc_default_config.argtypes = [C.POINTER(ScConfig), C.c_int32, C.c_int32]
    lib.sc_default_config.restype = None
    lib.sc_create.argtypes = [C.POINTER(ScConfig), C.POINTER(C.c_void_p)]
    lib.sc_destroy.argtypes = [C.c_void_p]
    lib.sc_destroy.restype = None
    lib.sc_last_error.argtypes = [C.c_void_p]
    lib.sc_last_error.restype = C.c_char_p
    lib.sc_stage_count.argtypes = [C.c_void_p]
    lib.sc_stage_info.argtypes = [C.c_void_p, C.c_int32, i32p, i32p, i32p]
    lib.sc_stage_taps.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_float)]
    lib.sc_output_capacity.argtypes = [C.c_void_p, C.c_int32]
    lib.sc_output_capacity.restype = C.c_int32
    lib.sc_start.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    lib.sc_stop.argtypes = [C.c_void_p, C.c_int32]
    lib.sc_is_recording.argtypes = [C.c_void_p, C.c_int32]
    lib.sc_process.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, i32p, C.c_int32]
    lib.sc_process_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, i32p, C.c_int32]
    lib.sc_sync.argtypes = [C.c_void_p]
    lib.sc_transmission_payload.argtypes = [C.c_uint64, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
    lib._sc_bound = True
    return lib


def transmission_payload(time_ms: int, frequency: int, sample_rate: int, iq_i8: np.ndarray) -> bytes:
    """DataController::pushTransmission's MQTT payload (reference sources/network/data_controller.cpp:27-42)."""
    lib = _bind(load_library())
    a = np.ascontiguousarray(iq_i8, dtype=np.int8).reshape(-1, 2)
    n = lib.sc_transmission_payload(time_ms, frequency, sample_rate, None, a.shape[0], None, 0)
    out = np.zeros(n, np.uint8)
    got = lib.sc_transmission_payload(time_ms, frequency, sample_rate, a.ctypes.data, a.shape[0], out.ctypes.data, n)
    if got != n:
        raise ValueError("payload")
    return out.tobytes()


class Channelizer:
    """All recording slots of one device (Recorder x recordersCount, reference sources/radio/sdr_device.cpp:39-41)."""

    def __init__(self, sample_rate: int, bandwidth: int, **overrides):
        self._lib = _bind(load_library())
        cfg = ScConfig()
        self._lib.sc_default_config(C.byref(cfg), int(sample_rate), int(bandwidth))
        for k, v in overrides.items():
            if not hasattr(cfg, k):
                raise TypeError(f"unknown sc_config field {k}")
            setattr(cfg, k, v)
        self.cfg = cfg
        h = C.c_void_p()
        st = self._lib.sc_create(C.byref(cfg), C.byref(h))
        if st != 0:
            raise SpecscanError(st, (self._lib.sc_last_error(None) or b"").decode())
        self._h = h
        self.stages = []
        for s in range(self._lib.sc_stage_count(h)):
            i, d, t = C.c_int32(), C.c_int32(), C.c_int32()
            self._lib.sc_stage_info(h, s, C.byref(i), C.byref(d), C.byref(t))
            self.stages.append((i.value, d.value, t.value))

    def _check(self, st):
        if st != 0:
            raise SpecscanError(st, (self._lib.sc_last_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_h", None):
            self._lib.sc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def stage_taps(self, stage: int) -> np.ndarray:
        t = np.zeros(self.stages[stage][2], np.float32)
        self._check(self._lib.sc_stage_taps(self._h, stage, t.ctypes.data_as(C.POINTER(C.c_float))))
        return t

    def output_capacity(self, nsamples: int) -> int:
        return int(self._lib.sc_output_capacity(self._h, int(nsamples)))

    def start(self, channel: int, shift_hz: int):
        self._check(self._lib.sc_start(self._h, int(channel), int(shift_hz)))

    def stop(self, channel: int):
        self._check(self._lib.sc_stop(self._h, int(channel)))

    def is_recording(self, channel: int) -> bool:
        return bool(self._lib.sc_is_recording(self._h, int(channel)))

    def process(self, iq: np.ndarray, want_cf32: bool = True):
        """iq: complex64 [n] (the device stream). Returns {channel: (int8 [m, 2], complex64 [m] or None)} for the active slots."""
        x = np.ascontiguousarray(iq, dtype=np.complex64)
        cap = max(self.output_capacity(x.size), 1)
        nch = self.cfg.channels
        i8 = np.zeros((nch, cap, 2), np.int8)
        cf = np.zeros((nch, cap), np.complex64) if want_cf32 else None
        counts = np.zeros(nch, np.int32)
        self._check(self._lib.sc_process(self._h, x.ctypes.data, x.size, i8.ctypes.data, cf.ctypes.data if want_cf32 else None,
                                         counts.ctypes.data_as(C.POINTER(C.c_int32)), cap))
        return {ch: (i8[ch, :counts[ch]].copy(), cf[ch, :counts[ch]].copy() if want_cf32 else None)
                for ch in range(nch) if self.is_recording(ch)}

    def process_device(self, iq, nsamples: int, out_i8=None, out_cf32=None, cap: int = 0):
        """torch tensors on this context's device (plain HBM allocations). Returns the per-channel counts (numpy); async: call sync()."""
        counts = np.zeros(self.cfg.channels, np.int32)
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())  # noqa: E731
        self._check(self._lib.sc_process_device(self._h, p(iq), int(nsamples), p(out_i8), p(out_cf32),
                                                counts.ctypes.data_as(C.POINTER(C.c_int32)), int(cap)))
        return counts

    def sync(self):
        self._check(self._lib.sc_sync(self._h))
