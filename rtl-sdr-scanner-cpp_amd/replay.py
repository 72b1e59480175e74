"""Replay front end: the reference's raw dump files through the drop-in boundary.

* names: ``make_raw_file_name`` / ``parse_raw_file_name`` — getRawFileName (reference sources/utils/radio_utils.cpp:78-84)
  and its inverse, with the field convention of the reference's own reader (scripts/converter.py:58-59);
* ``RawFileSink`` — FileSink<T>(itemSize, flushable=false) (sources/radio/blocks/file_sink.h): what the reference
  attaches to its source (``_fc.raw``) and to PSD (``_power.raw``) at sources/radio/sdr_device.cpp:173-181;
* ``RawIqReader`` — re-frames a ``_fc.raw`` / ``.cs8`` / ``.cu8`` dump into the frames the chain takes;
* ``replay_file`` — a reader thread fills the engine's pinned slots while earlier batches cross PCIe and run
  (ss_feed_*, include/specscan.h); yields the per-batch results in order, and the end-to-end (file + PCIe inclusive)
  rate in ``ReplayStats``.

The naming, sink and reader are host/raw_file.h (C++, libspecscan_host.so); this module binds them with ctypes."""
from __future__ import annotations

import ctypes as C
import dataclasses
import queue
import threading
import time

import numpy as np

from . import abi
from .tracker import load_host_library

KIND_CF32, KIND_CS8, KIND_CU8, KIND_F32 = 0, 1, 2, 3
CS8_FILE_SCALE = 1.0 / 127.5  # converter.py:33: np.int8 -> complex64 / 127.5


def _lib():
    lib = load_host_library()
    if not getattr(lib, "_srf_bound", False):
        lib.srf_make_name.argtypes = [C.c_char_p, C.c_char_p, C.c_int32, C.c_int32] + [C.c_int] * 6 + [C.c_char_p, C.c_int]
        lib.srf_make_name.restype = C.c_int
        lib.srf_parse_name.argtypes = [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                       C.c_char_p, C.c_char_p]
        lib.srf_parse_name.restype = C.c_int
        lib.srf_sink_create.argtypes = [C.c_int64]
        lib.srf_sink_create.restype = C.c_void_p
        lib.srf_sink_destroy.argtypes = [C.c_void_p]
        lib.srf_sink_start.argtypes = [C.c_void_p, C.c_char_p]
        lib.srf_sink_stop.argtypes = [C.c_void_p]
        lib.srf_sink_work.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        lib.srf_sink_work.restype = C.c_int
        lib.srf_reader_open.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int]
        lib.srf_reader_open.restype = C.c_void_p
        lib.srf_reader_close.argtypes = [C.c_void_p]
        lib.srf_reader_items.argtypes = [C.c_void_p]
        lib.srf_reader_items.restype = C.c_int64
        lib.srf_reader_read_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        lib.srf_reader_read_frames.restype = C.c_int
        lib.srf_reader_read_frames_parallel.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        lib.srf_reader_read_frames_parallel.restype = C.c_int
        lib._srf_bound = True
    return lib


def make_raw_file_name(label: str, extension: str, frequency: int, sample_rate: int, when: time.struct_time | None = None) -> str:
    """``./<label>_YYYYMMDD_HHMMSS_<frequency>_<sample_rate>_<extension>.raw`` (local time, like the reference)."""
    t = when or time.localtime()
    buf = C.create_string_buffer(1024)
    n = _lib().srf_make_name(label.encode(), extension.encode(), int(frequency), int(sample_rate), t.tm_year, t.tm_mon, t.tm_mday, t.tm_hour,
                             t.tm_min, t.tm_sec, buf, 1024)
    if n < 0:
        raise ValueError("raw file name does not fit")
    return buf.value.decode()


@dataclasses.dataclass
class RawFileInfo:
    label: str
    extension: str
    frequency: int
    sample_rate: int
    kind: int
    timestamp: tuple


def parse_raw_file_name(path: str) -> RawFileInfo:
    freq, rate, kind = C.c_int32(), C.c_int32(), C.c_int()
    ymd = (C.c_int * 6)()
    label, ext = C.create_string_buffer(64), C.create_string_buffer(64)
    if _lib().srf_parse_name(path.encode(), C.byref(freq), C.byref(rate), C.byref(kind), ymd, label, ext) != 0:
        raise ValueError(f"not a raw dump name: {path}")
    return RawFileInfo(label.value.decode(), ext.value.decode(), freq.value, rate.value, kind.value, tuple(ymd))


class RawFileSink:
    """Items are written only between ``start_recording`` and ``stop_recording``; the file appears at the first item."""

    def __init__(self, item_bytes: int):
        self._lib = _lib()
        self._h = self._lib.srf_sink_create(int(item_bytes))
        self._item = int(item_bytes)
        if not self._h:
            raise ValueError("item_bytes must be positive")

    def start_recording(self, filename: str):
        self._lib.srf_sink_start(self._h, filename.encode())

    def stop_recording(self):
        self._lib.srf_sink_stop(self._h)

    def work(self, items: np.ndarray) -> int:
        a = np.ascontiguousarray(items)
        if a.nbytes % self._item:
            raise ValueError("not a whole number of items")
        n = self._lib.srf_sink_work(self._h, a.ctypes.data, a.nbytes // self._item)
        if n < 0:
            raise OSError("raw file sink: open/write failed")
        return n

    def close(self):
        if self._h:
            self._lib.srf_sink_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RawIqReader:
    def __init__(self, path: str, kind: int, fft_size: int, decim: int = 1):
        self._lib = _lib()
        self._h = self._lib.srf_reader_open(path.encode(), int(kind), int(fft_size), int(decim))
        if not self._h:
            raise OSError(f"cannot open {path} as IQ kind {kind}")
        self.kind, self.n = kind, fft_size
        self.items = int(self._lib.srf_reader_items(self._h))

    def read_into(self, frames: np.ndarray, max_frames: int, threads: int = 1) -> int:
        """Fills ``frames`` (C-contiguous, at least max_frames frames) with the next frames; returns how many (0 = end).
        threads > 1 spreads the copy out of the page cache over that many readers."""
        n = self._lib.srf_reader_read_frames_parallel(self._h, frames.ctypes.data, int(max_frames), int(threads))
        if n < 0:
            raise OSError("raw IQ read failed")
        return n

    def read(self, max_frames: int) -> np.ndarray:
        out = np.empty((max_frames, self.n), np.complex64) if self.kind == KIND_CF32 else \
            np.empty((max_frames, self.n, 2), np.int8 if self.kind == KIND_CS8 else np.uint8)
        return out[: self.read_into(out, max_frames)]

    def close(self):
        if self._h:
            self._lib.srf_reader_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@dataclasses.dataclass
class ReplayStats:
    frames: int = 0
    batches: int = 0
    candidates: int = 0
    bytes_read: int = 0
    seconds: float = 0.0
    read_seconds: float = 0.0  # time the reader thread spent inside file reads

    def msamples_per_sec(self, fft_size: int) -> float:
        return self.frames * fft_size / self.seconds / 1e6 if self.seconds > 0 else 0.0


def engine_overrides_for(info: RawFileInfo) -> dict:
    """ss_config fields a dump's name implies (format and the int8 scale converter.py uses)."""
    if info.kind == KIND_CS8:
        return {"in_format": abi.SS_FMT_CS8, "int_scale": CS8_FILE_SCALE}
    if info.kind == KIND_CU8:
        return {"in_format": abi.SS_FMT_CU8, "int_scale": CS8_FILE_SCALE}
    if info.kind == KIND_CF32:
        return {"in_format": abi.SS_FMT_CF32}
    raise ValueError("a _power.raw file holds spectra, not IQ")


def replay_file(engine, path: str, kind: int | None = None, batch: int | None = None, depth: int = 3, cand_cap: int = 1 << 20,
                want_psd: bool = False, frame_period_ms: float | None = None, stats: ReplayStats | None = None, read_threads: int = 4):
    """Generator: streams the dump through ``engine`` (a SpectrumEngine whose in_format matches the file) and yields
    one result dict per batch, in order (copies: safe to keep). A reader thread fills pinned slots; up to ``depth - 1``
    batches are in flight behind the one being read.

    frame_period_ms: when given, frame k carries the timestamp round(k * period) — the clock the NoiseLearner's
    2000 ms window runs on (noise_learner.cpp:23); otherwise learning counts ``learn_frames``."""
    cfg = engine.cfg
    if kind is None:
        kind = parse_raw_file_name(path).kind
    want_fmt = {KIND_CF32: abi.SS_FMT_CF32, KIND_CS8: abi.SS_FMT_CS8, KIND_CU8: abi.SS_FMT_CU8}[kind]
    if cfg.in_format != want_fmt:
        raise ValueError("engine in_format does not match the file (see engine_overrides_for)")
    batch = int(batch or cfg.max_batch)
    if not 0 < batch <= cfg.max_batch:
        raise ValueError("batch must be in 1..max_batch")
    reader = RawIqReader(path, kind, cfg.fft_size, cfg.decim)
    feed = engine.feed(depth=depth, cand_cap=cand_cap, want_psd=want_psd)
    st = stats if stats is not None else ReplayStats()
    frame_bytes = cfg.fft_size * (8 if kind == KIND_CF32 else 2)
    submitted: queue.Queue = queue.Queue()
    free = threading.Semaphore(depth)  # a slot is free again once its batch has been collected
    err: list = []

    def produce():
        first = 0
        try:
            while True:
                free.acquire()
                buf = feed.acquire()
                t0 = time.perf_counter()
                got = reader.read_into(buf, batch, read_threads)
                st.read_seconds += time.perf_counter() - t0
                if got == 0:
                    break
                t_ms = None
                if frame_period_ms is not None:
                    t_ms = np.round((first + np.arange(got)) * frame_period_ms).astype(np.int64)
                feed.submit(got, t_ms, tag=first)
                st.bytes_read += got * frame_bytes
                submitted.put(got)
                first += got
        except Exception as e:  # surfaced by the consumer
            err.append(e)
        finally:
            submitted.put(None)

    t_start = time.perf_counter()
    th = threading.Thread(target=produce, name="specscan-replay-reader", daemon=True)
    th.start()
    try:
        while True:
            got = submitted.get()
            if got is None:
                break
            r = feed.collect()
            out = {k: (np.array(v) if isinstance(v, np.ndarray) else v) for k, v in r.items()}
            out["first_frame"] = out.pop("tag")
            free.release()
            st.frames += out["nframes"]
            st.batches += 1
            st.candidates += int(out["cand_off"][-1])
            st.seconds = time.perf_counter() - t_start
            yield out
        if err:
            raise err[0]
    finally:
        for _ in range(depth):
            free.release()  # unblock the reader if the consumer stopped early
        th.join(timeout=10)
        st.seconds = time.perf_counter() - t_start
        feed.close()
        reader.close()
