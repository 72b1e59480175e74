"""Raw dump files of the reference: names, sink, reader. CPU only.
The name is pinned against the reference's own getRawFileName (oracle/_ref, compiled from
sources/utils/radio_utils.cpp in place) and against the field convention of its reader scripts/converter.py."""
import os
import re
import time

import numpy as np
import pytest

import rtl_sdr_scanner_cpp_amd as pkg
from rtl_sdr_scanner_cpp_amd import replay

from oracle import oracle


def test_name_matches_reference_get_raw_file_name():
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built")
    import ctypes as C
    R = oracle.ref()
    for _ in range(3):  # the reference reads the wall clock itself: retry if a second boundary fell in between
        buf = C.create_string_buffer(1024)
        now = time.localtime()
        n = R.ref_get_raw_file_name(b"full", b"fc", 145_000_000, 2_048_000, buf, 1024)
        assert n > 0
        ours = replay.make_raw_file_name("full", "fc", 145_000_000, 2_048_000, now)
        if time.localtime().tm_sec == now.tm_sec:
            break
    assert buf.value.decode() == ours


def test_name_known_answer_and_converter_fields():
    t = time.struct_time((2025, 3, 7, 9, 5, 1, 0, 0, -1))
    name = replay.make_raw_file_name("full", "power", 144_500_000, 2_048_000, t)
    assert name == "./full_20250307_090501_144500000_2048000_power.raw"  # the snprintf format of radio_utils.cpp:82
    # the reference's own reader takes fields 3 and 4 of the bare name split at [._] (converter.py:58-59)
    f = re.split(r"[._]", os.path.basename(name))
    assert int(f[3]) == 144_500_000 and int(f[4]) == 2_048_000
    info = replay.parse_raw_file_name(name)
    assert (info.label, info.extension, info.frequency, info.sample_rate) == ("full", "power", 144_500_000, 2_048_000)
    assert info.kind == replay.KIND_F32 and info.timestamp == (2025, 3, 7, 9, 5, 1)


@pytest.mark.parametrize("name,kind", [
    ("./full_20250307_090501_144500000_2048000_fc.raw", replay.KIND_CF32),
    ("/data/recording_20240101_000000_-5000_32000_fc.raw", replay.KIND_CF32),  # a negative shift prints as %d
    ("full_20250307_090501_100000000_20000000.cs8", replay.KIND_CS8),          # converter.py:32: extension == "cs8"
    ("full_20250307_090501_100000000_20000000_cs8.raw", replay.KIND_CS8),      # ... or the name ends with cs8
    ("full_20250307_090501_100000000_2400000.cu8", replay.KIND_CU8),
])
def test_parse_kinds(name, kind):
    info = replay.parse_raw_file_name(name)
    assert info.kind == kind and info.sample_rate > 0
    if "-5000" in name:
        assert info.frequency == -5000


@pytest.mark.parametrize("bad", ["full_fc.raw", "full_2025_090501_1_2_fc.raw", "full_20250307_090501_abc_2048000_fc.raw",
                                 "full_20250307_090501_1000_0_fc.raw"])
def test_parse_rejects(bad):
    with pytest.raises(ValueError):
        replay.parse_raw_file_name(bad)


def test_sink_records_only_between_start_and_stop(tmp_path):
    """FileSink::work drops items unless recording, and opens the file at the first saved item (file_sink.h:19-29,61-68)."""
    n = 64
    sink = replay.RawFileSink(n * 4)  # FileSink<float>(fftSize): one item = one PSD row
    rows = np.arange(5 * n, dtype=np.float32).reshape(5, n)
    assert sink.work(rows[:1]) == 1  # not recording: consumed, nothing written
    path = tmp_path / "full_20250307_090501_144500000_2048000_power.raw"
    sink.start_recording(str(path))
    assert not path.exists()  # lazily opened
    sink.work(rows[1:3])
    sink.work(rows[3:4])
    sink.stop_recording()
    sink.work(rows[4:5])  # dropped again
    got = np.fromfile(path, np.float32).reshape(-1, n)
    np.testing.assert_array_equal(got, rows[1:4])
    # restart on a retune: a new file (sdr_device.cpp:64-65,75-76)
    path2 = tmp_path / "full_20250307_090502_145500000_2048000_power.raw"
    sink.start_recording(str(path2))
    sink.work(rows[4:5])
    sink.close()
    np.testing.assert_array_equal(np.fromfile(path2, np.float32), rows[4])


def test_sink_open_failure_raises(tmp_path):
    sink = replay.RawFileSink(8)
    sink.start_recording(str(tmp_path / "no_such_dir" / "x_20250307_090501_1_2_fc.raw"))
    with pytest.raises(OSError):  # FileSink::save throws std::runtime_error("open file failed")
        sink.work(np.zeros(4, np.complex64))


@pytest.mark.parametrize("decim", [1, 5])
def test_reader_reframes_like_stream_to_vector_and_decimator(tmp_path, decim):
    """Items of N*D samples, the first N kept (decimator.h:15-22); the trailing partial item is dropped."""
    n, items = 128, 7
    rng = np.random.default_rng(3)
    stream = (rng.standard_normal(items * n * decim + 37) + 1j * rng.standard_normal(items * n * decim + 37)).astype(np.complex64)
    path = tmp_path / "full_20250307_090501_144500000_2048000_fc.raw"
    sink = replay.RawFileSink(8)  # FileSink<gr_complex>(1): one item = one sample
    sink.start_recording(str(path))
    sink.work(stream[:1000])
    sink.work(stream[1000:])
    sink.close()
    rd = replay.RawIqReader(str(path), replay.KIND_CF32, n, decim)
    assert rd.items == items
    a = rd.read(3)
    b = rd.read(100)
    assert rd.read(4).shape[0] == 0
    want = stream[: items * n * decim].reshape(items, n * decim)[:, :n]
    np.testing.assert_array_equal(np.concatenate([a, b]), want)


def test_reader_cs8(tmp_path):
    n, items = 64, 5
    rng = np.random.default_rng(4)
    raw = rng.integers(-128, 128, size=(items * n + 3, 2), dtype=np.int8)
    path = tmp_path / "full_20250307_090501_100000000_20000000.cs8"
    raw.tofile(path)
    info = replay.parse_raw_file_name(str(path))
    assert replay.engine_overrides_for(info) == {"in_format": pkg.abi.SS_FMT_CS8, "int_scale": 1.0 / 127.5}
    rd = replay.RawIqReader(str(path), info.kind, n)
    got = rd.read(10)
    np.testing.assert_array_equal(got, raw[: items * n].reshape(items, n, 2))


def test_reader_rejects_power_files(tmp_path):
    p = tmp_path / "full_20250307_090501_144500000_2048000_power.raw"
    np.zeros(16, np.float32).tofile(p)
    with pytest.raises(OSError):
        replay.RawIqReader(str(p), replay.KIND_F32, 16)


def test_parallel_read_equals_sequential(tmp_path):
    """readFramesParallel (pread on disjoint ranges) returns the same frames as the single-stream read, including
    the short last batch."""
    n, items = 4096, 700  # 22 MiB of cf32: above the size where the parallel path engages
    rng = np.random.default_rng(5)
    data = rng.integers(0, 2**31, size=(items * n + 11, 2), dtype=np.int64).astype(np.float32).view(np.complex64).reshape(-1)
    path = tmp_path / "full_20250307_090501_144500000_2048000_fc.raw"
    data.tofile(path)
    a = replay.RawIqReader(str(path), replay.KIND_CF32, n)
    b = replay.RawIqReader(str(path), replay.KIND_CF32, n)
    buf_a, buf_b = np.empty((300, n), np.complex64), np.empty((300, n), np.complex64)
    for want in (300, 300, 100, 0):
        ga = a.read_into(buf_a, 300, threads=1)
        gb = b.read_into(buf_b, 300, threads=4)
        assert ga == gb == want
        np.testing.assert_array_equal(buf_a[:ga].view(np.uint64), buf_b[:gb].view(np.uint64))
