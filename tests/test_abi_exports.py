"""CPU-side checks of the boundary: the HIP library builds for gfx950, loads, exports every symbol
include/specscan.h declares, and refuses to compute without a GPU (no CPU fallback)."""
import os
import re

import pytest

import rtl_sdr_scanner_cpp_amd as pkg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    pkg.build.build_lib()
    return pkg.load_library()


def _declared(header):
    text = open(os.path.join(ROOT, header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b((?:ss|orc)_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    names = _declared("include/specscan.h")
    assert set(names) == set(pkg.engine.EXPORTS)
    for name in names:
        assert hasattr(lib, name), name


def test_oracle_exports_the_same_set(oracle_mod):
    L = oracle_mod.lib()
    for name in pkg.engine.EXPORTS:
        if name in ("ss_device_count", "ss_process_device", "ss_flush", "ss_sync", "ss_stream", "ss_input_wait", "ss_get_stats", "ss_kernel_timing", "ss_kernel_timing_read", "ss_kernel_timing_read_slots", "ss_kernel_timing_read_frames", "ss_selftest", "ss_spectrogram_size", "ss_spectrogram_read") or name.startswith("ss_feed_"):  # (the oracle has its own orc_spectrogram_* object)
            continue  # device-only entry points (streams, pinned staging, PCIe pipelining)
        assert hasattr(L, "orc_" + name[3:]), name


def test_default_config_follows_reference_constants(lib, oracle_mod):
    import ctypes as C
    for L, p in ((lib, "ss_"), (oracle_mod.lib(), "orc_")):
        pkg.abi.bind(L, p)
        for fs, n, d in ((2_048_000, 8192, 5), (20_000_000, 131072, 3), (250_000, 1024, 4), (61_440_000, 262144, 4)):
            cfg = pkg.abi.SsConfig()
            getattr(L, p + "default_config")(C.byref(cfg), fs, 100_000_000)
            assert (cfg.fft_size, cfg.decim) == (n, d), (p, fs, cfg.fft_size, cfg.decim)
            assert (cfg.grouping_x, cfg.grouping_y, cfg.start_level, cfg.learn_ms) == (21, 21, 8.0, 2000)
            assert (cfg.range_lo, cfg.range_hi) == (100_000_000 - fs // 2, 100_000_000 + fs // 2)


def test_no_gpu_means_loud_failure(lib):
    if lib.ss_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError):
        pkg.SpectrumEngine(2_048_000, 145_000_000)
    import ctypes as C
    cfg = pkg.abi.SsConfig()
    lib.ss_default_config(C.byref(cfg), 2_048_000, 145_000_000)
    h = C.c_void_p()
    assert lib.ss_create(C.byref(cfg), C.byref(h)) == pkg.abi.SS_ERR_NO_DEVICE
    assert b"HIP device" in lib.ss_last_error(None)


def test_product_library_never_reads_the_environment():
    """Implementation choices can be overridden from the environment only in the -DSS_DIAG build (libspecscan_diag.so, A/B
    tests and measurement scripts): the shipped library has no getenv at all."""
    import subprocess
    from rtl_sdr_scanner_cpp_amd import build
    for lib_path, wanted in ((build.LIB, False), (build.LIB_DIAG, True)):
        if not os.path.exists(lib_path):
            pytest.skip(f"{lib_path} not built")
        syms = subprocess.run(["nm", "-D", "--undefined-only", lib_path], capture_output=True, text=True).stdout
        assert (" getenv" in syms) == wanted, lib_path


def test_product_never_imports_the_oracle():
    """The package must not reference oracle/ (voids parity claims otherwise)."""
    pkgdir = os.path.join(ROOT, "rtl-sdr-scanner-cpp_amd")
    for dirpath, _, files in os.walk(pkgdir):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, fn)).read()
                assert "liboracle" not in text and "from oracle" not in text and "import oracle" not in text, fn


def test_spectrogram_payload_layout(lib, oracle_mod):
    """uint64 ms | int32 start | int32 stop | int32 step | uint32 size | int8 row (data_controller.cpp:44-57)."""
    import ctypes as C
    import struct
    import numpy as np
    rng = np.random.default_rng(5)
    for fs, size, freq in ((2_048_000, 2048, 145_000_000), (20_000_000, 16384, 433_920_000), (250_000, 250, 27_000_000), (1_000_001, 3, 7)):
        row = rng.integers(-128, 128, size, dtype=np.int8)
        got = pkg.engine.spectrogram_payload(1_726_000_000_123, freq, fs, row)
        want = struct.pack("<Qiiii", 1_726_000_000_123, freq - fs // 2, freq + fs // 2, fs // size, size) + row.tobytes()
        assert got == want
        buf = np.zeros(len(got), np.uint8)
        L = oracle_mod.lib()
        L.orc_spectrogram_payload.argtypes = [C.c_uint64, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
        assert L.orc_spectrogram_payload(1_726_000_000_123, freq, fs, row.ctypes.data, size, buf.ctypes.data, len(buf)) == len(got)
        assert buf.tobytes() == got
    assert lib.ss_spectrogram_payload(0, 0, 1000, None, 0, None, 0) < 0  # the reference never frames an empty row (step = rate / size)
    short = np.zeros(8, np.uint8)
    assert lib.ss_spectrogram_payload(0, 0, 1000, np.zeros(4, np.int8).ctypes.data, 4, short.ctypes.data, 8) < 0
