"""The Spectrogram side branch and the MQTT framing, pinned on the reference's OWN code: oracle/_ref now holds
radio/blocks/spectrogram.cpp and network/data_controller.cpp compiled in place (oracle/Makefile; Mqtt is a stub that keeps
what is published, oracle/stubs/network/mqtt.h). The C restatement (orc_spectrogram_*) and the product's host-side framing
helpers (ss_spectrogram_payload, sc_transmission_payload: no GPU involved) must reproduce it byte for byte."""
import ctypes as C
import struct

import numpy as np
import pytest

import rtl_sdr_scanner_cpp_amd as pkg


def _psd_rows(rng, nframes, n):
    """dB rows like PSD::work's: a noise floor around -60 dB with a few stronger stretches, some exact integers in the mix."""
    x = (-60.0 + 6.0 * rng.standard_normal((nframes, n))).astype(np.float32)
    x[:, n // 5:n // 5 + n // 40] += 30.0
    x[:, n // 2] = -51.0
    return x


@pytest.mark.parametrize("n,fs", [(8192, 2_048_000), (2048, 512_000), (65536, 20_000_000), (1024, 250_000), (32768, 8_000_000)])
def test_restatement_equals_the_reference_spectrogram(ref_mod, n, fs):
    O = ref_mod
    rng = np.random.default_rng(n)
    L = O.lib()
    ref = O.RefSpectrogram(n, fs, 100_000_000)
    orc = L.orc_spectrogram_create(n, fs)
    assert L.orc_spectrogram_size(orc) == ref.size == min(16384, O.ref().ref_get_fft(fs, 1000), n)
    now = 5_000
    sent = 0
    for burst in range(6):
        rows = _psd_rows(rng, int(rng.integers(1, 40)), n)
        # inside a burst the clock stands still: Spectrogram::send's gate (now > last + 1000 ms) stays shut
        assert ref.work(rows, now) == 0
        for r in rows:
            L.orc_spectrogram_process(orc, r.ctypes.data_as(C.POINTER(C.c_float)))
        want_sum, want_cnt = ref.container()
        # the restatement's container, read through a send on a copy: sums and count must already agree bit for bit
        row8 = np.empty(ref.size, np.int8)
        mean = np.empty(ref.size, np.float32)
        # one more frame after the interval: the reference accumulates it, then publishes and clears
        now += 1_001
        last = _psd_rows(rng, 1, n)
        assert ref.work(last, now) == 1
        L.orc_spectrogram_process(orc, last[0].ctypes.data_as(C.POINTER(C.c_float)))
        cnt = L.orc_spectrogram_send(orc, row8.ctypes.data_as(C.POINTER(C.c_int8)), mean.ctypes.data_as(C.POINTER(C.c_float)))
        assert cnt == want_cnt + 1
        payload = ref.pop()
        sent += 1
        t_ms, start, stop, step, size = struct.unpack_from("<QiiiI", payload)
        assert (t_ms, start, stop, step, size) == (now, 100_000_000 - fs // 2, 100_000_000 + fs // 2, fs // ref.size, ref.size)
        got8 = np.frombuffer(payload, np.int8, offset=24)
        np.testing.assert_array_equal(got8, row8)  # int8(sum / count): the reference's own truncation
        # framing: the restatement's and the product's helper give the reference's bytes
        buf = np.zeros(len(payload), np.uint8)
        assert L.orc_spectrogram_payload(now, 100_000_000, fs, row8.ctypes.data, ref.size, buf.ctypes.data, buf.size) == len(payload)
        assert buf.tobytes() == payload
        assert pkg.engine.spectrogram_payload(now, 100_000_000, fs, row8) == payload
        assert ref.container()[1] == 0 and not ref.container()[0].any()  # container cleared (spectrogram.cpp:71-73)
    assert sent == 6
    L.orc_spectrogram_destroy(orc)


def test_reference_keeps_one_container_per_centre_frequency(ref_mod):
    O = ref_mod
    rng = np.random.default_rng(3)
    n, fs = 8192, 2_048_000
    ref = O.RefSpectrogram(n, fs, 100_000_000)
    a, b = _psd_rows(rng, 5, n), _psd_rows(rng, 7, n)
    ref.work(a, 1000)
    ref.set_frequency(102_000_000)
    ref.work(b, 1000)
    assert ref.container()[1] == 7
    ref.set_frequency(100_000_000)
    sum_a, cnt_a = ref.container()
    assert cnt_a == 5
    m = n // ref.size
    want = np.zeros(ref.size, np.float32)
    for r in a:  # spectrogram.cpp:51-58 in numpy: ascending fp32 sum of m bins, / m, accumulated frame by frame
        s = np.zeros(ref.size, np.float32)
        for j in range(m):
            s = s + r[j::m]
        want = want + s / np.float32(m)
    np.testing.assert_array_equal(sum_a, want)


def test_transmission_framing_equals_the_reference(ref_mod):
    rng = np.random.default_rng(8)
    iq = rng.integers(-128, 128, size=(1000, 2)).astype(np.int8)
    want = ref_mod.ref_transmission_payload(1_726_000_000_123, 145_000_000, 32_000, iq)
    assert pkg.channelizer.transmission_payload(1_726_000_000_123, 145_000_000, 32_000, iq) == want
    t_ms, start, stop, rate = struct.unpack_from("<QiiI", want)
    assert (t_ms, start, stop, rate) == (1_726_000_000_123, 145_000_000 - 16_000, 145_000_000 + 16_000, 32_000)
    np.testing.assert_array_equal(np.frombuffer(want, np.uint8, offset=20), (iq.reshape(-1).view(np.uint8) ^ 0x80))
