"""The reference's own gtest known-answer vectors (reference tests/*.cpp), run against
  (a) the C restatement oracle/liboracle.so and
  (b) oracle/_ref — the reference's own source files compiled in place — when it is built.
This is what pins the back end of the oracle."""
import ctypes as C

import numpy as np
import pytest


def _impls(oracle_mod):
    out = [("oracle", oracle_mod.lib(), "orc_")]
    if oracle_mod.have_ref():
        out.append(("ref", oracle_mod.ref(), "ref_"))
    return out


def fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


# ---- tests/test_utils.cpp:4-13 (Utils.AverageX) ----
def test_average_x(oracle_mod):
    inp = np.arange(1, 10, dtype=np.float32)
    want = np.array([2, 2.5, 3, 4, 5, 6, 7, 7.5, 8], dtype=np.float32)
    for name, lib, p in _impls(oracle_mod):
        out = np.zeros(9, np.float32)
        getattr(lib, p + "average")(fp(inp), fp(out), 9, 5)
        np.testing.assert_allclose(out, want, rtol=4 * np.finfo(np.float32).eps, err_msg=name)  # EXPECT_FLOAT_EQ = 4 ulp


# ---- tests/test_radio_utils.cpp:4-16 (RadioUtils.Fft) ----
@pytest.mark.parametrize("fs,step,n", [
    (2048000 - 1, 1000, 2048), (2048000, 1000, 2048), (2048000 + 1, 1000, 4096),
    (20480000 - 1, 625, 32768), (20480000, 625, 32768), (20480000 + 1, 625, 65536),
    (104857600 - 1, 100, 1048576), (104857600, 100, 1048576), (104857600 + 1, 100, 2097152)])
def test_get_fft(oracle_mod, fs, step, n):
    for name, lib, p in _impls(oracle_mod):
        assert getattr(lib, p + "get_fft")(fs, step) == n, name


# ---- tests/test_radio_utils.cpp:71-103 (RadioUtils.TunedFrequency) ----
@pytest.mark.parametrize("f,step,want", [
    (-999, 1000, -1000), (-1001, 1000, -1000), (-1499, 1000, -1000), (-1500, 1000, -1000), (-1501, 1000, -2000),
    (999, 1000, 1000), (1001, 1000, 1000), (1499, 1000, 1000), (1500, 1000, 2000), (1501, 1000, 2000),
    (499, 500, 500), (500, 500, 500), (501, 500, 500), (749, 500, 500), (750, 500, 1000), (751, 500, 1000),
    (999, 500, 1000), (1000, 500, 1000), (1001, 500, 1000), (1249, 500, 1000), (1250, 500, 1500), (1251, 500, 1500)])
def test_tuned_frequency(oracle_mod, f, step, want):
    for name, lib, p in _impls(oracle_mod):
        assert getattr(lib, p + "get_tuned_frequency")(f, step) == want, name


# ---- tests/test_collection_utils.cpp:4-46 (ContaisWithMargin0/1/2) ----
@pytest.mark.parametrize("margin,hits", [
    (0, {9: 0, 10: 1, 11: 0, 13: 0, 14: 1, 15: 0}),
    (1, {8: 0, 9: 1, 10: 1, 11: 1, 12: 0, 13: 1, 14: 1, 15: 1, 16: 0}),
    (2, {8: 0, 9: 1, 10: 1, 11: 1, 12: 0, 13: 1, 14: 1, 15: 1, 16: 0})])
def test_contains_with_margin(oracle_mod, margin, hits):
    keys = np.array([10, 14], np.int32)
    for name, lib, p in _impls(oracle_mod):
        for index, want in hits.items():
            found = C.c_int32(-1)
            got = getattr(lib, p + "contains_with_margin")(ip(keys), 2, index, margin, C.byref(found))
            assert got == want, (name, margin, index)


# ---- tests/test_collection_utils.cpp:48-63 (mostFrequentValue) ----
@pytest.mark.parametrize("data,want", [
    ([1, 2, 3, 4, 5, 5], 5), ([3, 3, 1, 1, 5, 5], 3), ([3, 3, 1, 1, 5, 5, 2, 2], 3), ([1, 1, 1, 1, 2, 5, 5, 5], 1),
    ([1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10], 1)])
def test_most_frequent_value(oracle_mod, data, want):
    d = np.array(data, np.int32)
    for name, lib, p in _impls(oracle_mod):
        assert getattr(lib, p + "most_frequent_value")(ip(d), d.size) == want, name


# ---- tests/test_collection_utils.cpp:81-106 (getMaxIndex) ----
@pytest.mark.parametrize("index,group,want", [
    (0, 0, 0), (0, 1, 0), (0, 2, 1), (0, 3, 1),
    (8, 0, 8), (8, 1, 8), (8, 2, 7), (8, 3, 7), (8, 4, 6), (8, 5, 6), (8, 6, 5),
    (2, 0, 2), (2, 1, 2), (2, 2, 3), (2, 3, 3), (2, 4, 4), (2, 5, 4), (2, 6, 4), (2, 7, 4), (2, 8, 4), (2, 9, 4)])
def test_get_max_index(oracle_mod, index, group, want):
    data = np.array([1, 2, 3, 4, 5, 4, 3, 2, 1], np.float32)
    for name, lib, p in _impls(oracle_mod):
        assert getattr(lib, p + "get_max_index")(fp(data), 9, index, group) == want, name


# ---- tests/test_averager.cpp ----
class _Avg:
    def __init__(self, name, lib, p, size, group):
        self.lib, self.p, self.size, self.group, self.is_ref = lib, p, size, group, name == "ref"
        self.h = getattr(lib, p + "averager_create")(size, group)

    def push(self, row):
        r = np.ascontiguousarray(row, np.float32)
        getattr(self.lib, self.p + "averager_push")(self.h, fp(r))

    def reset(self):
        getattr(self.lib, self.p + "averager_reset")(self.h)

    def average(self):
        if self.is_ref:
            out = np.empty(self.size, np.float32)
            self.lib.ref_averager_average(self.h, fp(out))
            return out
        return np.ctypeslib.as_array(self.lib.orc_averager_average(self.h), (self.size,)).copy()

    def data(self):
        rows = []
        for r in range(self.group):
            if self.is_ref:
                out = np.empty(self.size, np.float32)
                self.lib.ref_averager_row(self.h, r, fp(out))
                rows.append(out)
            else:
                rows.append(np.ctypeslib.as_array(self.lib.orc_averager_row(self.h, r), (self.size,)).copy())
        return np.stack(rows)


def test_averager_fixture_simple_and_big(oracle_mod):
    """AveragerTest.SimpleTest + SimpleBigTest (tests/test_averager.cpp:46-86): the averager equals the
    brute-force fp32 mean of the last 3 rows, -100 until 3 rows were pushed, ring contents match."""
    size, group = 5, 3
    for name, lib, p in _impls(oracle_mod):
        for rows in ([[1, 2, 3, 4, 5], [2, 3, 4, 5, 6], [3, 4, 5, 6, 7], [6, 7, 8, 9, 10], [7, 8, 9, 10, 11]],
                     [[1, 2, 3, 4, 5], [2, 3, 4, 5, 6]] + [[i * 11 + j * 7 for j in range(size)] for i in range(1, 123)]):
            a = _Avg(name, lib, p, size, group)
            raw = [np.zeros(size, np.float32) for _ in range(group)]
            for k, row in enumerate(rows):
                a.push(row)
                raw.append(np.array(row, np.float32))
                raw = raw[-group:]
                np.testing.assert_array_equal(a.data(), np.stack(raw), err_msg=name)
                if k + 1 < group:
                    np.testing.assert_array_equal(a.average(), np.full(size, -100, np.float32), err_msg=name)
                else:
                    s = np.zeros(size, np.float32)
                    for r in raw:
                        s = s + r  # fp32, oldest first, as the fixture's average() does
                    np.testing.assert_array_equal(a.average(), s / np.float32(group), err_msg=name)


def test_averager_exact_values_and_reset(oracle_mod):
    """Averager.SimpleTest (tests/test_averager.cpp:88-140)."""
    size = 5
    gen = lambda v: np.full(size, v, np.float32)  # noqa: E731
    for name, lib, p in _impls(oracle_mod):
        a = _Avg(name, lib, p, size, 3)
        for _ in range(2):
            np.testing.assert_array_equal(a.average(), gen(-100))
            np.testing.assert_array_equal(a.data(), np.stack([gen(0), gen(0), gen(0)]))
            for v, want_avg, want_rows in [(1, -100, (0, 0, 1)), (2, -100, (0, 1, 2)), (3, 2, (1, 2, 3)), (10, 5, (2, 3, 10)),
                                           (11, 8, (3, 10, 11))]:
                a.push(gen(v))
                np.testing.assert_array_equal(a.average(), gen(want_avg), err_msg=name)
                np.testing.assert_array_equal(a.data(), np.stack([gen(r) for r in want_rows]), err_msg=name)
            a.reset()
