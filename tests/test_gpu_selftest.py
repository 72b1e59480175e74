"""Device-side arithmetic shortcuts, checked exhaustively on the GPU (run with -m gpu)."""
import pytest

import rtl_sdr_scanner_cpp_amd as pkg

pytestmark = pytest.mark.gpu


def test_division_by_21_is_the_ieee_division_for_every_float():
    """detect_fused.h div_const<21>: q0 = s*RN(1/21); q = fma(fma(-21, q0, s), RN(1/21), q0) must equal
    s / 21.0f (the reference divides: averager.cpp:56, utils.cpp:50) for all 2^32 bit patterns in range."""
    assert pkg.load_library().ss_selftest(0, 0) == 0
