#!/usr/bin/env python
"""A/B builds of the diagnostics library for the cache-policy experiments (scripts/ab/, git-ignored)."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtl_sdr_scanner_cpp_amd as pkg

VARIANTS = {
    "base": [],
    "iqnt": ["SS_AUX_IQ=2"],
    "iqnt_sc0": ["SS_AUX_IQ=3"],
    "iqnt_sc1": ["SS_AUX_IQ=18"],
    "iqnt_detnt": ["SS_AUX_IQ=2", "SS_DET_NT=1"],
    "iqnt_psdnt": ["SS_AUX_IQ=2", "SS_AUX_PSD=2"],
    "psdnt": ["SS_AUX_PSD=2"],
}

if __name__ == "__main__":
    want = sys.argv[1:] or list(VARIANTS)
    with ThreadPoolExecutor(4) as ex:
        for path in ex.map(lambda t: pkg.build.build_variant(t, VARIANTS[t], force=True), want):
            print(path)
