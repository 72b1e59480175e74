#!/usr/bin/env python
"""A/B builds of the diagnostics library for the cache-policy experiments (scripts/ab/, git-ignored)."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtl_sdr_scanner_cpp_amd as pkg

VARIANTS = {
    "base": [],
    "iqdef": ["SS_AUX_IQ=0"],
    "psdnt": ["SS_AUX_PSD=2"],
    "psdsc1": ["SS_AUX_PSD=16"],
    "psdsc0sc1": ["SS_AUX_PSD=17"],
    "psdntsc1": ["SS_AUX_PSD=18"],
    "detnt": ["SS_DET_NT=1"],
    "rowsnomax": ["SS_ROWS_ABL=1"],   # rows kernel of the long transforms without the run maxima (garbage culling: timing only)
    "rowsnoring": ["SS_ROWS_ABL=2"],  # ... without the ring rows
    "ringlegacy": ["SS_RING_AT_DRAIN=0"],
    "colstw15": ["SS_COLS_TW6=0"],   # column tiles: one step-A twiddle table entry per k (fifteen loads per thread) instead of six
    "colsnotw": ["SS_COLS_TW6=0", "SS_COLS_ABL=1"],   # column tiles of the long transforms without the step-A twiddle table loads (garbage results: timing only)
    "colsnowin": ["SS_COLS_ABL=2"],  # ... without the window loads
    "c1024nowin": ["SS_C1024_ABL=1"],    # 2^20 points in two passes: column tiles without the window loads (garbage results: timing only)
    "c1024nostore": ["SS_C1024_ABL=2"],  # ... without the work-buffer stores
    "c1024noload": ["SS_C1024_ABL=4"],   # ... without the frame loads
    "c1024none": ["SS_C1024_ABL=7"],     # ... arithmetic and LDS only
    "colsstag1": ["SS_COLS_STAGGER=1"],   # long transforms: every other column tile starts ~3.4 us late (do the phases of a one-round launch overlap better out of step?)
    "colsstag2": ["SS_COLS_STAGGER=2"],
    "colsstag3": ["SS_COLS_STAGGER=3"],
    "rowsstag1": ["SS_ROWS_STAGGER=1"],   # ... every other group of eight row tiles
    "rowsstag2": ["SS_ROWS_STAGGER=2"],
    "bothstag2": ["SS_COLS_STAGGER=2", "SS_ROWS_STAGGER=1"],
    "segdpp": ["SS_SEGMAX_LDS=0"],   # 8192 points: the per-column maxima for the tile culling in registers (v_max_f32_dpp), as until session 15 of round 4
    "ordfe": ["SS_ORDER_FFT_FIRST=1"],   # 8192 points, launches without an order table: the frames dispatched ahead of the candidate lists
    "dif8w8": ["SS_DIF8_W=8"],   # 65536 points, the radix-8 fold: one residue per workgroup (64 registers, eight waves per SIMD)
    "dif8w4": ["SS_DIF8_W=4"],   # ... two residues per workgroup (128 registers, four waves per SIMD)
    "genrows": ["SS_GENERAL_ROW_LANES=0"],   # ... row by row on the scalar unit with session 33's expressions (the shipped form works the 36 rows out on 36 lanes)
    "genold": ["SS_GENERAL_ROW_OLD=1", "SS_GENERAL_ROW_LANES=0"],   # the averaging tiles' general path with its per-row address and value expressions as they were until session 33 of round 5
    "longwt": ["SS_AUX_WORK=16", "SS_AUX_ROWS=16"],   # long transforms: work buffer and dB / ring rows stored write-through (sc1) — measured in round 6: the row launches lose a third
    "steadyhalo": ["SS_STEADY_HALO=1"],   # 8192 points: the tiles at a batch's start (rows in the halo frames' plane) on a straight-line path of their own — measured in round 6: no gain
    "plancopy16": ["SS_PLAN_COPY_LOADS=16"],   # 8192 points: the plan copies its columns' maxima sixteen loads at a time — one round trip per 1024-frame batch — instead of eight (measured in round 6: no gain)
    "dif8acc": ["SS_DIF8_BFLY=0"],   # 65536 points: the fold as round 5's accumulating loop over q (round 6 ships a radix-8 butterfly per point)
    "colsnone": ["SS_COLS_TW6=0", "SS_COLS_ABL=3"],   # ... without either  # 8192 points, deep pipelining: ring rows written by three frame tiles of every call
}

if __name__ == "__main__":
    want = sys.argv[1:] or list(VARIANTS)
    with ThreadPoolExecutor(4) as ex:
        for path in ex.map(lambda t: pkg.build.build_variant(t, VARIANTS[t], force=True), want):
            print(path)
