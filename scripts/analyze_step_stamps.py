#!/usr/bin/env python
"""Summarise the per-workgroup stamps of one k_scan_step launch (libspecscan_diag.so, SS_STEP_STAMPS=<file>):
lifetimes per role, how many workgroups of each role are resident over time, when each CU slot chain ends."""
import sys

import numpy as np

d = np.loadtxt(sys.argv[1], dtype=np.int64)
blk, t0, t1, role, item, xcc, hw = d.T
T0 = t0.min()
st, en = (t0 - T0) / 100.0, (t1 - T0) / 100.0  # 100 MHz -> us
names = {1: "fft", 2: "det", 3: "emit"}
print(f"launch span {en.max():.1f} us, {len(d)} workgroups")
for r in (3, 2, 1):
    m = role == r
    if m.any():
        life = (en - st)[m]
        print(f"{names[r]:5s} n {m.sum():5d}  life us p10 {np.percentile(life, 10):6.2f} p50 {np.percentile(life, 50):6.2f} p90 {np.percentile(life, 90):6.2f} max {life.max():6.2f}"
              f" | start p0 {st[m].min():6.2f} p50 {np.percentile(st[m], 50):6.2f} p100 {st[m].max():6.2f} | end p50 {np.percentile(en[m], 50):6.2f} p100 {en[m].max():6.2f}")
ts = np.linspace(0, en.max(), 25)
print("t us     ", " ".join(f"{t:5.1f}" for t in ts))
for r in (1, 2, 3):
    m = role == r
    print(f"{names[r]:5s} res ", " ".join(f"{int(((st[m] <= t) & (en[m] > t)).sum()):5d}" for t in ts))
print("all   res ", " ".join(f"{int(((st <= t) & (en > t)).sum()):5d}" for t in ts))
cu = xcc * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 12) & 1) * 20 + ((hw >> 8) & 15)
ncu = len(np.unique(cu))
last = np.array([en[cu == c].max() for c in np.unique(cu)])
print(f"{ncu} CUs; last workgroup end per CU: p10 {np.percentile(last, 10):.1f} p50 {np.percentile(last, 50):.1f} p90 {np.percentile(last, 90):.1f} max {last.max():.1f}")
per = np.array([[(role[cu == c] == r).sum() for r in (1, 2, 3)] for c in np.unique(cu)])
print("items per CU (fft, det, emit): mean", per.mean(0).round(2), "min", per.min(0), "max", per.max(0))

import os
if os.path.exists(sys.argv[1] + ".det"):
    m = np.loadtxt(sys.argv[1] + ".det", dtype=np.int64)
    # tile t belongs to detect item t // 2: find its workgroup
    wg_of_item = {int(i): k for k, (r, i) in enumerate(zip(role, item)) if r == 2}
    p1, p2, tail = [], [], []
    for t, a_, b_ in m:
        k = wg_of_item.get(int(t) // 2)
        if k is None or a_ == 0:
            continue
        p1.append((a_ - t0[k]) / 100.0)
        p2.append((b_ - a_) / 100.0)
        tail.append((t1[k] - b_) / 100.0)
    for name, v in (("phase 1 (loads + time means)", p1), ("phase 2 (bin means + threshold)", p2), ("tail (atomics, exit)", tail)):
        v = np.array(v)
        print(f"detect {name:32s} us p10 {np.percentile(v, 10):5.2f} p50 {np.percentile(v, 50):5.2f} p90 {np.percentile(v, 90):5.2f}")
