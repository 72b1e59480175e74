#!/usr/bin/env python
"""Summarise the per-workgroup stamps of one k_scan_step launch (libspecscan_diag.so, SS_STEP_STAMPS=<file>):
lifetimes per role, how many workgroups of each role are resident over time, when each CU slot chain ends."""
import sys

import numpy as np

d = np.loadtxt(sys.argv[1], dtype=np.int64)
blk, t0, t1, role, item, xcc, hw = d.T
T0 = t0.min()
st, en = (t0 - T0) / 100.0, (t1 - T0) / 100.0  # 100 MHz -> us
names = {1: "fft", 2: "det", 3: "emit", 4: "plan", 5: "rows"}
print(f"launch span {en.max():.1f} us, {len(d)} workgroups")
for r in (4, 3, 2, 5, 1):
    m = role == r
    if m.any():
        life = (en - st)[m]
        print(f"{names[r]:5s} n {m.sum():5d}  life us p10 {np.percentile(life, 10):6.2f} p50 {np.percentile(life, 50):6.2f} p90 {np.percentile(life, 90):6.2f} max {life.max():6.2f}"
              f" | start p0 {st[m].min():6.2f} p50 {np.percentile(st[m], 50):6.2f} p100 {st[m].max():6.2f} | end p50 {np.percentile(en[m], 50):6.2f} p100 {en[m].max():6.2f}")
ts = np.linspace(0, en.max(), 25)
print("t us     ", " ".join(f"{t:5.1f}" for t in ts))
for r in (1, 2, 3, 4, 5):
    m = role == r
    if not m.any():
        continue
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
    # four stamps per evaluated tile: start, after the first pass of phase 1, after phase 1, after phase 2
    m = np.loadtxt(sys.argv[1] + ".det", dtype=np.int64)
    m = m[(m[:, 1] != 0) & (m[:, 3] != 0)]
    if len(m):
        tcols = int(sys.argv[2]) if len(sys.argv) > 2 else 256
        a0, a1, a2, a3 = [(m[:, k] - T0) / 100.0 for k in (1, 2, 3, 4)]
        print(f"{len(m)} tiles evaluated; per tile, us:")
        for name, v in (("start (launch clock)", a0), ("start -> first pass of phase 1 done", a1 - a0), ("-> phase 1 done (second pass, barrier)", a2 - a1),
                        ("-> phase 2 done", a3 - a2), ("whole tile", a3 - a0)):
            print(f"  {name:40s} p10 {np.percentile(v, 10):6.2f} p50 {np.percentile(v, 50):6.2f} p90 {np.percentile(v, 90):6.2f} max {v.max():6.2f}")
        if len(sys.argv) > 3:
            for t, s0, s1, s2, s3 in sorted(zip(m[:, 0], a0, a1, a2, a3), key=lambda r: (r[0] % tcols, r[0] // tcols)):
                print(f"  col {t % tcols:4d} ft_seq {t // tcols:3d}  start {s0:6.2f}  +{s1 - s0:5.2f} +{s2 - s1:5.2f} +{s3 - s2:5.2f}")
