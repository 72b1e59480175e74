"""Summarise the per-workgroup time stamps written by SS_DEBUG_TIMING=<file> (k_detect_fused)."""
import sys
import numpy as np
d = np.loadtxt(sys.argv[1])
t0 = d[:, 1].min()
st, mid, en, cls = (d[:, 1] - t0) / 100.0, (d[:, 2] - t0) / 100.0, (d[:, 3] - t0) / 100.0, d[:, 4]  # 100 MHz -> us
print("kernel span us %.1f, blocks %d" % (en.max(), len(d)))
for c in (1, 2, 3):
    m = cls == c
    if m.any():
        print("class %d n %d: dur mean %.2f max %.2f | phase1 mean %.2f | start min %.1f max %.1f" % (
            c, m.sum(), (en - st)[m].mean(), (en - st)[m].max(), (mid - st)[m].mean(), st[m].min(), st[m].max()))
print("start pct", np.round(np.percentile(st, [0, 10, 25, 50, 75, 90, 99, 100]), 1))
print("end   pct", np.round(np.percentile(en, [0, 10, 25, 50, 75, 90, 99, 100]), 1))
# concurrency over time
ts = np.linspace(0, en.max(), 30)
print("resident blocks:", [int(((st <= t) & (en > t)).sum()) for t in ts])
