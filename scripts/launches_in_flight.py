#!/usr/bin/env python
"""How many launches of a kernel are in flight at a time, from a rocprofv3 --kernel-trace CSV: the sum of the launches'
durations over the wall time they span, for the longest run of back-to-back launches (gaps below `--gap-us`), plus the
share of that time with 0 / 1 / 2 / 3+ launches running and the queues they ran on.
Usage: python scripts/launches_in_flight.py <kernel_trace.csv> [--kernel k_scan_step] [--gap-us 200]"""
import argparse
import csv
import collections


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--kernel", default="k_scan_step")
    ap.add_argument("--gap-us", type=float, default=200.0)
    a = ap.parse_args()
    rows = [r for r in csv.DictReader(open(a.trace)) if a.kernel in r["Kernel_Name"]]
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")) for r in rows)
    runs, cur = [], [ev[0]]
    for e in ev[1:]:
        if e[0] - max(x[1] for x in cur[-4:]) > a.gap_us * 1e3:
            runs.append(cur)
            cur = []
        cur.append(e)
    runs.append(cur)
    run = max(runs, key=len)
    t0, t1 = run[0][0], max(e[1] for e in run)
    busy = sum(e[1] - e[0] for e in run)
    points = sorted([(e[0], 1) for e in run] + [(e[1], -1) for e in run])
    depth, last, hist = 0, t0, collections.Counter()
    for t, d in points:
        hist[min(depth, 3)] += t - last
        depth += d
        last = t
    queues = collections.Counter(e[2] for e in run)
    print(f"{len(ev)} launches of *{a.kernel}* in the trace; longest back-to-back run: {len(run)} launches over {(t1 - t0) / 1e3:.1f} us")
    print(f"  mean launch duration {busy / len(run) / 1e3:.2f} us, wall time per launch {(t1 - t0) / len(run) / 1e3:.2f} us, launches in flight {busy / (t1 - t0):.2f}")
    print("  share of the time with 0 / 1 / 2 / 3+ launches running: " + " / ".join(f"{100.0 * hist[k] / (t1 - t0):.1f} %" for k in range(4)))
    print("  launches per hardware queue: " + ", ".join(f"queue {q}: {n}" for q, n in sorted(queues.items())))
    shapes = collections.defaultdict(list)
    for r in rows:
        shapes[int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("  mean duration by launch shape (workgroups: launches, us): " + "; ".join(f"{k}: {len(v)}, {sum(v) / len(v):.1f}" for k, v in sorted(shapes.items())))


if __name__ == "__main__":
    main()
