#!/usr/bin/env python
"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of one command, per kernel and grid size:
bytes = (2 FETCH_SIZE + WRITE_SIZE) KiB (the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md, re-checked by every session's
calibration kernels: 64 MiB read -> FETCH 32 779 KiB, 32 MiB written -> WRITE 32 768 KiB). The first and last launches of a shape are
learning / drain calls and are averaged in; a shape needs five launches to be listed.
usage: pmc_traffic_summary.py DIR PREFIX name:samples_per_call [name:samples ...]
   reads DIR/PREFIX_<name>_pmc_fetch.csv and DIR/PREFIX_<name>_pmc_write.csv"""
import collections
import csv
import sys


def per_shape(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and r["Kernel_Name"].startswith(("void ss::", "ss::")):
            acc[(r["Kernel_Name"].split("(")[0], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return acc


def main():
    d, prefix = sys.argv[1], sys.argv[2]
    for spec in sys.argv[3:]:
        name, samples = spec.split(":")
        samples = int(samples)
        f = per_shape(f"{d}/{prefix}_{name}_pmc_fetch.csv", "FETCH_SIZE")
        w = per_shape(f"{d}/{prefix}_{name}_pmc_write.csv", "WRITE_SIZE")
        print(f"== {name} ({samples} samples per call)")
        for key in sorted(f, key=lambda k: -len(f[k])):
            if key not in w or len(f[key]) < 5 or "k_fill" in key[0]:
                continue
            fk, wk = sum(f[key]) / len(f[key]), sum(w[key]) / len(w[key])
            b = (2 * fk + wk) * 1024
            print(f"   {key[0]} grid {key[1]}: {len(f[key])} launches, FETCH {fk:.0f} KiB, WRITE {wk:.0f} KiB -> (2 F + W) = {b / 1e6:.1f} MB per launch = {b / samples:.2f} B/sample")


if __name__ == "__main__":
    main()
