#!/usr/bin/env python
"""One line per bench.py result in a gpurun_out session directory (tag, us per call, MS/s, kernel slots) — the summaries under profiles/r05/
of the sessions whose scripts print their table only to the terminal. Usage: python scripts/r05/summarize.py gpurun_out/r05_sNN"""
import glob
import json
import os
import sys

d = sys.argv[1]
for p in sorted(glob.glob(os.path.join(d, "*.json"))):
    if os.path.basename(p).startswith("st_"):
        continue
    try:
        j = json.loads(open(p).read().strip().splitlines()[-1])
        par = j.get("parity") or {}
        print(f"{os.path.basename(p)[:-5]:28s} {1000 * j['ms_per_step']:8.1f} us  {j['value']:10.1f} {j['unit']}  kernels {[(k['slot'], k['us']) for k in j['roofline'].get('kernels', [])]}"
              + (f"  parity failed={par.get('failed')}" if par else ""))
    except Exception as e:  # noqa: BLE001
        print(f"{os.path.basename(p)[:-5]:28s} ERR {e}")
