#!/bin/bash
# round 3, GPU session 43: what the roles of the 8192-point step launch cost on the final tree (ablations, diagnostics build)
OUT=gpurun_out/r03_s43; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
B="timeout 200 python bench.py --no-cpu-baseline --no-also --diag-lib"
for rep in 1 2; do
for a in 0 1 2 3; do
  SS_ABLATE_ROLES=$a $B > $OUT/abl${a}_r$rep.json 2> $OUT/abl.err
done
SS_QUEUES=3 $B > $OUT/q3_r$rep.json 2> $OUT/abl.err
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r03_s43/*.json')):
    try:
        j = json.loads([l for l in open(f).read().splitlines() if l.startswith('{')][-1])
        print(os.path.basename(f), j['ms_per_step'], j['value'], j['roofline_chain']['frac'], j['roofline']['kernel_us'], j['roofline']['launches_in_flight'])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
