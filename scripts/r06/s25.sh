#!/bin/bash
# round 6, session 25: ss_sync waits for the two queues' tail events on the host and leaves the public stream's join out (drain_deep, join_pub =
# false). The 20-step form in four shapes, alternating (diagnostics build): the end of the timed region as eng.sync() + device-wide
# synchronisation ("eng") or the device-wide one alone ("dev"), with SS_LAZY_JOIN=1 (new) or 0 (old); then the pipeline / culling suites
# (the SS_LAZY_JOIN switch this script sets existed in the tree of this session only: drain_deep(join_pub = false), taken out again — DESIGN.md 4.1)
OUT=gpurun_out/r06_s25
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
B="--gpus 1 --no-cpu-baseline --no-also --no-parity --no-live-pmc --diag-lib --steps 20 --warmup 5"
for i in 1 2 3 4 5 6 7 8; do
  for v in neweng newdev oldeng olddev; do
    E="SS_LAZY_JOIN=1"; [ ${v:0:3} = old ] && E="SS_LAZY_JOIN=0"
    S=""; [ ${v:3} = eng ] && S="--sync-engine-first"
    env $E timeout 300 python bench.py $B $S > $OUT/k20_${v}_$i.json 2>/dev/null
  done
done
for i in 1 2; do for v in neweng olddev; do
  E="SS_LAZY_JOIN=1"; S="--sync-engine-first"; [ $v = olddev ] && { E="SS_LAZY_JOIN=0"; S=""; }
  env $E timeout 300 python bench.py --gpus 1 --no-cpu-baseline --no-also --no-parity --no-live-pmc --diag-lib --steps 200 --warmup 20 $S > $OUT/k200_${v}_$i.json 2>/dev/null
done; done
python - <<'PY'
import json, glob, statistics
acc = {}
for f in sorted(glob.glob('gpurun_out/r06_s25/*.json')):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    acc.setdefault('_'.join(f.split('/')[-1].split('_')[:2]), []).append(round(j['ms_per_step'] * 1e3, 2))
for k, v in acc.items():
    print(k, sorted(v), 'median', statistics.median(v))
PY
timeout 1200 python -m pytest tests/test_gpu_step_pipeline.py tests/test_gpu_stream_ordered.py tests/test_gpu_cull.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
