#!/bin/bash
# round 6, session 19: the drain's join behind a waiter on the public stream + a tail event (drain_deep) against the form until now
# (diagnostics build, SS_DRAIN_WAITER_US=0 SS_DRAIN_TAIL_EVENT=0), alternating, 20 and 200 steps; each lever alone; then the timeline
OUT=gpurun_out/r06_s19
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
B="--gpus 1 --no-cpu-baseline --no-also --no-parity --no-live-pmc --diag-lib"
for i in 1 2 3 4; do
  for v in new old waiter event; do
    E=""; [ $v = old ] && E="SS_DRAIN_WAITER_US=0 SS_DRAIN_TAIL_EVENT=0"; [ $v = waiter ] && E="SS_DRAIN_TAIL_EVENT=0"; [ $v = event ] && E="SS_DRAIN_WAITER_US=0"
    env $E timeout 300 python bench.py $B --steps 20 --warmup 5 > $OUT/k20_${v}_$i.json 2>/dev/null
    [ $i -le 2 ] && [ $v != waiter ] && [ $v != event ] && env $E timeout 300 python bench.py $B --steps 200 --warmup 20 > $OUT/k200_${v}_$i.json 2>/dev/null
  done
done
for i in 1 2; do timeout 300 python bench.py --gpus 1 --no-cpu-baseline --no-also --no-parity --no-live-pmc --steps 20 --warmup 5 > $OUT/k20_prod_$i.json 2>/dev/null; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_s19/*.json')):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], j['ms_per_step'], j['value'], j['roofline']['frac'], j['roofline']['kernel_us'])
PY
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python /root/repo/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-parity --no-live-pmc --no-kernel-timing > /dev/null 2>&1
cd /root/repo
python scripts/timeline_tail.py $(find /tmp/tl -name "*kernel_trace.csv" | head -1) 14 all > $OUT/timeline_k20_tail_all.txt; cat $OUT/timeline_k20_tail_all.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stream_ordered.py tests/test_gpu_cull.py -x -q -m gpu 2>&1 | tail -3
