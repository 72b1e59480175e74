#!/bin/bash
# round 6, session 16: 8192 points — the averaging tiles at a batch's start (rows in the halo frames' plane) on a straight-line path of their own
# (SS_STEADY_HALO=1) against the general path (scripts/ab/libspecscan_nosteadyhalo.so): the whole GPU suite, then alternating runs, 200 and 20
# steps, culled and with every tile evaluated
OUT=gpurun_out/r06_s16
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
timeout 1800 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
for i in 1 2 3; do
  for v in new old; do
    L="--diag-lib"; [ $v = old ] && L="--lib scripts/ab/libspecscan_nosteadyhalo.so"
    timeout 300 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-also --no-parity --no-live-pmc $L > $OUT/${v}_k200_$i.json 2>/dev/null
    timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-parity --no-live-pmc $L > $OUT/${v}_k20_$i.json 2>/dev/null
    [ $i = 1 ] && timeout 300 python bench.py --gpus 1 --steps 100 --warmup 20 --no-cull --no-cpu-baseline --no-also --no-parity --no-live-pmc $L > $OUT/${v}_nocull.json 2>/dev/null
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_s16/*.json')):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], j['ms_per_step'], j['value'], j['roofline']['frac'], j['roofline']['kernel_us'])
PY
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python /root/repo/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-parity --no-live-pmc --no-kernel-timing > /dev/null 2>&1
cd /root/repo
python scripts/timeline_tail.py $(find /tmp/tl -name "*kernel_trace.csv" | head -1) 8 > $OUT/timeline_k20_tail.txt; cat $OUT/timeline_k20_tail.txt
