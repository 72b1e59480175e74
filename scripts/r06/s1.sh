#!/bin/bash
# round 6, session 1: first contact of the round's tree — the new tests (retune + ss_read_window, two rank processes on one device),
# the whole GPU suite, smoke, the driver's form of the default line with the COMPACT last line (bench_full.json beside it), the same
# without side legs three times (no event-timed launch inside a short timed region any more), and the 20-step run as a timeline
OUT=gpurun_out/r06_s1
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_multi_rank.py tests/test_gpu_stated_configs.py -x -q -m gpu -k "multi or ranks or retune" > $OUT/pytest_new.txt 2>&1; tail -3 $OUT/pytest_new.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default_k20.json 2> $OUT/bench_default_k20.err ) 2> $OUT/bench_default_k20.time
cp bench_full.json $OUT/bench_full_k20.json 2>/dev/null
tail -c 300 $OUT/bench_default_k20.time
wc -c $OUT/bench_default_k20.json
tail -1 $OUT/bench_default_k20.json
for i in 1 2 3; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-parity --no-live-pmc > $OUT/k20_$i.json 2>/dev/null
  timeout 300 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-also --no-parity --no-live-pmc > $OUT/k200_$i.json 2>/dev/null
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_s1/k2*.json')):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], j['ms_per_step'], j['value'], j['roofline']['frac'], j['roofline']['kernel_us'], j['roofline'].get('kernel_timing'))
PY
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python /root/repo/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-parity --no-live-pmc --no-kernel-timing > /dev/null 2>&1
cd /root/repo
T=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
python scripts/timeline_tail.py $T 36 > $OUT/timeline_k20.txt; cat $OUT/timeline_k20.txt
