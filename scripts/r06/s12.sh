#!/bin/bash
# round 6, session 12: state of the tree after the drop-in fix — whole GPU suite, smoke, the driver's form of the default line (compact last line +
# bench_full.json: no event-timed launch inside any timed region now, `also` with the butterfly folds and the culled 262144-point chain)
OUT=gpurun_out/r06_s12
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default_k20.json 2> $OUT/bench_default_k20.err ) 2> $OUT/bench_default_k20.time
cp bench_full.json $OUT/bench_full_k20.json 2>/dev/null
tail -c 200 $OUT/bench_default_k20.time; wc -c $OUT/bench_default_k20.json; tail -1 $OUT/bench_default_k20.json
for i in 1 2 3; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-parity --no-live-pmc > $OUT/k20_$i.json 2>/dev/null
  timeout 300 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-also --no-parity --no-live-pmc > $OUT/k200_$i.json 2>/dev/null
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_s12/k2*.json')):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], j['ms_per_step'], j['value'], j['roofline']['frac'], j['roofline']['kernel_us'])
PY
