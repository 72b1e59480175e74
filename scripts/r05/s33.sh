#!/bin/bash
# round 5, session 33: the averaging tiles' general path with fewer instructions per row (address: one clamp and one multiply; value: masks,
# no branches) — the default line, alternating A/B against the expressions as they were (diagnostics builds), stamps, then the whole GPU suite
OUT=gpurun_out/r05_s33
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
run() {  # tag lib steps extra
  tag=$1; lib=$2; k=$3; shift 3
  timeout 300 python bench.py --steps $k --warmup 5 --no-cpu-baseline --no-also --no-parity --no-live-pmc --lib scripts/ab/libspecscan_$lib.so "$@" > $OUT/${tag}_${lib}_k$k.json 2> $OUT/${tag}_${lib}_k$k.err
  python - <<PY
import json
try:
    j = json.loads(open('$OUT/${tag}_${lib}_k$k.json').read().strip().splitlines()[-1])
    print('$tag $lib k=$k', j['ms_per_step'], j['value'], j['roofline']['frac'], j['roofline'].get('kernel_us'))
except Exception as e:
    print('$tag $lib k=$k ERR', e, open('$OUT/${tag}_${lib}_k$k.err').read()[-400:])
PY
}
for rep in 1 2 3; do
  for lib in base genold; do
    run default$rep $lib 20
    run default$rep $lib 200
  done
done
for lib in base genold; do
  run c5f16 $lib 100 --config 5 --frames 16 --sub
  run c3cf32 $lib 100 --config 3 --frames 128 --sub --fmt cf32
done
SS_STEP_STAMPS=$OUT/stamps_default.txt timeout 300 python bench.py --gpus 1 --no-parity --steps 100 --warmup 5 --no-cpu-baseline --no-also --lib scripts/ab/libspecscan_base.so > $OUT/st_default.json 2> $OUT/st_default.err
python scripts/analyze_step_stamps.py $OUT/stamps_default.txt 32 2>&1 | tee $OUT/stamps_default_summary.txt
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1
tail -4 $OUT/pytest_gpu.txt | cut -c1-300
