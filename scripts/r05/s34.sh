#!/bin/bash
# round 5, session 34: the general path of the averaging tiles with its 36 rows worked out on 36 lanes (v_readlane per row) instead of row by row on the scalar unit,
# values by masks — the default line, alternating A/B against the expressions as they were before session 33 (diagnostics builds), stamps, the whole GPU suite
OUT=gpurun_out/r05_s34
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
# (first: does the tree run at all — a memory fault here must not cost the session its minutes)
timeout 600 python -m pytest tests/test_gpu_cull.py tests/test_gpu_parity.py -x -q -m gpu -k "degenerate or ignored or 65536 or halo or short" > $OUT/pytest_first.txt 2>&1 || { tail -30 $OUT/pytest_first.txt; exit 1; }
tail -2 $OUT/pytest_first.txt
for rep in 1 2; do
  for lib in base genold; do
    run default$rep $lib 20
    run default$rep $lib 200
  done
done
for lib in base genold; do
  run c5f16 $lib 100 --config 5 --frames 16 --sub
  run c3cf32 $lib 100 --config 3 --frames 128 --sub --fmt cf32
  run c3cs8 $lib 100 --config 3 --frames 128 --sub
done
SS_STEP_STAMPS=$OUT/stamps_default.txt timeout 300 python bench.py --gpus 1 --no-parity --steps 100 --warmup 5 --no-cpu-baseline --no-also --lib scripts/ab/libspecscan_base.so > $OUT/st_default.json 2> $OUT/st_default.err
python scripts/analyze_step_stamps.py $OUT/stamps_default.txt 32 2>&1 | tee $OUT/stamps_default_summary.txt
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1
tail -4 $OUT/pytest_gpu.txt | cut -c1-300
