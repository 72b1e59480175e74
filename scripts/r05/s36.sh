#!/bin/bash
# round 5, session 36: 8192 points, deep pipelining — the re-transformed halo frames leave per-column maxima, so that the 64 tiles of a
# batch's first two frame tiles are tested like the others: the culling tests first (culled == unculled, and against the reference), then
# the default line with and without (SS_HALO_MAXIMA=0, diagnostics build), alternating; stamps; the whole GPU suite
OUT=gpurun_out/r05_s36
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_cull.py tests/test_gpu_stated_configs.py -x -q -m gpu -k "culled_lists_equal or deep_pipelined or threshold or ignored or short_calls or config2 or config1" > $OUT/pytest_first.txt 2>&1 || { tail -30 $OUT/pytest_first.txt; exit 1; }
tail -2 $OUT/pytest_first.txt
run() {  # tag steps env...
  tag=$1; k=$2; shift 2
  env "$@" timeout 300 python bench.py --steps $k --warmup 5 --no-cpu-baseline --no-also --no-parity --no-live-pmc --diag-lib > $OUT/${tag}_k$k.json 2> $OUT/${tag}_k$k.err
  python - <<PY
import json
try:
    j = json.loads(open('$OUT/${tag}_k$k.json').read().strip().splitlines()[-1])
    print('$tag k=$k', j['ms_per_step'], j['value'], j['roofline']['frac'], j['config'].get('tiles'))
except Exception as e:
    print('$tag k=$k ERR', e, open('$OUT/${tag}_k$k.err').read()[-400:])
PY
}
for rep in 1 2 3; do
  run new$rep 20 SS_X=0
  run new$rep 200 SS_X=0
  run old$rep 20 SS_HALO_MAXIMA=0
  run old$rep 200 SS_HALO_MAXIMA=0
done
SS_STEP_STAMPS=$OUT/stamps_default.txt timeout 300 python bench.py --gpus 1 --no-parity --steps 100 --warmup 5 --no-cpu-baseline --no-also --diag-lib > $OUT/st_default.json 2> $OUT/st_default.err
python scripts/analyze_step_stamps.py $OUT/stamps_default.txt 32 2>&1 | tee $OUT/stamps_default_summary.txt
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1
tail -4 $OUT/pytest_gpu.txt | cut -c1-300
