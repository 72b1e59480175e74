#!/bin/bash
# round 4, session 23: 8192 points — the first pairs of every list of a launch's tile plan on detect workgroups of their own, ahead of
# the FFT role (SS_PLAN_FIRST=n) against every pair behind an FFT workgroup's frame (what ships)
OUT=gpurun_out/r04_s23
mkdir -p $OUT
R=/root/repo
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
SS_TEST_USE_DIAG_LIB=1 SS_PLAN_FIRST=32 timeout 900 python -m pytest tests/test_gpu_cull.py tests/test_gpu_step_pipeline.py tests/test_gpu_wait_bound.py -m gpu -q -x --timeout 900 -p no:cacheprovider -k "not 65536 and not long" > $OUT/pytest_gpu.txt 2>&1
echo "tests rc=$?" >> $OUT/rc.txt
B="python bench.py --no-cpu-baseline --no-also --no-parity --no-live-pmc --diag-lib --gpus 1"
for rep in 1 2 3; do
  for pf in 0 16 32 64; do
    SS_PLAN_FIRST=$pf timeout 300 $B --steps 200 --warmup 20 > $OUT/c2_pf${pf}_k200_$rep.json 2>> $OUT/ab.err
  done
done
for rep in 1 2; do
  for pf in 0 32; do
    SS_PLAN_FIRST=$pf timeout 300 $B --steps 20 --warmup 5 > $OUT/c2_pf${pf}_k20_$rep.json 2>> $OUT/ab.err
  done
done
python - <<'PY'
import glob, json, os
for f in sorted(glob.glob('gpurun_out/r04_s23/c2*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j['ms_per_step'], j['value'], j['roofline']['frac'], [(k['slot'], k['us']) for k in j['roofline']['kernels']], j['config'].get('tiles', {}).get('wait_fallbacks'))
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
cat $OUT/rc.txt; tail -6 $OUT/pytest_gpu.txt | cut -c1-400; tail -3 $OUT/ab.err | cut -c1-300
