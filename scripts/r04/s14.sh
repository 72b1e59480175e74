#!/bin/bash
# round 4, session 14: 2^20 points — the plan of call k at the front of call k + 1's column launch (k_fft_cols1024_plan) and the
# Hamming taps formed in the column tiles (WCALC): the long-transform tests, then A/B against the separate plan launch / the tap table
OUT=gpurun_out/r04_s14
mkdir -p $OUT
R=/root/repo
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_stated_configs.py tests/test_gpu_cull.py tests/test_gpu_fullsize.py -m gpu -q -x -s --timeout 600 -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1
echo "tests rc=$?" >> $OUT/rc.txt
B="python bench.py --no-cpu-baseline --no-also --no-parity --warmup 5 --diag-lib --gpus 1"
for rep in 1 2; do
  timeout 300 $B --config 5 --steps 100 > $OUT/c5_new_$rep.json 2>> $OUT/ab.err
  SS_PLAN_FUSED=0 timeout 300 $B --config 5 --steps 100 > $OUT/c5_planown_$rep.json 2>> $OUT/ab.err
  SS_WIN_CALC=0 timeout 300 $B --config 5 --steps 100 > $OUT/c5_wintab_$rep.json 2>> $OUT/ab.err
  SS_PLAN_FUSED=0 SS_WIN_CALC=0 timeout 300 $B --config 5 --steps 100 > $OUT/c5_old_$rep.json 2>> $OUT/ab.err
done
timeout 300 $B --config 5 --steps 40 --frames 64 > $OUT/c5x64_new.json 2>> $OUT/ab.err
SS_PLAN_FUSED=0 SS_WIN_CALC=0 timeout 300 $B --config 5 --steps 40 --frames 64 > $OUT/c5x64_old.json 2>> $OUT/ab.err
python - <<'PY'
import glob, json, os
for f in sorted(glob.glob('gpurun_out/r04_s14/c5*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j['ms_per_step'], j['value'], [(k['slot'], k['us']) for k in j['roofline']['kernels']], j['config'].get('tiles'))
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
cat $OUT/rc.txt; tail -6 $OUT/pytest_gpu.txt | cut -c1-300; tail -3 $OUT/ab.err | cut -c1-300
