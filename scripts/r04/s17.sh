#!/bin/bash
# round 4, session 17: 65536 points in the shape of the 2^20-point chain — the column half as a launch of its own with the plan of the
# call before at its front, the row tiles as k_scan_step's FFT role (KIND 6) carrying the deferred stages, tile culling on — against
# the two older forms; the whole GPU suite on this tree
OUT=gpurun_out/r04_s17
mkdir -p $OUT
R=/root/repo
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1
echo "tests rc=$?" >> $OUT/rc.txt
B="python bench.py --no-cpu-baseline --no-also --no-parity --warmup 5 --diag-lib --gpus 1"
for rep in 1 2; do
  timeout 300 $B --config 3 --steps 100 > $OUT/c3_new_$rep.json 2>> $OUT/ab.err
  SS_ROWS256_STEP=0 timeout 300 $B --config 3 --steps 100 > $OUT/c3_cullold_$rep.json 2>> $OUT/ab.err
  SS_CULL_65536=0 timeout 300 $B --config 3 --steps 100 > $OUT/c3_nocull_$rep.json 2>> $OUT/ab.err
done
timeout 300 $B --config 3 --steps 400 --frames 16 > $OUT/c3_f16_new.json 2>> $OUT/ab.err
SS_CULL_65536=0 timeout 300 $B --config 3 --steps 400 --frames 16 > $OUT/c3_f16_nocull.json 2>> $OUT/ab.err
timeout 300 $B --config 3 --steps 100 --frames 64 > $OUT/c3_f64_new.json 2>> $OUT/ab.err
SS_CULL_65536=0 timeout 300 $B --config 3 --steps 100 --frames 64 > $OUT/c3_f64_nocull.json 2>> $OUT/ab.err
python - <<'PY'
import glob, json, os
for f in sorted(glob.glob('gpurun_out/r04_s17/c*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j['ms_per_step'], j['value'], [(k['slot'], k['us']) for k in j['roofline']['kernels']], j['config'].get('tiles'))
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
cat $OUT/rc.txt; tail -8 $OUT/pytest_gpu.txt | cut -c1-400; tail -3 $OUT/ab.err | cut -c1-300
