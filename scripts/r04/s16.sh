#!/bin/bash
# round 4, session 16: the row tiles of the long transforms with their ceiling loads in flight together (the load / store loop at the
# end of every workgroup was sixteen memory round trips one after the other); 65536 points: Hamming taps formed in the column tiles
OUT=gpurun_out/r04_s16
mkdir -p $OUT
R=/root/repo
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_stated_configs.py tests/test_gpu_cull.py tests/test_gpu_fullsize.py -m gpu -q -x -s --timeout 600 -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1
echo "tests rc=$?" >> $OUT/rc.txt
B="python bench.py --no-cpu-baseline --no-also --no-parity --warmup 5 --diag-lib --gpus 1"
for rep in 1 2; do
  timeout 300 $B --config 5 --steps 100 > $OUT/c5_new_$rep.json 2>> $OUT/ab.err
  timeout 300 $B --config 3 --steps 100 > $OUT/c3_new_$rep.json 2>> $OUT/ab.err
  SS_WIN_CALC=0 timeout 300 $B --config 3 --steps 100 > $OUT/c3_wintab_$rep.json 2>> $OUT/ab.err
  SS_CULL_65536=1 timeout 300 $B --config 3 --steps 100 > $OUT/c3_cull_$rep.json 2>> $OUT/ab.err
done
timeout 300 $B --config 5 --steps 40 --frames 64 > $OUT/c5x64_new.json 2>> $OUT/ab.err
timeout 300 $B --config 3 --steps 400 --frames 16 > $OUT/c3_f16_new.json 2>> $OUT/ab.err
SS_CULL_65536=1 timeout 300 $B --config 3 --steps 400 --frames 16 > $OUT/c3_f16_cull.json 2>> $OUT/ab.err
python - <<'PY'
import glob, json, os
for f in sorted(glob.glob('gpurun_out/r04_s16/c*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j['ms_per_step'], j['value'], [(k['slot'], k['us']) for k in j['roofline']['kernels']], j['config'].get('tiles'))
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
cat $OUT/rc.txt; tail -6 $OUT/pytest_gpu.txt | cut -c1-300; tail -3 $OUT/ab.err | cut -c1-300
