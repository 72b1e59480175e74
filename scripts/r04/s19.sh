#!/bin/bash
# round 4, session 19: the first 64 pairs of a long transform's plan on detect workgroups of their own, dispatched ahead of the launch's
# FFT role (SS_LIST_FIRST=0: every pair behind an FFT workgroup's tile, the launch's tail) — configs 3 and 5; long-transform tests
OUT=gpurun_out/r04_s19
mkdir -p $OUT
R=/root/repo
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_stated_configs.py tests/test_gpu_cull.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_degenerate_input.py tests/test_gpu_stream_ordered.py -m gpu -q -x --timeout 900 -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1
echo "tests rc=$?" >> $OUT/rc.txt
B="python bench.py --no-cpu-baseline --no-also --no-parity --warmup 5 --diag-lib --gpus 1"
for rep in 1 2; do
  timeout 300 $B --config 3 --steps 100 > $OUT/c3_new_$rep.json 2>> $OUT/ab.err
  SS_LIST_FIRST=0 timeout 300 $B --config 3 --steps 100 > $OUT/c3_first0_$rep.json 2>> $OUT/ab.err
  SS_CULL_65536=0 timeout 300 $B --config 3 --steps 100 > $OUT/c3_nocull_$rep.json 2>> $OUT/ab.err
  timeout 300 $B --config 5 --steps 100 > $OUT/c5_new_$rep.json 2>> $OUT/ab.err
  SS_LIST_FIRST=0 timeout 300 $B --config 5 --steps 100 > $OUT/c5_first0_$rep.json 2>> $OUT/ab.err
done
timeout 300 $B --config 3 --steps 400 --frames 16 > $OUT/c3_f16_new.json 2>> $OUT/ab.err
SS_CULL_65536=0 timeout 300 $B --config 3 --steps 400 --frames 16 > $OUT/c3_f16_nocull.json 2>> $OUT/ab.err
timeout 300 $B --config 3 --steps 100 --frames 64 > $OUT/c3_f64_new.json 2>> $OUT/ab.err
SS_CULL_65536=0 timeout 300 $B --config 3 --steps 100 --frames 64 > $OUT/c3_f64_nocull.json 2>> $OUT/ab.err
timeout 300 $B --config 5 --steps 40 --frames 64 > $OUT/c5x64_new.json 2>> $OUT/ab.err
python - <<'PY'
import glob, json, os
for f in sorted(glob.glob('gpurun_out/r04_s19/c*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j['ms_per_step'], j['value'], [(k['slot'], k['us']) for k in j['roofline']['kernels']], j['config'].get('tiles'))
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
cat $OUT/rc.txt; tail -8 $OUT/pytest_gpu.txt | cut -c1-400; tail -3 $OUT/ab.err | cut -c1-300
