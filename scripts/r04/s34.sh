#!/bin/bash
# round 4, session 34: 65536 points, ONE launch per call (SS_MERGE_65536=1 of the diagnostics build: the column half of call k beside the
# row half of call k - 1, the plan of call k - 2, detect(k - 3), emit(k - 4); scan_step.h KIND 7) against the two-launch form that ships
OUT=gpurun_out/r04_s34
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
SS_TEST_USE_DIAG_LIB=1 SS_MERGE_65536=1 timeout 900 python -m pytest tests/test_gpu_cull.py tests/test_gpu_stated_configs.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "65536 or long_rows or config3 or random" > $OUT/pytest_merge.txt 2>&1
echo "merge tests rc=$?" >> $OUT/rc.txt
B="python bench.py --no-cpu-baseline --no-also --no-parity --warmup 5 --gpus 1 --config 3 --diag-lib"
for rep in 1 2; do
  SS_MERGE_65536=1 timeout 300 $B --steps 100 > $OUT/c3_merge_$rep.json 2>> $OUT/ab.err
  timeout 300 $B --steps 100 > $OUT/c3_ships_$rep.json 2>> $OUT/ab.err
done
SS_MERGE_65536=1 timeout 300 $B --steps 200 --frames 64 > $OUT/c3_f64_merge.json 2>> $OUT/ab.err
timeout 300 $B --steps 200 --frames 64 > $OUT/c3_f64_ships.json 2>> $OUT/ab.err
SS_MERGE_65536=1 timeout 300 $B --steps 50 --frames 256 > $OUT/c3_f256_merge.json 2>> $OUT/ab.err
timeout 300 $B --steps 50 --frames 256 > $OUT/c3_f256_ships.json 2>> $OUT/ab.err
python - <<'PY'
import glob, json, os
for f in sorted(glob.glob('gpurun_out/r04_s34/c3_*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j['ms_per_step'], j['value'], [(k['slot'], k['us']) for k in j['roofline']['kernels']], j['config'].get('tiles'))
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
cat $OUT/rc.txt; tail -5 $OUT/pytest_merge.txt | cut -c1-500; tail -3 $OUT/ab.err | cut -c1-300
