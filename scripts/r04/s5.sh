#!/bin/bash
# round 4, session 5: the window taps in the column tiles' own order; column tiles of 16 columns by 1024 threads (SS_C1024_WIDE=1)
OUT=gpurun_out/r04_s5
mkdir -p $OUT
R=/root/repo
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_stated_configs.py tests/test_gpu_cull.py "tests/test_gpu_parity.py::test_psd_of_a_frame_does_not_depend_on_its_position_in_the_batch" -m gpu -q --timeout 600 -p no:cacheprovider > $OUT/pytest_prod.log 2>&1
echo "prod rc=$?" >> $OUT/rc.txt
SS_TEST_USE_DIAG_LIB=1 SS_C1024_WIDE=1 timeout 900 python -m pytest tests/test_gpu_stated_configs.py tests/test_gpu_cull.py -m gpu -q --timeout 600 -p no:cacheprovider -k "config5 or million or 1048576 or long" > $OUT/pytest_wide.log 2>&1
echo "wide rc=$?" >> $OUT/rc.txt
B="python bench.py --no-cpu-baseline --no-also --no-parity --warmup 5"
for i in 1 2; do
  timeout 300 $B --config 5 --gpus 1 --steps 100 > $OUT/c5_two$i.json 2>> $OUT/ab.err
  SS_C1024_WIDE=1 timeout 300 $B --diag-lib --config 5 --gpus 1 --steps 100 > $OUT/c5_wide$i.json 2>> $OUT/ab.err
  SS_FFT_TWOPASS=0 timeout 300 $B --diag-lib --config 5 --gpus 1 --steps 100 > $OUT/c5_three$i.json 2>> $OUT/ab.err
done
timeout 300 $B --config 5 --gpus 1 --steps 40 --frames 64 > $OUT/c5x64_two.json 2>> $OUT/ab.err
SS_C1024_WIDE=1 timeout 300 $B --diag-lib --config 5 --gpus 1 --steps 40 --frames 64 > $OUT/c5x64_wide.json 2>> $OUT/ab.err
SS_FFT_TWOPASS=0 timeout 300 $B --diag-lib --config 5 --gpus 1 --steps 40 --frames 64 > $OUT/c5x64_three.json 2>> $OUT/ab.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r04_s5/c5*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j['ms_per_step'], j['value'], j['roofline_chain']['frac'], [(k['slot'], k['us']) for k in j['roofline']['kernels']])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
cat $OUT/rc.txt; tail -4 $OUT/pytest_prod.log | cut -c1-300; tail -12 $OUT/pytest_wide.log | cut -c1-300
