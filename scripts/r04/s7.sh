#!/bin/bash
# round 4, session 7: the row tiles as k_scan_step FFT role (KIND 4) with the deferred stages riding on them, the column half a launch of its own
OUT=gpurun_out/r04_s7
mkdir -p $OUT
R=/root/repo
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_stated_configs.py tests/test_gpu_cull.py tests/test_gpu_fullsize.py "tests/test_gpu_parity.py::test_psd_of_a_frame_does_not_depend_on_its_position_in_the_batch" "tests/test_gpu_parity.py::test_alternative_implementations_meet_the_contract" -m gpu -q --timeout 600 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "tests rc=$?" >> $OUT/rc.txt
B="python bench.py --no-cpu-baseline --no-also --no-parity --warmup 5"
for i in 1 2; do
  timeout 300 $B --config 5 --gpus 1 --steps 100 > $OUT/c5_prod$i.json 2>> $OUT/ab.err
  SS_C1024_WIDE=0 timeout 300 $B --diag-lib --config 5 --gpus 1 --steps 100 > $OUT/c5_role$i.json 2>> $OUT/ab.err
  SS_FFT_TWOPASS=0 timeout 300 $B --diag-lib --config 5 --gpus 1 --steps 100 > $OUT/c5_three$i.json 2>> $OUT/ab.err
done
timeout 300 $B --config 5 --gpus 1 --steps 40 --frames 64 > $OUT/c5x64_prod.json 2>> $OUT/ab.err
SS_FFT_TWOPASS=0 timeout 300 $B --diag-lib --config 5 --gpus 1 --steps 40 --frames 64 > $OUT/c5x64_three.json 2>> $OUT/ab.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r04_s7/c*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j['ms_per_step'], j['value'], j['roofline_chain']['frac'], [(k['slot'], k['us']) for k in j['roofline']['kernels']])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
cat $OUT/rc.txt; tail -6 $OUT/pytest.log | cut -c1-300
