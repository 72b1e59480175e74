#!/bin/bash
# round 4, session 3: 2^20 points in two passes (fft1024_kernels.h) — the tests that touch it first, then the whole suite, then
# config 5 against round 3's three-pass form in one session
OUT=gpurun_out/r04_s3
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1
timeout 1500 python -m pytest tests/test_gpu_stated_configs.py tests/test_gpu_stream_ordered.py tests/test_gpu_degenerate_input.py -m gpu -q -s --timeout 900 -p no:cacheprovider > $OUT/pytest_first.log 2>&1
echo "first rc=$?" >> $OUT/rc.txt
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider --deselect tests/test_gpu_stated_configs.py --deselect tests/test_gpu_stream_ordered.py --deselect tests/test_gpu_degenerate_input.py > $OUT/pytest_rest.log 2>&1
echo "rest rc=$?" >> $OUT/rc.txt
B="python bench.py --no-cpu-baseline --no-also --no-parity --warmup 5 --diag-lib"
for i in 1 2; do
  timeout 300 $B --config 5 --gpus 1 --steps 100 > $OUT/c5_two$i.json 2>> $OUT/ab.err
  SS_FFT_TWOPASS=0 timeout 300 $B --config 5 --gpus 1 --steps 100 > $OUT/c5_three$i.json 2>> $OUT/ab.err
  SS_RING_ONLY=0 timeout 300 $B --config 5 --gpus 1 --steps 100 > $OUT/c5_two_psd$i.json 2>> $OUT/ab.err
done
timeout 300 $B --config 5 --gpus 1 --steps 40 --frames 64 > $OUT/c5x64_two.json 2>> $OUT/ab.err
SS_FFT_TWOPASS=0 timeout 300 $B --config 5 --gpus 1 --steps 40 --frames 64 > $OUT/c5x64_three.json 2>> $OUT/ab.err
timeout 300 $B --config 5 --gpus 1 --steps 100 --no-cull > $OUT/c5_two_nocull.json 2>> $OUT/ab.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r04_s3/c5*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j['ms_per_step'], j['value'], j['roofline_chain']['frac'], [(k['slot'], k['us']) for k in j['roofline']['kernels']], j['config'].get('tiles'))
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
cat $OUT/rc.txt
grep -v "^  File\|^$" $OUT/pytest_first.log | grep -n "passed\|failed\|FAILED\|Error\|^E " | head -40 | cut -c1-300
tail -5 $OUT/pytest_rest.log | cut -c1-300
