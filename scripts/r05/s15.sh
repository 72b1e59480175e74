#!/bin/bash
# round 5, session 15: 131072 points through the radix-16 fold in the product (KIND 9) — first contact: the steady-state and getFft
# tests against the reference, then the line
OUT=gpurun_out/r05_s15
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_stated_configs.py -m gpu -x -q -s -k "steady_state or getfft" > $OUT/pytest.txt 2>&1
grep "^\[config\|^\[getFft\|passed\|failed\|Error\|error" $OUT/pytest.txt | cut -c1-330
for f in 64 128 256; do
  timeout 300 python bench.py --config 3 --fft 131072 --frames $f --gpus 1 --sub --steps 60 --warmup 5 --no-cpu-baseline > $OUT/bench_131072_f$f.json 2> $OUT/bench_131072_f$f.err
  python - <<PY
import json
try:
    j = json.loads(open('$OUT/bench_131072_f$f.json').read().strip().splitlines()[-1])
    print('131072 f=$f', j['ms_per_step'], j['value'], [(k['slot'], k['us'], k['frames_per_launch']) for k in j['roofline']['kernels']], j['config']['tiles'])
    p = j.get('parity') or {}
    print('    parity', p.get('failed'), p.get('timed_path'), (p.get('all_bins_vs_fp64_fft_dB') or {}).get('engine_over_reference_rms'))
except Exception as e:
    print('131072 f=$f ERR', e, open('$OUT/bench_131072_f$f.err').read()[-800:])
PY
done
