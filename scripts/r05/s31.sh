#!/bin/bash
# round 5, session 31: the fold with its rows in blocks, one-flight tiles, the plan behind the fold, pairs behind the first workgroups — the
# whole GPU suite, then configs 3 (128 / 256 / 512 frames), 5 (16 / 64 frames) and 131072 points
OUT=gpurun_out/r05_s31
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1
tail -6 $OUT/pytest_gpu.txt | cut -c1-300
run() {  # config frames extra...
  c=$1; f=$2; shift 2
  timeout 300 python bench.py --config $c --frames $f --gpus 1 --sub --steps 100 --warmup 5 --no-cpu-baseline "$@" > $OUT/bench_c${c}_f$f$3.json 2> $OUT/bench_c${c}_f$f$3.err
  python - <<PY
import json
try:
    j = json.loads(open('$OUT/bench_c${c}_f$f$3.json').read().strip().splitlines()[-1])
    p = j.get('parity') or {}
    print('cfg $c f=$f $*', j['ms_per_step'], j['value'], [(k['slot'], k['us']) for k in j['roofline']['kernels']], 'parity', p.get('failed'), (p.get('timed_path') or {}).get('reference_candidates'))
except Exception as e:
    print('cfg $c f=$f ERR', e, open('$OUT/bench_c${c}_f$f$3.err').read()[-500:])
PY
}
run 3 128
run 3 256
run 3 512
run 5 16
run 5 64
run 3 64 --fft 131072
run 3 256 --fft 131072
run 3 64
run 3 16
run 3 128 --fft 131072
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also > $OUT/bench_default_k20.json 2> $OUT/bench_default_k20.err
python - <<PY
import json
j = json.loads(open('$OUT/bench_default_k20.json').read().strip().splitlines()[-1])
print('default k20', j['ms_per_step'], j['value'], j['roofline']['frac'])
PY
