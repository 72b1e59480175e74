#!/bin/bash
# round 6, session 3: 262144 points through the culled path (256-point column tiles + the 1024-point row tile, plan layout 3, the
# 65536-point two-launch pipeline): parity against the reference's own code, culled == unculled, the retune test; its rate against
# round 2's path in one session (SS_ROWS1024X256=0 on the diagnostics build); config 3's parity block with the butterfly fold
OUT=gpurun_out/r06_s3
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_stated_configs.py -x -q -m gpu -s -k "getfft or retune" > $OUT/pytest_262144_ref.txt 2>&1; tail -4 $OUT/pytest_262144_ref.txt; grep "getFft's own size" $OUT/pytest_262144_ref.txt
timeout 900 python -m pytest tests/test_gpu_cull.py -x -q -m gpu -k "262144 or intermediate" > $OUT/pytest_262144_cull.txt 2>&1; tail -4 $OUT/pytest_262144_cull.txt
for i in 1 2; do
  timeout 300 python bench.py --config 5 --gpus 1 --fft 262144 --frames 32 --steps 60 --warmup 5 --preheat-ms 150 --no-cpu-baseline --no-parity --sub --diag-lib > $OUT/x256_new_$i.json 2>/dev/null
  SS_ROWS1024X256=0 timeout 300 python bench.py --config 5 --gpus 1 --fft 262144 --frames 32 --steps 60 --warmup 5 --preheat-ms 150 --no-cpu-baseline --no-parity --sub --diag-lib > $OUT/x256_old_$i.json 2>/dev/null
  SS_PLAN_FUSED=0 timeout 300 python bench.py --config 5 --gpus 1 --fft 262144 --frames 32 --steps 60 --warmup 5 --preheat-ms 150 --no-cpu-baseline --no-parity --sub --diag-lib > $OUT/x256_planown_$i.json 2>/dev/null
done
timeout 300 python bench.py --config 5 --gpus 1 --fft 262144 --frames 64 --steps 40 --warmup 5 --preheat-ms 150 --no-cpu-baseline --no-parity --sub > $OUT/x256_prod_f64.json 2>/dev/null
timeout 300 python bench.py --config 5 --gpus 1 --fft 262144 --frames 16 --steps 80 --warmup 5 --preheat-ms 150 --no-cpu-baseline --no-parity --sub > $OUT/x256_prod_f16.json 2>/dev/null
timeout 400 python bench.py --config 5 --gpus 1 --fft 262144 --frames 32 --steps 60 --warmup 5 --preheat-ms 150 --no-cpu-baseline --sub > $OUT/x256_prod_parity.json 2>/dev/null
timeout 400 python bench.py --config 3 --gpus 1 --steps 100 --warmup 5 --preheat-ms 150 --no-cpu-baseline --sub > $OUT/cfg3_prod_parity.json 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_s3/*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        p = j.get('parity') or {}
        print(f.split('/')[-1], j['ms_per_step'], j['value'], [(k['slot'], k['us']) for k in j['roofline']['kernels']], j['config']['tile_culling'], j['config']['tiles']['evaluated_frac'],
              'parity', p.get('failed') or {k: p.get(k) for k in ('reference_candidates', 'inside_1e-3_dB_band')}, (p.get('all_bins_vs_fp64_fft_dB') or {}).get('engine_over_reference_rms'), (p.get('timed_path') or {}).get('tiles_culled'))
    except Exception as e:
        print(f, 'ERR', e)
PY
