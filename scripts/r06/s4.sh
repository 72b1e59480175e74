#!/bin/bash
# round 6, session 4: 262144 points, second form — run maxima as keys by atomic maxima (the column tiles clear them), the plan as the
# fold's decomposition (plan_x256_run, KIND 10), the new row tile in NO_CULL contexts too: tests, A/B against round 2's path and against
# the plan as a launch of its own; the whole GPU suite; the 8192-point parity block with the median / p99 ratios beside the rms ratio
OUT=gpurun_out/r06_s4
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_stated_configs.py -x -q -m gpu -s -k "getfft or retune" > $OUT/pytest_262144_ref.txt 2>&1; tail -2 $OUT/pytest_262144_ref.txt; grep "getFft's own size" $OUT/pytest_262144_ref.txt
timeout 900 python -m pytest tests/test_gpu_cull.py -x -q -m gpu -k "262144 or intermediate" > $OUT/pytest_262144_cull.txt 2>&1; tail -4 $OUT/pytest_262144_cull.txt
for i in 1 2; do
  timeout 300 python bench.py --config 5 --gpus 1 --fft 262144 --frames 32 --steps 60 --warmup 5 --preheat-ms 150 --no-cpu-baseline --no-parity --sub --diag-lib > $OUT/x256_new_$i.json 2>/dev/null
  SS_ROWS1024X256=0 timeout 300 python bench.py --config 5 --gpus 1 --fft 262144 --frames 32 --steps 60 --warmup 5 --preheat-ms 150 --no-cpu-baseline --no-parity --sub --diag-lib > $OUT/x256_old_$i.json 2>/dev/null
  SS_PLAN_FUSED=0 timeout 300 python bench.py --config 5 --gpus 1 --fft 262144 --frames 32 --steps 60 --warmup 5 --preheat-ms 150 --no-cpu-baseline --no-parity --sub --diag-lib > $OUT/x256_planown_$i.json 2>/dev/null
done
timeout 300 python bench.py --config 5 --gpus 1 --fft 262144 --frames 64 --steps 40 --warmup 5 --preheat-ms 150 --no-cpu-baseline --no-parity --sub > $OUT/x256_prod_f64.json 2>/dev/null
timeout 300 python bench.py --config 5 --gpus 1 --fft 262144 --frames 16 --steps 80 --warmup 5 --preheat-ms 150 --no-cpu-baseline --no-parity --sub > $OUT/x256_prod_f16.json 2>/dev/null
timeout 400 python bench.py --config 5 --gpus 1 --fft 262144 --frames 32 --steps 60 --warmup 5 --preheat-ms 150 --no-cpu-baseline --sub > $OUT/x256_prod_parity.json 2>/dev/null
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-live-pmc --sub > $OUT/cfg2_parity.json 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_s4/*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        p = j.get('parity') or {}
        ab = p.get('all_bins_vs_fp64_fft_dB') or {}
        print(f.split('/')[-1], j['ms_per_step'], j['value'], [(k['slot'], k['us']) for k in j['roofline']['kernels']], j['config']['tile_culling'], j['config']['tiles']['evaluated_frac'],
              'parity', p.get('failed') or {k: p.get(k) for k in ('reference_candidates', 'inside_1e-3_dB_band')}, {k: ab.get(k) for k in ('engine_over_reference_rms', 'engine_over_reference_median', 'engine_over_reference_p99')}, (p.get('timed_path') or {}).get('tiles_culled'))
        if ab: print('    ', ab.get('engine'), ab.get('reference'))
    except Exception as e:
        print(f, 'ERR', e)
PY
# the plan role's copy of its columns' maxima in one flight (sixteen loads) against round 5's two trips of eight: alternating runs
for i in 1 2 3; do
  for v in 16 8; do
    L="--diag-lib"; [ $v = 8 ] && L="--lib scripts/ab/libspecscan_plancopy8.so"
    timeout 300 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-also --no-parity --no-live-pmc $L > $OUT/plan${v}_k200_$i.json 2>/dev/null
    timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-parity --no-live-pmc $L > $OUT/plan${v}_k20_$i.json 2>/dev/null
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_s4/plan*.json')):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], j['ms_per_step'], j['value'], j['roofline']['frac'], j['roofline']['kernel_us'])
PY
timeout 120 scripts/ubench/mfma_fold_lab > $OUT/mfma_fold_lab.txt 2>&1; cat $OUT/mfma_fold_lab.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
