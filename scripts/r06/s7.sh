#!/bin/bash
# round 6, session 7: (a) how a pageable host buffer reaches the device fastest (scripts/pageable_copy_lab.py: ss_process at 2^20 points
# still spends 24 of its 26.6 ms in the copy); (b) 262144 points, ONE launch per call (KIND 12): tests, then its rate against the two-launch
# form (SS_MERGE_65536=0 on the diagnostics build), alternating
OUT=gpurun_out/r06_s7
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
timeout 300 python scripts/pageable_copy_lab.py > $OUT/pageable_copy_lab.txt 2>&1; cat $OUT/pageable_copy_lab.txt
timeout 900 python -m pytest tests/test_gpu_stated_configs.py -x -q -m gpu -s -k "getfft or retune" > $OUT/pytest_262144_ref.txt 2>&1; tail -2 $OUT/pytest_262144_ref.txt; grep "getFft's own size" $OUT/pytest_262144_ref.txt
timeout 900 python -m pytest tests/test_gpu_cull.py -x -q -m gpu -k "262144 or intermediate or random" > $OUT/pytest_262144_cull.txt 2>&1; tail -4 $OUT/pytest_262144_cull.txt
run() { timeout 300 python bench.py --gpus 1 --warmup 5 --preheat-ms 150 --no-cpu-baseline --no-parity --sub --diag-lib ${@:2} > $OUT/$1.json 2>/dev/null; }
for i in 1 2; do
  run x256_f32_one_$i --config 5 --fft 262144 --frames 32 --steps 60
  SS_MERGE_65536=0 run x256_f32_two_$i --config 5 --fft 262144 --frames 32 --steps 60
  run x256_f16_one_$i --config 5 --fft 262144 --frames 16 --steps 80
  SS_MERGE_65536=0 run x256_f16_two_$i --config 5 --fft 262144 --frames 16 --steps 80
  run x256_f64_$i --config 5 --fft 262144 --frames 64 --steps 40
done
timeout 400 python bench.py --config 5 --gpus 1 --fft 262144 --frames 32 --steps 60 --warmup 5 --preheat-ms 150 --no-cpu-baseline --sub > $OUT/x256_prod_parity.json 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_s7/*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        p = j.get('parity') or {}
        print(f.split('/')[-1], j['ms_per_step'], j['value'], [(k['slot'], k['us']) for k in j['roofline']['kernels']], j['config']['tile_culling'], j['config']['tiles']['evaluated_frac'],
              p.get('failed') or {k: p.get(k) for k in ('reference_candidates', 'inside_1e-3_dB_band')}, (p.get('timed_path') or {}).get('tiles_culled'))
    except Exception as e:
        print(f, 'ERR', e)
PY
