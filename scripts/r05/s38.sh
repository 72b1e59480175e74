#!/bin/bash
# round 5, session 38: the driver's form of the default line, complete (its `also` entries, live PMC traffic, CPU baseline, parity), on the
# final binaries (those of session 37's 243 green tests); the alternatives test with the new switch
OUT=gpurun_out/r05_s38
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "alternative_implementations and (HALO or PLAN_FIRST)" > $OUT/pytest_alt.txt 2>&1; tail -2 $OUT/pytest_alt.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default_k20.json 2> $OUT/bench_default_k20.err
python - <<'PY'
import json
j = json.loads(open('gpurun_out/r05_s38/bench_default_k20.json').read().strip().splitlines()[-1])
print('default k20', j['ms_per_step'], j['value'], j['roofline']['frac'], j['roofline'].get('traffic'), 'parity failed', (j.get('parity') or {}).get('failed'))
for a in j.get('also', []):
    print('   also', a.get('variant'), a.get('baseline_config'), a.get('frames_per_batch'), a.get('ms_per_step'), a.get('value'), a.get('error') or '', (a.get('roofline_chain') or {}).get('pmc_bytes_per_sample_from_profiles'), 'parity failed', (a.get('parity') or {}).get('failed'), {k: a[k] for k in ('ss_process_MSps', 'ss_feed_MSps') if k in a})
print('cpu', (j.get('cpu_baseline') or {}).get('value'))
PY
