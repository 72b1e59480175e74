#!/bin/bash
# round 4, session 38: evidence of the final tree (the GPU suite ran on it in session 37: 231 passed) — smoke, the driver's form of the
# default line with its `also` entries and live PMC traffic, the 200-step line, kernel statistics and PMC passes of configs 3 and 5
OUT=gpurun_out/r04_s38
mkdir -p $OUT
R=/root/repo
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default_k20.json 2> $OUT/bench_default_k20.err
timeout 900 python bench.py --no-also > $OUT/bench_default_k200.json 2> $OUT/bench_default_k200.err
cd /tmp
for c in 3 5; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof$c -- python $R/bench.py --config $c --gpus 1 --sub --no-parity --steps 60 --warmup 5 --no-cpu-baseline > $R/$OUT/prof$c.log 2>&1
  cp $R/$OUT/prof$c/*/*_kernel_stats.csv $R/$OUT/s38_kernel_stats_cfg$c.csv 2>/dev/null
  rm -rf $R/$OUT/prof$c
  for k in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --pmc $k --kernel-trace --output-format csv -d $R/$OUT/pmc_${k}_cfg$c -- python $R/bench.py --config $c --gpus 1 --sub --no-parity --steps 30 --warmup 5 --preheat-ms 0 --no-cpu-baseline > $R/$OUT/pmc_${k}_cfg$c.log 2>&1
  done
  cp $R/$OUT/pmc_FETCH_SIZE_cfg$c/*/*_counter_collection.csv $R/$OUT/s38_cfg${c}_pmc_fetch.csv
  cp $R/$OUT/pmc_WRITE_SIZE_cfg$c/*/*_counter_collection.csv $R/$OUT/s38_cfg${c}_pmc_write.csv
  rm -rf $R/$OUT/pmc_FETCH_SIZE_cfg$c $R/$OUT/pmc_WRITE_SIZE_cfg$c
done
cd $R
python - <<'PY'
import json, glob, os
for f in ['gpurun_out/r04_s38/bench_default_k20.json', 'gpurun_out/r04_s38/bench_default_k200.json']:
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j['ms_per_step'], j['value'], j['roofline']['frac'], [(k['slot'], k['us']) for k in j['roofline']['kernels']])
        print('   traffic', j['roofline'].get('traffic'), j['roofline'].get('traffic_over_algorithmic'), j['roofline'].get('traffic_gbs'), j['roofline'].get('traffic_frac_of_peak'))
        for a in j.get('also', []):
            print('   also', a.get('variant'), a.get('baseline_config'), a.get('frames_per_batch'), a.get('ms_per_step'), a.get('value'), a.get('error'), (a.get('parity') or {}).get('timed_path'), (a.get('parity') or {}).get('failed'))
        if j.get('cpu_baseline'): print('   cpu', j['cpu_baseline']['value'], j['cpu_baseline']['one_thread'], j['cpu_baseline']['cores'])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
for c in 3 5; do echo "== cfg $c"; python scripts/pmc_summary.py $OUT/s38_cfg${c}_pmc_fetch.csv $OUT/s38_cfg${c}_pmc_write.csv | grep -A2 "scan_step\|rows\|cols1024\|plan_long"; done
tail -3 $OUT/smoke.txt | cut -c1-300
