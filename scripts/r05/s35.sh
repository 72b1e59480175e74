#!/bin/bash
# round 5, session 35: evidence of the final tree (the binaries session 34 ran the whole GPU suite on: 243 passed) — PMC passes (FETCH_SIZE /
# WRITE_SIZE, separate passes) and kernel statistics of configs 3 and 5 as they ship and of config 3 in 512-frame calls, smoke, the driver's
# form of the default line (20 steps: `also` entries, live PMC traffic, CPU baseline), the 200-step line, rocprofv3 kernel statistics +
# launch overlap of the default command
OUT=gpurun_out/r05_s35
mkdir -p $OUT
R=/root/repo
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
cd /tmp
for c in 3 5; do
  for k in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --pmc $k --kernel-trace --output-format csv -d $R/$OUT/pmc_${k}_cfg$c -- python $R/bench.py --config $c --gpus 1 --sub --no-parity --steps 30 --warmup 5 --preheat-ms 0 --no-cpu-baseline > $R/$OUT/pmc_${k}_cfg$c.log 2>&1
  done
  cp $R/$OUT/pmc_FETCH_SIZE_cfg$c/*/*_counter_collection.csv $R/$OUT/s35_cfg${c}_pmc_fetch.csv
  cp $R/$OUT/pmc_WRITE_SIZE_cfg$c/*/*_counter_collection.csv $R/$OUT/s35_cfg${c}_pmc_write.csv
  cp $R/$OUT/s35_cfg${c}_pmc_fetch.csv $R/$OUT/s35_cfg${c}_pmc_write.csv $R/profiles/r05/   # (what bench.py's PMC_SET reads, for the lines below)
  rm -rf $R/$OUT/pmc_FETCH_SIZE_cfg$c $R/$OUT/pmc_WRITE_SIZE_cfg$c
done
for k in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $k --kernel-trace --output-format csv -d $R/$OUT/pmc_${k}_cfg3f512 -- python $R/bench.py --config 3 --frames 512 --gpus 1 --sub --no-parity --steps 20 --warmup 5 --preheat-ms 0 --no-cpu-baseline > $R/$OUT/pmc_${k}_cfg3f512.log 2>&1
done
cp $R/$OUT/pmc_FETCH_SIZE_cfg3f512/*/*_counter_collection.csv $R/$OUT/s35_cfg3f512_pmc_fetch.csv
cp $R/$OUT/pmc_WRITE_SIZE_cfg3f512/*/*_counter_collection.csv $R/$OUT/s35_cfg3f512_pmc_write.csv
rm -rf $R/$OUT/pmc_FETCH_SIZE_cfg3f512 $R/$OUT/pmc_WRITE_SIZE_cfg3f512
cd $R
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default_k20.json 2> $OUT/bench_default_k20.err
timeout 900 python bench.py --no-also > $OUT/bench_default_k200.json 2> $OUT/bench_default_k200.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof2 -- python $R/bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-also --no-parity --no-live-pmc > $R/$OUT/prof2.log 2>&1
cp $R/$OUT/prof2/*/*_kernel_stats.csv $R/$OUT/s35_kernel_stats.csv 2>/dev/null
python $R/scripts/launches_in_flight.py $R/$OUT/prof2/*/*_kernel_trace.csv > $R/$OUT/s35_launches_in_flight.txt 2>&1
rm -rf $R/$OUT/prof2
for c in 3 5; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof$c -- python $R/bench.py --config $c --gpus 1 --sub --no-parity --steps 60 --warmup 5 --no-cpu-baseline > $R/$OUT/prof$c.log 2>&1
  cp $R/$OUT/prof$c/*/*_kernel_stats.csv $R/$OUT/s35_kernel_stats_cfg$c.csv 2>/dev/null
  rm -rf $R/$OUT/prof$c
done
timeout 300 python $R/bench.py --config 3 --frames 512 --gpus 1 --sub --steps 80 --warmup 5 --no-cpu-baseline > $R/$OUT/bench_cfg3_f512.json 2> $R/$OUT/bench_cfg3_f512.err
cd $R
python - <<'PY'
import json, glob, os
for f in ['gpurun_out/r05_s35/bench_default_k20.json', 'gpurun_out/r05_s35/bench_default_k200.json']:
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j['ms_per_step'], j['value'], j['roofline']['frac'], j['roofline_chain']['frac'], [(k['slot'], k['us']) for k in j['roofline']['kernels']], j['config'].get('tiles'))
        print('   traffic', j['roofline'].get('traffic'), j['roofline'].get('traffic_over_algorithmic'), json.dumps(j['roofline'].get('traffic_live'))[:300])
        for a in j.get('also', []):
            print('   also', a.get('variant'), a.get('baseline_config'), a.get('frames_per_batch'), a.get('ms_per_step'), a.get('value'), a.get('error'), (a.get('roofline') or {}).get('traffic'), (a.get('parity') or {}).get('timed_path'), (a.get('parity') or {}).get('failed'))
        if j.get('cpu_baseline'): print('   cpu', j['cpu_baseline']['value'], j['cpu_baseline']['one_thread'], j['cpu_baseline']['cores'])
        if 'parity' in j: print('   parity', json.dumps(j['parity'])[:500])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
for c in 3 5 3f512; do echo "== cfg $c"; python scripts/pmc_summary.py $OUT/s35_cfg${c}_pmc_fetch.csv $OUT/s35_cfg${c}_pmc_write.csv | grep -A2 "scan_step\|rows\|cols1024\|plan_long\|sub_dft"; done
tail -c 600 $OUT/bench_cfg3_f512.json | cut -c1-600
tail -3 $OUT/smoke.txt | cut -c1-300; head -8 $OUT/s35_launches_in_flight.txt
