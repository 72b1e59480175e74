#!/bin/bash
# round 6, session 26: THE evidence session of the final binaries (session 14 once more, on the tree as it is committed at the end of the round) — whole GPU suite, smoke, the driver's form of the default line (compact +
# full), 200-step and 20-step lines, rocprofv3 kernel statistics + launch overlap of the default command, kernel statistics and PMC passes
# (FETCH_SIZE / WRITE_SIZE, separate passes, --kernel-trace only) of configs 3 and 5 and of getFft's sizes 131072 / 262144, the calibration
# kernels (scripts/ubench/hbm_calib) in passes of their own, the soak (60 random detect-mode sessions, culled == unculled)
OUT=gpurun_out/r06_s26
mkdir -p $OUT
R=/root/repo
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default_k20.json 2> $OUT/bench_default_k20.err ) 2> $OUT/bench_default_k20.time
cp bench_full.json $OUT/bench_full_k20.json 2>/dev/null
tail -c 120 $OUT/bench_default_k20.time; wc -c $OUT/bench_default_k20.json
timeout 900 python bench.py --no-also > $OUT/bench_default_k200.json 2> $OUT/bench_default_k200.err
cp bench_full.json $OUT/bench_full_k200.json 2>/dev/null
for i in 1 2 3; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-parity --no-live-pmc > $OUT/k20_$i.json 2>/dev/null
  timeout 300 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-also --no-parity --no-live-pmc > $OUT/k200_$i.json 2>/dev/null
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof2 -- python $R/bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-also --no-parity --no-live-pmc > $R/$OUT/prof2.log 2>&1
cp $R/$OUT/prof2/*/*_kernel_stats.csv $R/$OUT/s26_kernel_stats.csv 2>/dev/null
python $R/scripts/launches_in_flight.py $R/$OUT/prof2/*/*_kernel_trace.csv > $R/$OUT/s26_launches_in_flight.txt 2>&1
rm -rf $R/$OUT/prof2
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/tl -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-parity --no-live-pmc --no-kernel-timing > /dev/null 2>&1
python $R/scripts/timeline_tail.py $(find $R/$OUT/tl -name "*kernel_trace.csv" | head -1) 30 > $R/$OUT/s26_timeline_k20.txt 2>&1
rm -rf $R/$OUT/tl
declare -A ARGS
ARGS[cfg3]="--config 3"
ARGS[cfg3f512]="--config 3 --frames 512"
ARGS[cfg5]="--config 5"
ARGS[n131072]="--config 3 --fft 131072 --frames 64"
ARGS[n262144]="--config 5 --fft 262144 --frames 32"
for c in cfg3 cfg3f512 cfg5 n131072 n262144; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_$c -- python $R/bench.py ${ARGS[$c]} --gpus 1 --sub --no-parity --steps 60 --warmup 5 --no-cpu-baseline --no-kernel-timing > $R/$OUT/prof_$c.log 2>&1
  cp $R/$OUT/prof_$c/*/*_kernel_stats.csv $R/$OUT/s26_kernel_stats_$c.csv 2>/dev/null
  rm -rf $R/$OUT/prof_$c
  for k in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --pmc $k --kernel-trace --output-format csv -d $R/$OUT/pmc_${k}_$c -- python $R/bench.py ${ARGS[$c]} --gpus 1 --sub --no-parity --steps 30 --warmup 5 --preheat-ms 0 --no-cpu-baseline --no-kernel-timing > $R/$OUT/pmc_${k}_$c.log 2>&1
  done
  cp $R/$OUT/pmc_FETCH_SIZE_$c/*/*_counter_collection.csv $R/$OUT/s26_${c}_pmc_fetch.csv 2>/dev/null
  cp $R/$OUT/pmc_WRITE_SIZE_$c/*/*_counter_collection.csv $R/$OUT/s26_${c}_pmc_write.csv 2>/dev/null
  rm -rf $R/$OUT/pmc_FETCH_SIZE_$c $R/$OUT/pmc_WRITE_SIZE_$c
done
for k in FETCH_SIZE WRITE_SIZE; do
  timeout 100 rocprofv3 --pmc $k --kernel-trace --output-format csv -d $R/$OUT/pmc_${k}_calib -- $R/scripts/ubench/hbm_calib > $R/$OUT/pmc_${k}_calib.log 2>&1
  cp $R/$OUT/pmc_${k}_calib/*/*_counter_collection.csv $R/$OUT/s26_calib_pmc_$k.csv 2>/dev/null
  rm -rf $R/$OUT/pmc_${k}_calib
done
cd $R
python - <<'PY'
import json, glob, os, csv
for f in ['gpurun_out/r06_s26/bench_default_k20.json', 'gpurun_out/r06_s26/bench_default_k200.json'] + sorted(glob.glob('gpurun_out/r06_s26/k2*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        r = j['roofline']
        print(os.path.basename(f), j['ms_per_step'], j['value'], r['frac'], r['kernel_us'], r.get('launches_in_flight'), r.get('traffic'), r.get('traffic_over_algorithmic'))
        for a in j.get('also', []): print('   also', {k: v for k, v in a.items() if k not in ('parity', 'ss_process_pieces_ms')}, (a.get('parity') or {}))
        if j.get('cpu_baseline'): print('   cpu', j['cpu_baseline']['value'], j['cpu_baseline']['one_thread'], j['cpu_baseline']['cores'])
        if j.get('parity'): print('   parity', json.dumps(j['parity'])[:700])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
def avg(path, match=None):
    rows = [r for r in csv.DictReader(open(path)) if (match is None or match in r['Kernel_Name'])]
    by = {}
    for r in rows: by.setdefault(r['Kernel_Name'][:60], []).append(float(r['Counter_Value']))
    return {k: (round(sum(v) / len(v), 1), len(v)) for k, v in by.items() if sum(v) / len(v) > 1000}
for c in ('cfg3', 'cfg3f512', 'cfg5', 'n131072', 'n262144', 'calib'):
    for kind in ('fetch', 'write') if c != 'calib' else ('FETCH_SIZE', 'WRITE_SIZE'):
        p = f'gpurun_out/r06_s26/s26_{c}_pmc_{kind}.csv'
        if os.path.exists(p): print(c, kind, 'KiB per launch (mean, launches):', avg(p))
PY
head -8 $OUT/s26_launches_in_flight.txt; tail -12 $OUT/s26_timeline_k20.txt
SS_FUZZ_CULL_SEEDS=60 timeout 1500 python -m pytest tests/test_gpu_cull.py -x -q -m gpu -k random > $OUT/soak.txt 2>&1; tail -2 $OUT/soak.txt
