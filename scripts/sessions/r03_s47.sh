#!/bin/bash
# round 3, GPU session 47 (final tree, after the ring rows moved to the drain): PMC traffic of configs 2 / 3 / 5 (separate FETCH_SIZE and WRITE_SIZE passes; bench.py reads the
# copies under profiles/r03/), the whole GPU suite, smoke, the benchmark line with its `also` entries and the CPU baseline, the
# 20-step form, no-cull and detect mode, rocprofv3 kernel statistics + trace of the same commands, config 1 on the box's host cores
OUT=gpurun_out/r03_s47; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for c in 2 3 5; do
  EXTRA="--config $c --gpus 1 --sub"; [ $c = 2 ] && EXTRA="--no-also"
  for k in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --pmc $k --kernel-trace --output-format csv -d $R/$OUT/pmc_${k}_cfg$c -- python $R/bench.py $EXTRA --steps 30 --warmup 5 --preheat-ms 0 --no-cpu-baseline > $R/$OUT/pmc_${k}_cfg$c.log 2>&1
  done
  cp $R/$OUT/pmc_FETCH_SIZE_cfg$c/*/*_counter_collection.csv $R/$OUT/s47_cfg${c}_pmc_fetch.csv
  cp $R/$OUT/pmc_WRITE_SIZE_cfg$c/*/*_counter_collection.csv $R/$OUT/s47_cfg${c}_pmc_write.csv
  cp $R/$OUT/s47_cfg${c}_pmc_*.csv $R/profiles/r03/
  rm -rf $R/$OUT/pmc_FETCH_SIZE_cfg$c $R/$OUT/pmc_WRITE_SIZE_cfg$c
done
cd $R
timeout 900 python -m pytest tests -q -m gpu -s > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
grep -h "^\[" $OUT/pytest_gpu.txt > $OUT/stated_configs_parity.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; tail -4 $OUT/smoke.txt
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cut -c1-200 $OUT/bench_default.json
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_k20.json 2> $OUT/bench_k20.err; cut -c1-200 $OUT/bench_k20.json
timeout 120 python bench.py --no-cpu-baseline --no-also --no-cull > $OUT/bench_nocull.json 2> $OUT/bench_nocull.err
timeout 120 python bench.py --no-cpu-baseline --no-also --no-psd-out > $OUT/bench_detect.json 2> $OUT/bench_detect.err
timeout 120 python bench.py --no-cpu-baseline --config 3 --gpus 1 > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err
timeout 120 python bench.py --no-cpu-baseline --config 3 --gpus 1 --no-cull > $OUT/bench_cfg3_nocull.json 2> $OUT/bench_cfg3.err
timeout 120 python bench.py --no-cpu-baseline --config 5 --gpus 1 > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err
timeout 120 python bench.py --no-cpu-baseline --config 5 --gpus 1 --no-cull > $OUT/bench_cfg5_nocull.json 2> $OUT/bench_cfg5.err
timeout 120 python bench.py --no-cpu-baseline --config 5 --gpus 1 --frames 64 --steps 50 > $OUT/bench_cfg5_f64.json 2> $OUT/bench_cfg5.err
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -- python $R/bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-also > $R/$OUT/prof.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof3 -- python $R/bench.py --config 3 --gpus 1 --sub --steps 200 --warmup 10 --no-cpu-baseline > $R/$OUT/prof3.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof5 -- python $R/bench.py --config 5 --gpus 1 --sub --steps 100 --warmup 10 --no-cpu-baseline > $R/$OUT/prof5.log 2>&1
cd $R
python scripts/launches_in_flight.py $OUT/prof/*/*_kernel_trace.csv | tee $OUT/launches_in_flight.txt
cp $OUT/prof/*/*_kernel_stats.csv $OUT/kernel_stats.csv; cp $OUT/prof3/*/*_kernel_stats.csv $OUT/kernel_stats_cfg3.csv; cp $OUT/prof5/*/*_kernel_stats.csv $OUT/kernel_stats_cfg5.csv
rm -rf $OUT/prof $OUT/prof3 $OUT/prof5
head -3 $OUT/kernel_stats.csv | cut -c1-160; head -6 $OUT/kernel_stats_cfg3.csv | cut -c1-160; head -7 $OUT/kernel_stats_cfg5.csv | cut -c1-160
timeout 200 python bench.py --config 1 --cpu-seconds 10 > $OUT/bench_cfg1.json 2> $OUT/bench_cfg1.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r03_s47/bench_*.json')):
    try:
        j = json.loads([l for l in open(f).read().splitlines() if l.startswith('{')][-1])
        r = j.get('roofline') or {}
        ks = {k['slot']: (k['us'], k['pmc_bytes_per_launch_from_profiles']) for k in r.get('kernels', [])}
        print(os.path.basename(f), j['ms_per_step'], j['value'], (j.get('roofline_chain') or {}).get('frac'), (j.get('roofline_chain') or {}).get('pmc_bytes_per_sample_from_profiles'), ks)
        for a in j.get('also', []):
            print('   also', a.get('baseline_config'), a.get('ms_per_step'), a.get('value'), a.get('error'), a.get('roofline_chain', {}).get('pmc_bytes_per_sample_from_profiles'))
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
# the launcher paths on this one-GPU box (two ranks share the device over gloo: functional only)
timeout 300 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_gpus2_bands.json 2> $OUT/bench_gpus2_bands.err; cut -c1-160 $OUT/bench_gpus2_bands.json
timeout 300 python bench.py --config 5 --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_gpus2_frames.json 2> $OUT/bench_gpus2_frames.err; cut -c1-160 $OUT/bench_gpus2_frames.json; tail -2 $OUT/bench_gpus2_frames.err
