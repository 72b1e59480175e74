#!/bin/bash
# round 5, session 4: the fold with two residues per workgroup as the product's form (KIND 8, 128 registers) — the whole GPU suite,
# config 3 at 128 / 256 / 512-frame calls with parity samples at the timed call size, kernel statistics and fabric bytes of config 3
OUT=gpurun_out/r05_s4
mkdir -p $OUT
R=/root/repo
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1
tail -15 $OUT/pytest_gpu.txt | cut -c1-300
for f in 128 256 512; do
  timeout 600 python bench.py --config 3 --frames $f --gpus 1 --sub --steps 100 --warmup 5 --no-cpu-baseline > $OUT/bench_cfg3_f$f.json 2> $OUT/bench_cfg3_f$f.err
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof3 -- python $R/bench.py --config 3 --gpus 1 --sub --no-parity --steps 60 --warmup 5 --no-cpu-baseline > $R/$OUT/prof3.log 2>&1
cp $R/$OUT/prof3/*/*_kernel_stats.csv $R/$OUT/s4_kernel_stats_cfg3.csv 2>/dev/null
rm -rf $R/$OUT/prof3
for k in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $k --kernel-trace --output-format csv -d $R/$OUT/pmc_${k}_cfg3 -- python $R/bench.py --config 3 --gpus 1 --sub --no-parity --steps 30 --warmup 5 --preheat-ms 0 --no-cpu-baseline > $R/$OUT/pmc_${k}_cfg3.log 2>&1
done
cp $R/$OUT/pmc_FETCH_SIZE_cfg3/*/*_counter_collection.csv $R/$OUT/s4_cfg3_pmc_fetch.csv
cp $R/$OUT/pmc_WRITE_SIZE_cfg3/*/*_counter_collection.csv $R/$OUT/s4_cfg3_pmc_write.csv
rm -rf $R/$OUT/pmc_FETCH_SIZE_cfg3 $R/$OUT/pmc_WRITE_SIZE_cfg3
cd $R
python - <<'PY'
import json
for f in ['bench_cfg3_f128.json', 'bench_cfg3_f256.json', 'bench_cfg3_f512.json']:
    try:
        j = json.loads(open('gpurun_out/r05_s4/' + f).read().strip().splitlines()[-1])
        print(f, j['ms_per_step'], j['value'], j['config']['tiles'], [(k['slot'], k['us'], k['launches_timed'], k['frames_per_launch'], k['frac_of_peak']) for k in j['roofline']['kernels']], j['roofline']['frac'], j['roofline_chain']['frac'])
        print('   parity', (j.get('parity') or {}).get('timed_path'), (j.get('parity') or {}).get('failed'))
    except Exception as e:
        print(f, 'ERR', e); print(open('gpurun_out/r05_s4/' + f.replace('.json', '.err')).read()[-1500:])
PY
python scripts/pmc_summary.py $OUT/s4_cfg3_pmc_fetch.csv $OUT/s4_cfg3_pmc_write.csv | head -20
head -6 $OUT/s4_kernel_stats_cfg3.csv | cut -c1-200
