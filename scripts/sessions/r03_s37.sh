#!/bin/bash
# round 3, GPU session 37: suite on the tree, timeline of a 20-step run (fill and drain), PMC traffic of configs 3 and 5
OUT=gpurun_out/r03_s37; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
cd /tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/trace_k20 -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --preheat-ms 100 > $R/$OUT/trace_k20.log 2>&1
for c in 3 5; do
  for k in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --pmc $k --kernel-trace --output-format csv -d $R/$OUT/pmc_${k}_cfg$c -- python $R/bench.py --config $c --gpus 1 --steps 30 --warmup 5 --preheat-ms 0 --no-cpu-baseline --sub > $R/$OUT/pmc_${k}_cfg$c.log 2>&1
  done
  timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$OUT/pmc_FETCH_SIZE_cfg${c}_nocull -- python $R/bench.py --config $c --gpus 1 --steps 30 --warmup 5 --preheat-ms 0 --no-cpu-baseline --sub --no-cull > $R/$OUT/pmc_nocull_cfg$c.log 2>&1
done
cd $R
cp $OUT/trace_k20/*/*_kernel_trace.csv $OUT/trace_k20.csv
python scripts/timeline_tail.py $OUT/trace_k20.csv 32 | tee $OUT/timeline_k20.txt
for d in $OUT/pmc_*; do [ -d $d ] && cp $d/*/*_counter_collection.csv $d.csv; done
rm -rf $OUT/trace_k20 $OUT/pmc_*_cfg3 $OUT/pmc_*_cfg5 $OUT/pmc_*_nocull
python scripts/pmc_summary.py $OUT/pmc_*.csv 2>&1 | tail -60
