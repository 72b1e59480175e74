#!/bin/bash
# round 2, GPU session 32: final tree — whole GPU suite, 36 seeds of the long-run test, smoke, the benchmark line (200 and 20 steps),
# rocprofv3 kernel statistics + trace and PMC traffic of the same command
set -x
OUT=gpurun_out/r02_s32; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
SS_TEST_DEEP_SEEDS=36 timeout 900 python -m pytest tests/test_gpu_step_pipeline.py -q -m gpu -k long_runs > $OUT/deep_seeds.txt 2>&1; tail -2 $OUT/deep_seeds.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
timeout 300 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cut -c1-260 $OUT/bench_default.json
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_k20.json 2> $OUT/bench_k20.err; cut -c1-260 $OUT/bench_k20.json
timeout 300 python bench.py --spectrogram --no-cpu-baseline > $OUT/bench_spectrogram.json 2> $OUT/bench_spectrogram.err; cut -c1-260 $OUT/bench_spectrogram.json
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_fetch -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --preheat-ms 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_write -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --preheat-ms 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/pmc_write.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/launches_in_flight.py $OUT/prof/*/*_kernel_trace.csv | tee $OUT/launches_in_flight.txt
head -3 $OUT/prof/*/*_kernel_stats.csv | cut -c1-200
