#!/bin/bash
# round 6, session 15: the whole GPU suite on the DIAGNOSTICS build with no switch set (it must behave like the product), a 150-session soak
# of the culling paths on the product library, and `bench.py --gpus 2` — two rank processes on this box's one device over gloo — as a line
OUT=gpurun_out/r06_s15
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
SS_TEST_USE_DIAG_LIB=1 timeout 1800 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu_diag.txt 2>&1; tail -3 $OUT/pytest_gpu_diag.txt
SS_FUZZ_CULL_SEEDS=150 timeout 1800 python -m pytest tests/test_gpu_cull.py -x -q -m gpu -k random > $OUT/soak150.txt 2>&1; tail -2 $OUT/soak150.txt
SS_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 100 --warmup 10 > $OUT/bench_gpus2_bands_one_gpu.json 2> $OUT/bench_gpus2.err; tail -1 $OUT/bench_gpus2_bands_one_gpu.json | cut -c1-1500
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-parity --no-live-pmc | tail -1 | cut -c1-400
