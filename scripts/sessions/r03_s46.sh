#!/bin/bash
# round 3, GPU session 46: the whole GPU suite on the diagnostics build with the alternatives of DESIGN.md 6f, extended fuzz
OUT=gpurun_out/r03_s46; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for alt in "SS_DEEP=0" "SS_PIPELINE=0" "SS_CULL=0" "SS_STEP_LONG=0" "SS_QUEUES=3"; do
  env SS_TEST_USE_DIAG_LIB=1 $alt timeout 600 python -m pytest tests -q -m gpu -x > "$OUT/pytest_${alt//=/_}.txt" 2>&1; echo "$alt: $(tail -1 "$OUT/pytest_${alt//=/_}.txt")"
done
SS_FUZZ_SEEDS=80 SS_FUZZ_SEEDS2=50 SS_FUZZ_SEEDS3=50 SS_FUZZ_SEEDS4=20 SS_FUZZ_CULL_SEEDS=80 timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_cull.py -q -m gpu > $OUT/fuzz_extended.txt 2>&1; echo "fuzz: $(tail -1 $OUT/fuzz_extended.txt)"
