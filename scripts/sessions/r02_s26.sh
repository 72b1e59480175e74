#!/bin/bash
# round 2, GPU session 26: dispatch orders and FFT workgroup shapes once consecutive launches overlap (deep pipelining)
set -x
OUT=gpurun_out/r02_s26; mkdir -p $OUT; rm -f $OUT/summary.txt
run() { local name=$1; shift
  env "$@" timeout 300 python bench.py --diag-lib --no-cpu-baseline 2> $OUT/$name.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['ms_per_step'], d['value'], d['roofline']['kernel_us'], d['roofline']['launches_in_flight'])" | tee -a $OUT/summary.txt
}
for rep in 1 2; do
run default SS_X=0
run F_then_D 'SS_STEP_ORDER=E*|F*,D*'
run D_then_F 'SS_STEP_ORDER=E*|D*,F*'
run D_F_E 'SS_STEP_ORDER=|D*,F*,E*'
run F_D_E 'SS_STEP_ORDER=|F*,D*,E*'
run D256F256 'SS_STEP_ORDER=E*|D256,F256'
run D64F64 'SS_STEP_ORDER=E*|D64,F64'
run D512_F 'SS_STEP_ORDER=E*|D512,F1100'
run fft_per_wg2 SS_FFT_PER_WG=2
run fft_2slots SS_FFT_PER_WG=-1
run in_order SS_DEEP=0
done
