#!/bin/bash
# round 2, GPU session 5: where does the run-to-run spread of k_scan_step come from? orders in one process on the same buffers
set -x
OUT=gpurun_out/r02_s5; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python scripts/step_order_lab.py --rounds 3 "E*|D128,F128" "E*|D256,F256" "F*|D*,E*" "E*,D*|F*" "E*|D64,F64" "E*|D128,F256" "E*|D128,F512" > $OUT/lab_same_buffers.txt 2>&1; cat $OUT/lab_same_buffers.txt
timeout 600 python scripts/step_order_lab.py --rounds 6 --realloc "E*|D128,F128" "E*|D256,F256" > $OUT/lab_realloc.txt 2>&1; cat $OUT/lab_realloc.txt
timeout 300 python scripts/step_order_lab.py --rounds 2 --env SS_STEP_ORDER_INLINE=1 "E*|D128,F128" "F*|D*,E*" > $OUT/lab_inline.txt 2>&1; cat $OUT/lab_inline.txt
