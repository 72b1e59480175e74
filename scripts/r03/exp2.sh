#!/bin/bash
# round 3, experiment 2: where does the deep-pipelined step spend its time with tile culling on?
OUT=gpurun_out/r03_s2
mkdir -p $OUT
cd /root/repo
B="python bench.py --no-cpu-baseline --steps 200 --warmup 5 --diag-lib"
SS_CULL_STATS=1 timeout 300 $B > $OUT/bench_diag.json 2> $OUT/bench_diag.err
SS_CULL_STATS=1 timeout 300 $B --start-level 100 > $OUT/bench_diag_sl100.json 2> $OUT/bench_diag_sl100.err
SS_CULL_STATS=1 timeout 300 $B --start-level 100 --no-psd-out > $OUT/bench_diag_sl100_detect.json 2> $OUT/bench_diag_sl100_detect.err
SS_STEP_STAMPS=$OUT/stamps.txt timeout 300 $B > $OUT/bench_diag_stamps.json 2> $OUT/bench_diag_stamps.err
python scripts/analyze_step_stamps.py $OUT/stamps.txt > $OUT/stamps_summary.txt 2>&1
SS_STEP_STAMPS=$OUT/stamps_nocull.txt timeout 300 $B --no-cull > $OUT/bench_diag_stamps_nocull.json 2> $OUT/bench_diag_stamps_nocull.err
python scripts/analyze_step_stamps.py $OUT/stamps_nocull.txt > $OUT/stamps_nocull_summary.txt 2>&1
SS_DEEP=0 timeout 300 $B > $OUT/bench_diag_nodeep.json 2> $OUT/bench_diag_nodeep.err
SS_DEEP=0 timeout 300 $B --start-level 100 > $OUT/bench_diag_nodeep_sl100.json 2> $OUT/bench_diag_nodeep_sl100.err
for o in "E*|D128,F1024" "E*|F*,D*" "F*|D*,E*" "E*|D*,F*" "E*|D64,F256"; do
  tag=$(echo "$o" | tr -d '*|,' )
  SS_STEP_ORDER="$o" timeout 300 $B > $OUT/bench_order_$tag.json 2> $OUT/bench_order_$tag.err
done
LAB_FFT_ONLY=1 timeout 200 scripts/ubench/launch_overlap_lab > $OUT/lab.txt 2>&1
LAB_FFT_ONLY=1 timeout 200 scripts/ubench/launch_overlap_lab_iqnt > $OUT/lab_iqnt.txt 2>&1
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r03_s2/bench_*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j['ms_per_step'], j['value'], j['roofline_chain']['frac'], j['roofline']['kernel_us'], j['config']['candidates_per_batch'])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
grep -h "specscan diag" $OUT/*.err
cat $OUT/stamps_summary.txt $OUT/stamps_nocull_summary.txt $OUT/lab.txt $OUT/lab_iqnt.txt
rocprofv3 --list-avail 2>/dev/null > $OUT/counters_all.txt
grep -i -E "mall|_EA_|EA0_|DRAM|HBM" $OUT/counters_all.txt | head -60
