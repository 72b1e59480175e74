#!/bin/bash
# round 2, GPU session 27: the numbers DESIGN.md / README / BASELINE.md quote for the final state (deep pipelining), rocprofv3
# kernel statistics and PMC traffic of the same command
set -x
OUT=gpurun_out/r02_s27; mkdir -p $OUT
export TMPDIR=/tmp
B="python bench.py --steps 300 --warmup 20 --no-cpu-baseline"
timeout 300 python bench.py --steps 300 --warmup 20 > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 300 $OUT/bench_default.json
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_k20.json 2> $OUT/bench_k20.err
timeout 200 $B --sync-every-step > $OUT/bench_sync_every_step.json 2>&1
SS_DEEP=0 timeout 200 $B --diag-lib > $OUT/bench_diag_roles_in_order.json 2>&1
timeout 200 $B --diag-lib > $OUT/bench_diag_default.json 2>&1
SS_PIPELINE=0 timeout 200 $B --diag-lib > $OUT/bench_diag_no_stage_pipelining.json 2>&1
timeout 200 $B --no-psd-out > $OUT/bench_detect_mode.json 2>&1
timeout 200 $B --planes > $OUT/bench_planes.json 2>&1
timeout 200 $B --spectrogram > $OUT/bench_spectrogram.json 2>&1
timeout 200 $B --decim 5 > $OUT/bench_decim5.json 2>&1
timeout 200 $B --fmt cs8 > $OUT/bench_cs8.json 2>&1
timeout 200 $B --frames 256 > $OUT/bench_frames256.json 2>&1
timeout 200 $B --frames 512 > $OUT/bench_frames512.json 2>&1
timeout 200 $B --frames 2048 > $OUT/bench_frames2048.json 2>&1
timeout 200 $B --frames 4096 > $OUT/bench_frames4096.json 2>&1
timeout 200 $B --sets 7 > $OUT/bench_seven_sets.json 2>&1
timeout 300 python bench.py --config 3 --steps 200 --warmup 10 --no-cpu-baseline > $OUT/bench_cfg3.json 2>&1
timeout 300 python bench.py --config 5 --gpus 1 --steps 100 --warmup 5 --no-cpu-baseline > $OUT/bench_cfg5.json 2>&1
timeout 300 python bench.py --gpus 2 --steps 100 --warmup 10 > $OUT/bench_gpus2_gloo.json 2> $OUT/bench_gpus2.err
for f in $OUT/bench_*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d['roofline'] or {}
    print("%-46s %9.1f MS/s %.4f ms/step kernel %s us in flight %s frac %s chain %.4f" % (sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], r.get('kernel_us'), r.get('launches_in_flight'), r.get('frac'), (d.get('roofline_chain') or {}).get('frac', float('nan'))))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done | tee $OUT/summary.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_fetch -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --preheat-ms 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_write -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --preheat-ms 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/pmc_write.log 2>&1
cd $GRAFT_REPO_ROOT; find $OUT -name "*kernel_stats.csv" | xargs -n1 head -6; ls $OUT/prof/*/ $OUT/pmc_fetch/*/ | head
