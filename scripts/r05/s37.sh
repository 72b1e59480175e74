#!/bin/bash
# round 5, session 37: the plan's copy of the per-column maxima eight loads at a time (its workgroups publish their lists before the first
# frame workgroups are through), with the halo frames' maxima of session 36 — the whole GPU suite, the default line of the PRODUCT library
# (20 and 200 steps), rocprofv3 kernel statistics + launch overlap, then diagnostics-build A/Bs (SS_HALO_MAXIMA=0) and stamps
OUT=gpurun_out/r05_s37
mkdir -p $OUT
R=/root/repo
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_cull.py tests/test_gpu_stated_configs.py -x -q -m gpu -k "culled_lists_equal or deep_pipelined or threshold or ignored or short_calls or config2" > $OUT/pytest_first.txt 2>&1 || { tail -30 $OUT/pytest_first.txt; exit 1; }
tail -2 $OUT/pytest_first.txt
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1
tail -3 $OUT/pytest_gpu.txt | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-also > $OUT/bench_default_k20.json 2> $OUT/bench_default_k20.err
timeout 600 python bench.py --no-also > $OUT/bench_default_k200.json 2> $OUT/bench_default_k200.err
python - <<'PY'
import json, os
for f in ['gpurun_out/r05_s37/bench_default_k20.json', 'gpurun_out/r05_s37/bench_default_k200.json']:
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j['ms_per_step'], j['value'], j['roofline']['frac'], [(k['slot'], k['us']) for k in j['roofline']['kernels']], j['config'].get('tiles'))
        print('   traffic', j['roofline'].get('traffic'), j['roofline'].get('traffic_over_algorithmic'), 'parity failed', (j.get('parity') or {}).get('failed'), 'cpu', (j.get('cpu_baseline') or {}).get('value'))
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof2 -- python $R/bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-also --no-parity --no-live-pmc > $R/$OUT/prof2.log 2>&1
cp $R/$OUT/prof2/*/*_kernel_stats.csv $R/$OUT/s37_kernel_stats.csv 2>/dev/null
python $R/scripts/launches_in_flight.py $R/$OUT/prof2/*/*_kernel_trace.csv > $R/$OUT/s37_launches_in_flight.txt 2>&1
rm -rf $R/$OUT/prof2
head -4 $R/$OUT/s37_launches_in_flight.txt
cd $R
run() {  # tag steps env...
  tag=$1; k=$2; shift 2
  env "$@" timeout 300 python bench.py --steps $k --warmup 5 --no-cpu-baseline --no-also --no-parity --no-live-pmc --diag-lib > $OUT/${tag}_k$k.json 2> $OUT/${tag}_k$k.err
  python - <<PY
import json
try:
    j = json.loads(open('$OUT/${tag}_k$k.json').read().strip().splitlines()[-1])
    print('$tag k=$k', j['ms_per_step'], j['value'], j['roofline']['frac'], j['config'].get('tiles'))
except Exception as e:
    print('$tag k=$k ERR', e, open('$OUT/${tag}_k$k.err').read()[-400:])
PY
}
for rep in 1 2; do
  run new$rep 200 SS_X=0
  run old$rep 200 SS_HALO_MAXIMA=0
done
SS_STEP_STAMPS=$OUT/stamps_default.txt timeout 300 python bench.py --gpus 1 --no-parity --steps 100 --warmup 5 --no-cpu-baseline --no-also --diag-lib > $OUT/st_default.json 2> $OUT/st_default.err
python scripts/analyze_step_stamps.py $OUT/stamps_default.txt 32 2>&1 | tee $OUT/stamps_default_summary.txt | head -12
