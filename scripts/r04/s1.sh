#!/bin/bash
# round 4, session 1: the whole GPU suite on the new tree (new tests first), the default bench line with its `also` entries,
# the wake-up latency A/B (HSA_ENABLE_INTERRUPT), the L2-scratch lab, a kernel trace of the default line
OUT=gpurun_out/r04_s1
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0
NEW="tests/test_gpu_wait_bound.py tests/test_gpu_stream_ordered.py tests/test_gpu_degenerate_input.py tests/test_gpu_stated_configs.py"
timeout 1200 python -m pytest $NEW -m gpu -q -s --timeout 900 -p no:cacheprovider > $OUT/pytest_new.log 2>&1
echo "new tests rc=$?" > $OUT/rc.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider --deselect tests/test_gpu_wait_bound.py --deselect tests/test_gpu_stream_ordered.py --deselect tests/test_gpu_degenerate_input.py --deselect tests/test_gpu_stated_configs.py > $OUT/pytest_rest.log 2>&1
echo "rest rc=$?" >> $OUT/rc.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_default_k20.json 2> $OUT/bench_default_k20.err
echo "bench rc=$?" >> $OUT/rc.txt
B="python bench.py --no-cpu-baseline --no-also --warmup 5"
for i in 1 2; do
  timeout 300 $B --steps 20 > $OUT/k20_a$i.json 2>> $OUT/ab.err
  HSA_ENABLE_INTERRUPT=0 timeout 300 $B --steps 20 > $OUT/k20_noint$i.json 2>> $OUT/ab.err
done
timeout 300 $B --steps 200 > $OUT/k200.json 2>> $OUT/ab.err
HSA_ENABLE_INTERRUPT=0 timeout 300 $B --steps 200 > $OUT/k200_noint.json 2>> $OUT/ab.err
timeout 120 scripts/ubench/l2_scratch_lab 128 200 > $OUT/l2_scratch_lab.txt 2>&1
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/$OUT/prof -o trace -- python /root/repo/bench.py --no-cpu-baseline --no-also --warmup 5 --steps 200 > /root/repo/$OUT/prof_bench.json 2> /root/repo/$OUT/prof.err)
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r04_s1/k*.json')) + ['gpurun_out/r04_s1/bench_default_k20.json']:
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j['ms_per_step'], j['value'], j['roofline_chain']['frac'], j['roofline']['kernel_us'], j['config'].get('tail_us'), j['config'].get('tiles'))
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
tail -5 $OUT/pytest_new.log; tail -3 $OUT/pytest_rest.log; cat $OUT/l2_scratch_lab.txt
