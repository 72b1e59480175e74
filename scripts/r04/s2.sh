#!/bin/bash
# round 4, session 2: the new test files one process each (a crash in one must not hide the others), then an A/B of the
# 8192-point step against round 3's library in one session (alternating runs)
OUT=gpurun_out/r04_s2
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0
export PYTHONFAULTHANDLER=1
for f in test_gpu_stream_ordered test_gpu_degenerate_input test_gpu_stated_configs; do
  timeout 1200 python -X faulthandler -m pytest tests/$f.py -m gpu -q -s --timeout 900 -p no:cacheprovider > $OUT/$f.log 2>&1
  echo "$f rc=$?" >> $OUT/rc.txt
done
B="python bench.py --no-cpu-baseline --no-also --no-parity --warmup 5"
for i in 1 2 3; do
  timeout 300 $B --steps 200 --diag-lib > $OUT/k200_new$i.json 2>> $OUT/ab.err
  timeout 300 $B --steps 200 --lib scripts/ab/libspecscan_r03.so > $OUT/k200_r03_$i.json 2>> $OUT/ab.err
  timeout 300 $B --steps 20 --diag-lib > $OUT/k20_new$i.json 2>> $OUT/ab.err
  timeout 300 $B --steps 20 --lib scripts/ab/libspecscan_r03.so > $OUT/k20_r03_$i.json 2>> $OUT/ab.err
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r04_s2/k*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j['ms_per_step'], j['value'], j['roofline_chain']['frac'], j['roofline']['kernel_us'])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
cat $OUT/rc.txt
for f in test_gpu_stream_ordered test_gpu_degenerate_input test_gpu_stated_configs; do echo "== $f"; grep -v "^  File\|^$" $OUT/$f.log | tail -25 | cut -c1-400; done
