#!/bin/bash
OUT=gpurun_out/r03_s13
mkdir -p $OUT
cd /root/repo
B="python bench.py --no-cpu-baseline --warmup 5"
for v in base iqdef psdnt psdsc1 psdsc0sc1 psdntsc1 detnt; do
  timeout 300 $B --steps 200 --lib scripts/ab/libspecscan_$v.so > $OUT/bench_${v}_200.json 2> $OUT/bench_${v}_200.err
  timeout 300 $B --steps 20 --lib scripts/ab/libspecscan_$v.so > $OUT/bench_${v}_20.json 2> $OUT/bench_${v}_20.err
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r03_s13/bench_*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j['ms_per_step'], j['value'], j['roofline_chain']['frac'], j['roofline']['kernel_us'], j['config']['candidates_per_batch'])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /root/repo/$OUT/trace20 -- python /root/repo/bench.py --no-cpu-baseline --steps 20 --warmup 5 --no-kernel-timing > /root/repo/$OUT/trace20.log 2>&1
find /root/repo/$OUT/trace20 -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} /root/repo/$OUT/trace20_kernel_trace.csv
rm -rf /root/repo/$OUT/trace20
ls -la /root/repo/$OUT | head -40
