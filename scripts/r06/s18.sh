#!/bin/bash
# round 6, session 18: where the 20-step form's fixed 38 us go (GPU span 458 us, host-measured 496): the host thread's wait policy
# (hipSetDeviceFlags: auto / spin / yield) in alternating runs, 20 and 200 steps; then the 20-step timeline with every kernel of the library
OUT=gpurun_out/r06_s18
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
for i in 1 2 3 4; do
  for w in auto spin yield; do
    timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-parity --no-live-pmc --host-wait $w > $OUT/k20_${w}_$i.json 2>/dev/null
    cp bench_full.json $OUT/full_k20_${w}_$i.json
    [ $i -le 2 ] && { timeout 300 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-also --no-parity --no-live-pmc --host-wait $w > $OUT/k200_${w}_$i.json 2>/dev/null; cp bench_full.json $OUT/full_k200_${w}_$i.json; }
  done
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-parity --no-live-pmc --host-wait spin --sync-engine-first > $OUT/k20_spineng_$i.json 2>/dev/null
  cp bench_full.json $OUT/full_k20_spineng_$i.json
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_s18/full_*.json')):
    j = json.load(open(f))
    print(f.split('/')[-1], j['ms_per_step'], j['value'], j['roofline']['frac'], j['config']['tail_us'], j['config']['host_enqueue_ms_per_step'])
PY
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python /root/repo/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-parity --no-live-pmc --no-kernel-timing > /dev/null 2>&1
cd /root/repo
python scripts/timeline_tail.py $(find /tmp/tl -name "*kernel_trace.csv" | head -1) 12 all > $OUT/timeline_k20_tail_all.txt; cat $OUT/timeline_k20_tail_all.txt
