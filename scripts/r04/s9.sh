#!/bin/bash
# round 4, session 9: more than two launch queues for the 8192-point step (diagnostics build, SS_QUEUES), the multi-rank path on one
# GPU (two ranks over gloo sharing the device: bands and frame ranges), config 1 on the box's host cores, the new API test
OUT=gpurun_out/r04_s9
mkdir -p $OUT
R=/root/repo
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_stream_ordered.py -m gpu -q --timeout 600 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "tests rc=$?" >> $OUT/rc.txt
B="python bench.py --no-cpu-baseline --no-also --no-parity --warmup 5 --diag-lib"
for i in 1 2; do
  for q in 2 3 4; do
    SS_QUEUES=$q timeout 300 $B --steps 200 > $OUT/k200_q${q}_$i.json 2>> $OUT/ab.err
    SS_QUEUES=$q timeout 300 $B --steps 20 > $OUT/k20_q${q}_$i.json 2>> $OUT/ab.err
  done
done
timeout 600 python bench.py --gpus 2 --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_gpus2_bands.json 2> $OUT/bench_gpus2_bands.err
timeout 600 python bench.py --config 5 --gpus 2 --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_gpus2_frames.json 2> $OUT/bench_gpus2_frames.err
timeout 300 python bench.py --config 1 --cpu-seconds 8 > $OUT/bench_cfg1.json 2> $OUT/bench_cfg1.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r04_s9/k*.json')) + sorted(glob.glob('gpurun_out/r04_s9/bench_*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j['ms_per_step'], j['value'], j.get('n_gpus'), (j.get('roofline_chain') or {}).get('frac'), j['config'].get('dist_backend'), j['config'].get('file_fed', {}).get('value') if isinstance(j['config'].get('file_fed'), dict) else '')
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
cat $OUT/rc.txt; tail -3 $OUT/pytest.log | cut -c1-300; tail -3 $OUT/bench_gpus2_frames.err | cut -c1-300
