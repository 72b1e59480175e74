#!/bin/bash
# round 4, session 35: one launch per call at 65536 points — dispatch orders of its row and column tiles
OUT=gpurun_out/r04_s35
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp SS_MERGE_65536=1
B="python bench.py --no-cpu-baseline --no-also --no-parity --warmup 5 --gpus 1 --config 3 --diag-lib"
i=0
for o in "E*|D*,R1,F1" "E*|D*,R*,F*" "E*|D*,F*,R*" "E*|D*,R2,F2" "E*|D*,R8,F8" "E*|D*,R32,F32" "D*|E*,R1,F1" "|R1,F1,D1,E1"; do
  i=$((i+1))
  SS_STEP_ORDER_MERGED="$o" timeout 300 $B --steps 100 > $OUT/c3_f128_o$i.json 2>> $OUT/ab.err
  SS_STEP_ORDER_MERGED="$o" timeout 300 $B --steps 200 --frames 64 > $OUT/c3_f64_o$i.json 2>> $OUT/ab.err
  echo "o$i = $o" >> $OUT/orders.txt
done
SS_MERGE_65536=0 timeout 300 $B --steps 100 > $OUT/c3_f128_ships.json 2>> $OUT/ab.err
python - <<'PY'
import glob, json, os
for f in sorted(glob.glob('gpurun_out/r04_s35/c3_*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j['ms_per_step'], j['value'], [(k['slot'], k['us']) for k in j['roofline']['kernels']])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
cat $OUT/orders.txt; tail -3 $OUT/ab.err | cut -c1-300
