#!/bin/bash
# round 4, session 11: 65536 points with tile culling and no dB plane in detect mode (what 2^20 got in sessions 4-8) against the form
# that shipped so far (SS_CULL_65536=0) and against culling with a dB plane AND ring rows (SS_RING_ONLY=0); whole GPU suite first
OUT=gpurun_out/r04_s11
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1
echo "tests rc=$?" >> $OUT/rc.txt
B="python bench.py --no-cpu-baseline --no-also --no-parity --warmup 5 --diag-lib --config 3 --gpus 1"
for rep in 1 2; do
  for fr in 128 16; do
    st=100; [ $fr = 16 ] && st=400
    timeout 300 $B --steps $st --frames $fr > $OUT/c3_f${fr}_new_$rep.json 2>> $OUT/ab.err
    SS_CULL_65536=0 timeout 300 $B --steps $st --frames $fr > $OUT/c3_f${fr}_nocull_$rep.json 2>> $OUT/ab.err
    SS_RING_ONLY=0 timeout 300 $B --steps $st --frames $fr > $OUT/c3_f${fr}_twoplanes_$rep.json 2>> $OUT/ab.err
  done
done
timeout 300 python bench.py --no-cpu-baseline --no-also --warmup 5 --config 3 --gpus 1 --steps 100 --sub > $OUT/c3_prod_parity.json 2>> $OUT/ab.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r04_s11/c3*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j['ms_per_step'], j['value'], [(k['slot'], k['us']) for k in j['roofline']['kernels']], j['config'].get('tiles'), json.dumps(j.get('parity'))[:400] if j.get('parity') else '')
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
cat $OUT/rc.txt; tail -8 $OUT/pytest_gpu.txt | cut -c1-400; tail -5 $OUT/ab.err | cut -c1-300
