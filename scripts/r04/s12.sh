#!/bin/bash
# round 4, session 12: the side lane — plan and deferred stages of a long transform's call on a queue of their own beside the next
# call's column and row launches — against the passengers' form (SS_SIDE_LANE=0): whole GPU suite, then configs 3 and 5
OUT=gpurun_out/r04_s12
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1
echo "tests rc=$?" >> $OUT/rc.txt
B="python bench.py --no-cpu-baseline --no-also --no-parity --warmup 5 --diag-lib --gpus 1"
for rep in 1 2; do
  for fr in 128 16; do
    st=100; [ $fr = 16 ] && st=400
    timeout 300 $B --config 3 --steps $st --frames $fr > $OUT/c3_f${fr}_side_$rep.json 2>> $OUT/ab.err
    SS_SIDE_LANE=0 timeout 300 $B --config 3 --steps $st --frames $fr > $OUT/c3_f${fr}_pass_$rep.json 2>> $OUT/ab.err
    SS_SIDE_LANE=0 SS_CULL_65536=0 timeout 300 $B --config 3 --steps $st --frames $fr > $OUT/c3_f${fr}_r03_$rep.json 2>> $OUT/ab.err
    SS_CULL_65536=0 timeout 300 $B --config 3 --steps $st --frames $fr > $OUT/c3_f${fr}_side_nocull_$rep.json 2>> $OUT/ab.err
  done
  for fr in 16 64; do
    st=100; [ $fr = 64 ] && st=40
    timeout 300 $B --config 5 --steps $st --frames $fr > $OUT/c5_f${fr}_side_$rep.json 2>> $OUT/ab.err
    SS_SIDE_LANE=0 timeout 300 $B --config 5 --steps $st --frames $fr > $OUT/c5_f${fr}_pass_$rep.json 2>> $OUT/ab.err
  done
done
timeout 300 python bench.py --no-cpu-baseline --no-also --warmup 5 --config 3 --gpus 1 --steps 100 --sub > $OUT/c3_prod_parity.json 2>> $OUT/ab.err
timeout 300 python bench.py --no-cpu-baseline --no-also --warmup 5 --config 5 --gpus 1 --steps 100 --sub > $OUT/c5_prod_parity.json 2>> $OUT/ab.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r04_s12/c*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j['ms_per_step'], j['value'], [(k['slot'], k['us']) for k in j['roofline']['kernels']], j['config'].get('tiles'), j['config'].get('host_enqueue_ms_per_step'), json.dumps(j.get('parity'))[:300] if j.get('parity') else '')
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
cat $OUT/rc.txt; tail -15 $OUT/pytest_gpu.txt | cut -c1-400; tail -5 $OUT/ab.err | cut -c1-300
