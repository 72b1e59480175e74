#!/bin/bash
# round 4, session 15: the whole GPU suite on the tree with the fused plan, the formed taps and the per-column maxima through LDS
# (SS_SEGMAX_LDS=1); A/B of the maxima: scripts/ab/libspecscan_base.so (LDS) against libspecscan_segdpp.so (registers, as before)
OUT=gpurun_out/r04_s15
mkdir -p $OUT
R=/root/repo
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1
echo "tests rc=$?" >> $OUT/rc.txt
B="python bench.py --no-cpu-baseline --no-also --no-parity --gpus 1"
for rep in 1 2 3; do
  timeout 300 $B --lib scripts/ab/libspecscan_base.so --steps 200 --warmup 20 > $OUT/c2_lds_k200_$rep.json 2>> $OUT/ab.err
  timeout 300 $B --lib scripts/ab/libspecscan_segdpp.so --steps 200 --warmup 20 > $OUT/c2_dpp_k200_$rep.json 2>> $OUT/ab.err
  timeout 300 $B --lib scripts/ab/libspecscan_base.so --steps 20 --warmup 5 > $OUT/c2_lds_k20_$rep.json 2>> $OUT/ab.err
  timeout 300 $B --lib scripts/ab/libspecscan_segdpp.so --steps 20 --warmup 5 > $OUT/c2_dpp_k20_$rep.json 2>> $OUT/ab.err
done
timeout 300 $B --lib scripts/ab/libspecscan_base.so --steps 100 --warmup 20 --no-cull > $OUT/c2_lds_nocull.json 2>> $OUT/ab.err
timeout 300 $B --lib scripts/ab/libspecscan_segdpp.so --steps 100 --warmup 20 --no-cull > $OUT/c2_dpp_nocull.json 2>> $OUT/ab.err
python - <<'PY'
import glob, json, os
for f in sorted(glob.glob('gpurun_out/r04_s15/c2*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j['ms_per_step'], j['value'], j['roofline']['frac'], [(k['slot'], k['us']) for k in j['roofline']['kernels']])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
cat $OUT/rc.txt; tail -6 $OUT/pytest_gpu.txt | cut -c1-300; tail -3 $OUT/ab.err | cut -c1-300
