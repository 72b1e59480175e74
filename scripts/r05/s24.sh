#!/bin/bash
# round 5, session 24: per-workgroup stamps of one fold launch (65536 points, int8) at 96 / 112 / 128 frames per call: when do the
# fold's workgroups and the passengers start and end, and on which CUs
OUT=gpurun_out/r05_s24
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
for f in 96 112 128 512; do
  SS_STEP_STAMPS=$OUT/stamps_f$f.txt timeout 300 python bench.py --config 3 --frames $f --gpus 1 --sub --no-parity --steps 100 --warmup 5 --no-cpu-baseline --lib scripts/ab/libspecscan_base.so > $OUT/f$f.json 2> $OUT/f$f.err
  echo "== $f frames"; tail -c 300 $OUT/f$f.err
  python scripts/analyze_step_stamps.py $OUT/stamps_f$f.txt 2>&1 | tee $OUT/stamps_f${f}_summary.txt
done
