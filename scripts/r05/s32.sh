#!/bin/bash
# round 5, session 32: per-workgroup and per-tile stamps of the launches of the other chains, read the way session 24-30 read the fold's:
# 2^20 points x 16 frames (config 5: the row launch with its passengers), 65536 points CF32 (KIND 7, one launch per call), the default line
OUT=gpurun_out/r05_s32
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
LIB=scripts/ab/libspecscan_base.so
stamps() {  # tag tcols bench-args...
  tag=$1; tc=$2; shift 2
  SS_STEP_STAMPS=$OUT/stamps_$tag.txt timeout 300 python bench.py "$@" --gpus 1 --no-parity --steps 100 --warmup 5 --no-cpu-baseline --lib $LIB > $OUT/st_$tag.json 2> $OUT/st_$tag.err
  echo "== stamps $tag"; tail -c 200 $OUT/st_$tag.err
  python scripts/analyze_step_stamps.py $OUT/stamps_$tag.txt $tc 2>&1 | tee $OUT/stamps_${tag}_summary.txt
}
stamps cfg5_f16 4096 --config 5 --frames 16 --sub
stamps cfg5_f64 4096 --config 5 --frames 64 --sub
stamps cfg3_cf32_f128 256 --config 3 --frames 128 --sub --fmt cf32
stamps default 32 --no-also
