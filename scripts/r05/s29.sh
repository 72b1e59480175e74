#!/bin/bash
# round 5, session 29: where a pair of tiles spends its 6-15 us — four stamps per tile (start, first pass of phase 1, phase 1, phase 2)
OUT=gpurun_out/r05_s29
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
LIB=scripts/ab/libspecscan_base.so
stamps() {  # tag frames env...
  tag=$1; f=$2; shift 2
  env "$@" SS_STEP_STAMPS=$OUT/stamps_${tag}_f$f.txt timeout 300 python bench.py --config 3 --frames $f --gpus 1 --sub --no-parity --steps 100 --warmup 5 --no-cpu-baseline --lib $LIB > $OUT/st_${tag}_f$f.json 2> $OUT/st_${tag}_f$f.err
  echo "== stamps $tag, $f frames"
  python scripts/analyze_step_stamps.py $OUT/stamps_${tag}_f$f.txt 256 list 2>&1 | tee $OUT/stamps_${tag}_f${f}_summary.txt
}
stamps FPED_l0 64 'SS_STEP_ORDER=F*,P*,E*,D*' SS_LIST_FIRST=0
stamps FED 128 SS_X=0
