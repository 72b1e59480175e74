#!/bin/bash
# round 2, GPU session 14: SQ counters of k_scan_step (what is the step waiting for?)
set -x
OUT=gpurun_out/r02_s14; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_sq1 -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --preheat-ms 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/pmc_sq1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_sq2 -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --preheat-ms 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/pmc_sq2.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_sq3 -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --preheat-ms 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/pmc_sq3.log 2>&1
SS_PIPELINE=0 timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_sq1_nopipe -- python $GRAFT_REPO_ROOT/bench.py --diag-lib --steps 30 --warmup 5 --preheat-ms 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/pmc_sq1_nopipe.log 2>&1
SS_PIPELINE=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace_nopipe -- python $GRAFT_REPO_ROOT/bench.py --diag-lib --steps 100 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/trace_nopipe.log 2>&1
cd $GRAFT_REPO_ROOT; ls $OUT/*/*/ | head -30; tail -3 $OUT/pmc_sq3.log
