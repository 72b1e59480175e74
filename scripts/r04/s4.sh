#!/bin/bash
# round 4, session 4: the reference-NaN tests again; config 5 in two passes with the sharded counters; what the 1024-point column
# tile spends its time on (ablation builds); fabric bytes of config 5 in two and in three passes; what the per-launch events cost the
# headline
OUT=gpurun_out/r04_s4
mkdir -p $OUT
R=/root/repo
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_degenerate_input.py "tests/test_gpu_wait_bound.py::test_consumers_that_never_hear_from_the_plan_workgroups_make_the_plan_themselves" tests/test_gpu_cull.py -m gpu -q --timeout 600 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "tests rc=$?" >> $OUT/rc.txt
B="python bench.py --no-cpu-baseline --no-also --no-parity --warmup 5"
for i in 1 2; do
  timeout 300 $B --config 5 --gpus 1 --steps 100 > $OUT/c5_two$i.json 2>> $OUT/ab.err
  SS_FFT_TWOPASS=0 timeout 300 $B --diag-lib --config 5 --gpus 1 --steps 100 > $OUT/c5_three$i.json 2>> $OUT/ab.err
done
for v in c1024nowin c1024nostore c1024noload c1024none; do
  timeout 300 $B --lib scripts/ab/libspecscan_$v.so --config 5 --gpus 1 --steps 100 > $OUT/c5_$v.json 2>> $OUT/ab.err
done
cd /tmp
for mode in two three; do
  EXTRA="--config 5 --gpus 1 --sub --no-parity"; ENVV=""
  [ $mode = three ] && EXTRA="$EXTRA --diag-lib" && export SS_FFT_TWOPASS=0
  for k in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --pmc $k --kernel-trace --output-format csv -d $R/$OUT/pmc_${k}_$mode -- python $R/bench.py $EXTRA --steps 30 --warmup 5 --preheat-ms 0 --no-cpu-baseline > $R/$OUT/pmc_${k}_$mode.log 2>&1
  done
  unset SS_FFT_TWOPASS
  cp $R/$OUT/pmc_FETCH_SIZE_$mode/*/*_counter_collection.csv $R/$OUT/s4_cfg5_${mode}_pmc_fetch.csv
  cp $R/$OUT/pmc_WRITE_SIZE_$mode/*/*_counter_collection.csv $R/$OUT/s4_cfg5_${mode}_pmc_write.csv
  rm -rf $R/$OUT/pmc_FETCH_SIZE_$mode $R/$OUT/pmc_WRITE_SIZE_$mode
done
cd $R
for i in 1 2; do
  timeout 300 $B --steps 200 > $OUT/k200_every8_$i.json 2>> $OUT/ab.err
  timeout 300 $B --steps 200 --time-every 50 > $OUT/k200_every50_$i.json 2>> $OUT/ab.err
  timeout 300 $B --steps 200 --no-kernel-timing > $OUT/k200_notiming_$i.json 2>> $OUT/ab.err
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r04_s4/c5*.json')) + sorted(glob.glob('gpurun_out/r04_s4/k200*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j['ms_per_step'], j['value'], j['roofline_chain']['frac'], [(k['slot'], k['us']) for k in j['roofline']['kernels']])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
PY
python scripts/pmc_summary.py $OUT/s4_cfg5_two_pmc_fetch.csv $OUT/s4_cfg5_two_pmc_write.csv
python scripts/pmc_summary.py $OUT/s4_cfg5_three_pmc_fetch.csv $OUT/s4_cfg5_three_pmc_write.csv
cat $OUT/rc.txt; tail -15 $OUT/pytest.log | cut -c1-300
