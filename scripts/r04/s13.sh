#!/bin/bash
# round 4, session 13: the side lane with cheaper events (no system-scope fence) and without the memset, against the passengers' form;
# a kernel trace of the side-lane run to see which launches really run side by side
OUT=gpurun_out/r04_s13
mkdir -p $OUT
R=/root/repo
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_stated_configs.py tests/test_gpu_cull.py -m gpu -q -x --timeout 600 -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1
echo "tests rc=$?" >> $OUT/rc.txt
B="python bench.py --no-cpu-baseline --no-also --no-parity --warmup 5 --diag-lib --gpus 1"
for rep in 1 2; do
  timeout 300 $B --config 3 --steps 100 > $OUT/c3_f128_side_$rep.json 2>> $OUT/ab.err
  SS_SIDE_LANE=0 timeout 300 $B --config 3 --steps 100 > $OUT/c3_f128_pass_$rep.json 2>> $OUT/ab.err
  timeout 300 $B --config 3 --steps 400 --frames 16 > $OUT/c3_f16_side_$rep.json 2>> $OUT/ab.err
  timeout 300 $B --config 5 --steps 100 > $OUT/c5_f16_side_$rep.json 2>> $OUT/ab.err
  SS_SIDE_LANE=0 timeout 300 $B --config 5 --steps 100 > $OUT/c5_f16_pass_$rep.json 2>> $OUT/ab.err
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/trace -- python $R/bench.py --no-cpu-baseline --no-also --no-parity --warmup 5 --diag-lib --gpus 1 --config 3 --steps 40 --preheat-ms 50 --no-kernel-timing > $R/$OUT/trace.log 2>&1
cd $R
python - <<'PY'
import csv, glob, json, os
for f in sorted(glob.glob('gpurun_out/r04_s13/c*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), j['ms_per_step'], j['value'], [(k['slot'], k['us']) for k in j['roofline']['kernels']], j['config'].get('host_enqueue_ms_per_step'))
    except Exception as e:
        print(os.path.basename(f), 'ERR', e)
t = glob.glob('gpurun_out/r04_s13/trace/*/*_kernel_trace.csv')
if t:
    rows = [r for r in csv.DictReader(open(t[0])) if 'ss::' in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    rows = rows[-60:-20]
    t0 = int(rows[0]['Start_Timestamp'])
    for r in rows:
        print('%-28s q%-3s grid %-8s start %8.1f end %8.1f us' % (r['Kernel_Name'][:28], r['Queue_Id'], r['Grid_Size'], (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3))
PY
rm -rf $OUT/trace
cat $OUT/rc.txt; tail -4 $OUT/pytest_gpu.txt | cut -c1-300; tail -3 $OUT/ab.err | cut -c1-300
