#!/bin/bash
# round 5, session 26: the fold's launches — the detect workgroups that take the list's first pairs dispatched AHEAD of the fold's
# workgroups (they wait on memory for 7-15 us and need no VALU to speak of: session 25's stamps), everything else behind; instruction
# cache counters of the launch (60 KB of code on a 64 KB cache shared by two CUs)
OUT=gpurun_out/r05_s26
mkdir -p $OUT
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
LIB=scripts/ab/libspecscan_base.so
run() {  # tag frames env...
  tag=$1; f=$2; shift 2
  env "$@" timeout 300 python bench.py --config 3 --frames $f --gpus 1 --sub --no-parity --steps 200 --warmup 5 --no-cpu-baseline --lib $LIB > $OUT/${tag}_f$f.json 2> $OUT/${tag}_f$f.err
  python - <<PY
import json
try:
    j = json.loads(open('$OUT/${tag}_f$f.json').read().strip().splitlines()[-1])
    print('f=$f $tag', j['ms_per_step'], j['value'], [(k['slot'], k['us']) for k in j['roofline']['kernels']])
except Exception as e:
    print('f=$f $tag ERR', e, open('$OUT/${tag}_f$f.err').read()[-600:])
PY
}
for f in 128 64 256 512 16; do
  run FED $f SS_X=0
  run D64FPED $f 'SS_STEP_ORDER=D64,F*,P*,E*,D*'
  run D64FEPD $f 'SS_STEP_ORDER=D64,F*,E*,P*,D*'
  run PD64FED $f 'SS_STEP_ORDER=P*,D64,F*,E*,D*'
  run D256FPED_l256 $f 'SS_STEP_ORDER=D256,F*,P*,E*,D*' SS_LIST_FIRST=256
done
stamps() {  # tag frames env...
  tag=$1; f=$2; shift 2
  env "$@" SS_STEP_STAMPS=$OUT/stamps_${tag}_f$f.txt timeout 300 python bench.py --config 3 --frames $f --gpus 1 --sub --no-parity --steps 100 --warmup 5 --no-cpu-baseline --lib $LIB > $OUT/st_${tag}_f$f.json 2> $OUT/st_${tag}_f$f.err
  echo "== stamps $tag, $f frames"
  python scripts/analyze_step_stamps.py $OUT/stamps_${tag}_f$f.txt 2>&1 | tee $OUT/stamps_${tag}_f${f}_summary.txt
}
stamps D64FPED 128 'SS_STEP_ORDER=D64,F*,P*,E*,D*'
stamps D64FPED 64 'SS_STEP_ORDER=D64,F*,P*,E*,D*'
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_INST[A-Z_]*\|SQ_INST_CYCLES[A-Z_]*\|SQ_BUSY_CY[A-Z_]*\|SQ_WAVE_CYCLES\|SQ_ACTIVE_INST[A-Z_]*" | sort -u > /root/repo/$OUT/counters_avail.txt
cat /root/repo/$OUT/counters_avail.txt | tr '\n' ' '; echo
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  k=$(echo $set | tr ' ' '_')
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /root/repo/$OUT/pmc_$k -- python /root/repo/bench.py --config 3 --frames 128 --gpus 1 --sub --no-parity --steps 30 --warmup 5 --preheat-ms 0 --no-cpu-baseline > /root/repo/$OUT/pmc_$k.log 2>&1
  cp /root/repo/$OUT/pmc_$k/*/*_counter_collection.csv /root/repo/$OUT/pmc_$k.csv 2>/dev/null
  rm -rf /root/repo/$OUT/pmc_$k
  python - <<PY
import csv, collections
try:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open('/root/repo/$OUT/pmc_$k.csv')):
        name = r['Kernel_Name']
        if 'k_scan_step' not in name: continue
        acc[name[-40:]][r['Counter_Name']].append(float(r['Counter_Value']))
    for n, d in acc.items():
        print(n, {c: (len(v), sum(v)/len(v)) for c, v in d.items()})
except Exception as e:
    print('pmc $k ERR', e, open('/root/repo/$OUT/pmc_$k.log').read()[-500:])
PY
done
