#!/bin/bash
# round 5, session 1: the radix-8 DIF lab (scripts/ubench/dif8_lab) — 65536-point frames without a work buffer: timing of the four
# variants at 128 / 256 / 512 / 64 frames, the same without the ceiling subtraction, kernel statistics, fabric bytes and SQ counters
OUT=gpurun_out/r05_s1
mkdir -p $OUT
R=/root/repo
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 300 scripts/ubench/dif8_lab > $OUT/dif8_lab.txt 2>&1
# (dif8_lab_nothr: the variant without the ceiling subtraction — since session 18 the kernel never subtracts)
cd /tmp
for v in 0 1; do
  DIF8_ONLY=$v timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof$v -- $R/scripts/ubench/dif8_lab 128 > $R/$OUT/prof$v.log 2>&1
  cp $R/$OUT/prof$v/*/*_kernel_stats.csv $R/$OUT/dif8_kernel_stats_v$v.csv 2>/dev/null
  rm -rf $R/$OUT/prof$v
done
for k in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  tag=$(echo $k | tr ' ' '_' | cut -c1-40)
  for v in 1 3; do
    DIF8_ONLY=$v timeout 200 rocprofv3 --pmc $k --kernel-trace --output-format csv -d $R/$OUT/pmc_${tag}_v$v -- $R/scripts/ubench/dif8_lab 128 > $R/$OUT/pmc_${tag}_v$v.log 2>&1
    cp $R/$OUT/pmc_${tag}_v$v/*/*_counter_collection.csv $R/$OUT/dif8_pmc_${tag}_v$v.csv 2>/dev/null
    rm -rf $R/$OUT/pmc_${tag}_v$v
  done
done
cd $R
cat $OUT/dif8_lab.txt
cat $OUT/dif8_lab_nothr.txt
head -5 $OUT/dif8_kernel_stats_v*.csv
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/r05_s1/dif8_pmc_*.csv')):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        if 'dif8' not in row.get('Kernel_Name', ''): continue
        a = acc[row['Counter_Name']]
        a[0] += float(row['Counter_Value']); a[1] += 1
    print(f.split('/')[-1], {k: round(v[0] / max(v[1], 1), 1) for k, v in acc.items()}, 'launches', max([v[1] for v in acc.values()] + [0]))
PY
