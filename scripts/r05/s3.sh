#!/bin/bash
# round 5, session 3: dif8_lab with the two-residues-per-workgroup variant (index 4) beside the others
OUT=gpurun_out/r05_s3
mkdir -p $OUT
R=/root/repo
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 300 scripts/ubench/dif8_lab 128 256 512 64 > $OUT/dif8_lab.txt 2>&1
cd /tmp
for k in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" FETCH_SIZE WRITE_SIZE; do
  tag=$(echo $k | tr ' ' '_' | cut -c1-40)
  DIF8_ONLY=4 timeout 200 rocprofv3 --pmc $k --kernel-trace --output-format csv -d $R/$OUT/pmc_${tag} -- $R/scripts/ubench/dif8_lab 128 > $R/$OUT/pmc_${tag}.log 2>&1
  cp $R/$OUT/pmc_${tag}/*/*_counter_collection.csv $R/$OUT/dif8x2_pmc_${tag}.csv 2>/dev/null
  rm -rf $R/$OUT/pmc_${tag}
done
cd $R
cat $OUT/dif8_lab.txt
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/r05_s3/dif8x2_pmc_*.csv')):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        if 'dif8' not in row.get('Kernel_Name', ''): continue
        a = acc[row['Counter_Name']]
        a[0] += float(row['Counter_Value']); a[1] += 1
    print(f.split('/')[-1], {k: round(v[0] / max(v[1], 1), 1) for k, v in acc.items()})
PY
