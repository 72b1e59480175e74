#!/bin/bash
# round 5, session 13: the headline's fabric bytes by role — PMC passes (FETCH_SIZE, WRITE_SIZE; separate passes) of the default command
# line as it ships, with every tile evaluated (--no-cull), and with roles taken out (SS_ABLATE_ROLES of the diagnostics build: 1 = no
# detect role, 2 = no emit role, 3 = neither): what each role moves per launch is the difference
OUT=gpurun_out/r05_s13
mkdir -p $OUT
R=/root/repo
cd /tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
pass() {  # tag extra-args env...
  tag=$1; extra=$2; shift 2
  for k in FETCH_SIZE WRITE_SIZE; do
    env "$@" timeout 200 rocprofv3 --pmc $k --kernel-trace --output-format csv -d $R/$OUT/pmc_${k}_$tag -- python $R/bench.py --steps 30 --warmup 5 --preheat-ms 0 --no-cpu-baseline --no-also --no-parity --no-live-pmc --sub $extra > $R/$OUT/pmc_${k}_$tag.log 2>&1
    cp $R/$OUT/pmc_${k}_$tag/*/*_counter_collection.csv $R/$OUT/s13_${tag}_pmc_$(echo $k | tr A-Z a-z | cut -d_ -f1).csv 2>/dev/null
    rm -rf $R/$OUT/pmc_${k}_$tag
  done
}
pass ships "" SS_X=0
pass nocull "--no-cull" SS_X=0
pass diag_all "--diag-lib" SS_X=0
pass diag_nodet "--diag-lib" SS_ABLATE_ROLES=1
pass diag_noemit "--diag-lib" SS_ABLATE_ROLES=2
pass diag_neither "--diag-lib" SS_ABLATE_ROLES=3
cd $R
python3 - <<'PY'
import csv, glob
def mean(path, shape=None):
    vals = [float(r['Counter_Value']) for r in csv.DictReader(open(path)) if 'k_scan_step' in r['Kernel_Name'] and (shape is None or int(r['Grid_Size']) == shape)]
    return (sum(vals) / len(vals), len(vals)) if vals else (0.0, 0)
for tag in ['ships', 'nocull', 'diag_all', 'diag_nodet', 'diag_noemit', 'diag_neither']:
    try:
        out = {}
        for shape in (None,):
            f, nf = mean(f'gpurun_out/r05_s13/s13_{tag}_pmc_fetch.csv')
            w, nw = mean(f'gpurun_out/r05_s13/s13_{tag}_pmc_write.csv')
        # steady-state shapes only: the most frequent grid size
        import collections
        rows = [r for r in csv.DictReader(open(f'gpurun_out/r05_s13/s13_{tag}_pmc_fetch.csv')) if 'k_scan_step' in r['Kernel_Name']]
        common = collections.Counter(int(r['Grid_Size']) for r in rows).most_common(1)[0][0]
        f, nf = mean(f'gpurun_out/r05_s13/s13_{tag}_pmc_fetch.csv', common)
        w, nw = mean(f'gpurun_out/r05_s13/s13_{tag}_pmc_write.csv', common)
        print(f'{tag:14s} grid {common // 512:5d} wgs  launches {nf:3d}  FETCH {f:9.1f} KiB  WRITE {w:9.1f} KiB  bytes {(2 * f + w) * 1024 / 1e6:7.2f} MB per launch = {(2 * f + w) * 1024 / 100663296:.3f} x algorithmic')
    except Exception as e:
        print(tag, 'ERR', e)
PY
