#!/bin/bash
# round 3, GPU session 21: whole GPU suite (with the timed-path parity cases), smoke against oracle/_ref, the benchmark line
# (200 and 20 steps, no-cull, detect mode), rocprofv3 kernel statistics + trace and PMC traffic of the same command
OUT=gpurun_out/r03_s21; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests -q -m gpu -s > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
grep -h "^\[" $OUT/pytest_gpu.txt > $OUT/stated_configs_parity.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; tail -3 $OUT/smoke.txt
timeout 120 python bench.py --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err; cut -c1-200 $OUT/bench_default.json
timeout 120 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/bench_k20.json 2> $OUT/bench_k20.err; cut -c1-200 $OUT/bench_k20.json
timeout 120 python bench.py --no-cpu-baseline --no-cull > $OUT/bench_nocull.json 2> $OUT/bench_nocull.err; cut -c1-200 $OUT/bench_nocull.json
timeout 120 python bench.py --no-cpu-baseline --no-psd-out > $OUT/bench_detect.json 2> $OUT/bench_detect.err; cut -c1-200 $OUT/bench_detect.json
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -- python $R/bench.py --steps 200 --warmup 10 --no-cpu-baseline > $R/$OUT/prof.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$OUT/pmc_fetch -- python $R/bench.py --steps 30 --warmup 5 --preheat-ms 0 --no-cpu-baseline > $R/$OUT/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$OUT/pmc_write -- python $R/bench.py --steps 30 --warmup 5 --preheat-ms 0 --no-cpu-baseline > $R/$OUT/pmc_write.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$OUT/pmc_fetch_nocull -- python $R/bench.py --steps 30 --warmup 5 --preheat-ms 0 --no-cpu-baseline --no-cull > $R/$OUT/pmc_fetch_nocull.log 2>&1
cd $R
python scripts/launches_in_flight.py $OUT/prof/*/*_kernel_trace.csv | tee $OUT/launches_in_flight.txt
cp $OUT/prof/*/*_kernel_stats.csv $OUT/kernel_stats.csv
for k in pmc_fetch pmc_write pmc_fetch_nocull; do cp $OUT/$k/*/*_counter_collection.csv $OUT/$k.csv; done
rm -rf $OUT/prof $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_fetch_nocull
head -4 $OUT/kernel_stats.csv | cut -c1-200
python - <<'PY'
import csv
for kind in ("pmc_fetch", "pmc_write", "pmc_fetch_nocull"):
    rows = [r for r in csv.DictReader(open(f"gpurun_out/r03_s21/{kind}.csv")) if "k_scan_step" in r["Kernel_Name"]]
    by = {}
    for r in rows:
        by.setdefault(int(r["Grid_Size"]), []).append(float(r["Counter_Value"]))
    for g, v in sorted(by.items()):
        print(kind, "grid", g, "launches", len(v), "mean KiB", round(sum(v) / len(v), 1))
PY
