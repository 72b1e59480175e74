"""Average the rocprofv3 --pmc counter_collection CSVs per kernel."""
import collections
import csv
import sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"][:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if not k.startswith(("void ss::", "ss::")):
        continue
    print(k)
    for c, vals in sorted(v.items()):
        print("    %-26s %.5g  (n=%d)" % (c, sum(vals) / len(vals), len(vals)))
