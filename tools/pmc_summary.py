"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel: mean counter value per dispatch.
usage: python tools/pmc_summary.py <counter_collection.csv> [more.csv ...]"""
import csv
import sys
from collections import defaultdict

for path in sys.argv[1:]:
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    with open(path) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name") or row.get("Kernel Name") or ""
            short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
            cname = row.get("Counter_Name") or row.get("Counter Name")
            val = float(row.get("Counter_Value") or row.get("Counter Value") or 0)
            a = acc[short][cname]
            a[0] += val
            a[1] += 1
    print(path)
    for k, cs in sorted(acc.items(), key=lambda kv: -sum(v[0] for v in kv[1].values())):
        for c, (tot, n) in cs.items():
            print(f"  {k:48s} {c:12s} dispatches {n:6d}  mean {tot / n:14.3f}  total {tot:16.1f}")
