#!/bin/bash
# one PMC group + the kernel stats of a short bench run, kernel names with their template arguments:
#   tools/pmc_k.sh <tag> "<counters>" [bench args...]
tag=$1; grp=$2; shift; shift
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- $B > $out/bench.json 2> $out/stats.log
f=$(find $out -name "*kernel_stats.csv" | head -1)
python - "$f" <<'P'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    print(r["Name"].replace("(anonymous namespace)::", "")[:50], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us", r["Percentage"])
P
rocprofv3 --pmc $grp --output-format csv -d $out/pmc -o pmc -- $B > /dev/null 2> $out/pmc.log
f=$(find $out/pmc -name "*counter_collection.csv" | head -1)
python - "$f" <<'P'
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for row in csv.DictReader(open(sys.argv[1])):
    name = (row.get("Kernel_Name") or "").replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:34]
    a = acc[name][row.get("Counter_Name")]
    a[0] += float(row.get("Counter_Value") or 0); a[1] += 1
for k, cs in sorted(acc.items(), key=lambda kv: -sum(v[0] for v in kv[1].values()))[:8]:
    print(k, {c: round(t / n) for c, (t, n) in cs.items()})
P
