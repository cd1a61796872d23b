#!/bin/bash
# per-kernel durations of bench.py's step (kernel trace only): tools/bench_stats.sh <tag> [bench args...]
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra $* > $out/bench.json 2> $out/stats.log
f=$(find $out -name "*kernel_stats.csv" | head -1)
python - "$f" <<'P'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:10]:
    print(r["Name"].replace("(anonymous namespace)::", "")[:50], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us", r["Percentage"])
P
