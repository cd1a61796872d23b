#!/bin/bash
# per-kernel durations of the splat-only loop (config C2): tools/splat_prof.sh [B] [tag]
B=${1:-32}; tag=${2:-splat}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o sp -- python tools/splat_bench.py $B 10 > $out/run.log 2> $out/stats.log
tail -1 $out/run.log
f=$(find $out -name "*kernel_stats.csv" | head -1)
python - "$f" <<'P'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print(r["Name"][:70], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us", r["Percentage"])
P
