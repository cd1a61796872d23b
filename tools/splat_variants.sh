#!/bin/bash
# composite kernel time for every tuning build under gpurun_exp/ (PS_HIP_LIB)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in "" $(ls gpurun_exp/*.so); do
  tag=$(basename "${lib:-base}" .so); out=gpurun_out/spv_$tag; mkdir -p $out
  PS_HIP_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o s -- python tools/splat_bench.py 32 10 > $out/log 2>&1
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  echo "== $tag: $(grep ms/batch $out/log | tail -1)"
  python - "$f" <<'P'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:3]:
    print("    ", r["Name"].replace("(anonymous namespace)::", "")[:40], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
P
done
