#!/bin/bash
# kernel stats of a short bench run for every tuning build under gpurun_exp/ (PS_HIP_LIB)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in "" $(ls gpurun_exp/*.so); do
  tag=$(basename "${lib:-base}" .so); out=gpurun_out/exp_$tag; mkdir -p $out
  PS_HIP_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra $* > $out/bench.json 2> $out/stats.log
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  echo "== $tag"
  python - "$f" <<'P'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    n = r["Name"].replace("(anonymous namespace)::", "")
    if "gemm" in n or "column_tp" in n: print("  ", n[:40], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
P
done
