#!/bin/bash
# k_gemm_wg: item tiles per workgroup (arguments like 1,2,2 = tuning values wg_ti_out, wg_ti_in, wg_ti_dil, set through
# PS_WG_TI_OUT / PS_WG_TI_IN / PS_WG_TI_DIL at handle creation): bit-identity tests and kernel times per combination
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for ti in "$@"; do
  IFS=, read o i d <<< "$ti"; export PS_WG_TI_OUT=$o PS_WG_TI_IN=$i PS_WG_TI_DIL=$d
  echo "== wg_ti out,in,dil = $ti: $(python -m pytest tests/test_lmconv_gpu.py -x -q -k 'launch_forms or workgroup_gemm' 2>&1 | tail -1)"
  out=gpurun_out/ti_$ti; mkdir -p $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $out/bench.json 2> $out/stats.log
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'P'
import csv, sys
tot = 0; n = 0
for r in csv.DictReader(open(sys.argv[1])):
    nm = r["Name"].replace("(anonymous namespace)::", "")
    if "k_gemm" in nm:
        print("    ", nm[:36], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
        tot += float(r["TotalDurationNs"]); n += int(r["Calls"])
print("     all k_gemm*: avg %.1f us over %d calls" % (tot / n / 1e3, n))
P
done
