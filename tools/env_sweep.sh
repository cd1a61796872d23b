#!/bin/bash
# A/B of tuning values read from the environment at handle creation (PS_<NAME>): for every "name:VAR=val,VAR=val" argument the
# un-profiled bench line (ms per step) and the per-kernel averages of a rocprofv3 kernel trace of the same command.
#   tools/env_sweep.sh base: sort1:PS_ITEM_SORT=1 "ti2:PS_ITEM_SORT=2,PS_WG_TI_OUT=2"
# PS_SWEEP_BENCH_ARGS: extra bench.py arguments (default: --views 128); PS_SWEEP_TRACE_ONLY=1: the kernel trace only.  Runs on the GPU box (gpurun), from the repository root.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS=${PS_SWEEP_BENCH_ARGS:---views 128}
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  out=gpurun_out/sweep_$name; mkdir -p $out
  (
    IFS=, ; for kv in $envs; do [ -n "$kv" ] && export "$kv"; done; unset IFS
    if [ -z "${PS_SWEEP_TRACE_ONLY:-}" ]; then python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra $ARGS > $out/bench.json 2> $out/bench.err; else rm -f $out/bench.json; fi
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra $ARGS > $out/bench_trace.json 2> $out/stats.log
  )
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  python - "$name" "$envs" "$out/bench.json" "$f" <<'P'
import csv, json, sys
name, envs, bj, f = sys.argv[1:5]
try:
    r = json.loads(open(bj).read().strip().splitlines()[-1])
    head = "%.3f ms/step  %.1f frames/s  column launch %.1f us" % (r["ms_per_step"], r["value"], r["roofline"]["avg_launch_us"])
except Exception as e:
    head = "bench failed: %r" % (e,)
print("== %s [%s]: %s" % (name, envs, head))
rows = []
for r in csv.DictReader(open(f)):
    nm = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("pslm::", "")
    rows.append((float(r["TotalDurationNs"]), nm.split("(")[0][:40], int(r["Calls"]), float(r["AverageNs"]) / 1e3))
for tot, nm, calls, avg in sorted(rows, reverse=True)[:12]:
    print("     %-40s %6d calls  avg %8.1f us  total %8.2f ms" % (nm, calls, avg, tot / 1e6))
P
done
