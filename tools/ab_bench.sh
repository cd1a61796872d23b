#!/bin/bash
# A/B of library builds / tuning values on one box: tools/ab_bench.sh <rounds> <steps> name=ENV1=v+ENV2=v ... ; alternating runs of the
# headline step (python bench.py --no-extra --no-cpu-baseline --no-live-pmc), one JSON line per run under gpurun_out/ab_<name>_<round>.json
# and a summary (ms per step, event-timed us per launch by kernel) on stdout.
rounds=$1; steps=$2; shift 2
mkdir -p gpurun_out
for r in $(seq 1 $rounds); do
  for cfg in "$@"; do
    name=${cfg%%=*}; envs=${cfg#*=}; [ "$envs" = "$cfg" ] && envs=""
    env $(echo $envs | tr '+' ' ') python bench.py --no-extra --no-cpu-baseline --no-live-pmc --steps $steps --warmup 5 $PS_AB_ARGS \
        > gpurun_out/ab_${name}_${r}.json 2> gpurun_out/ab_${name}_${r}.err || tail -5 gpurun_out/ab_${name}_${r}.err
  done
done
python - "$@" <<'PY'
import glob, json, sys
for cfg in sys.argv[1:]:
    name = cfg.split("=")[0]
    for path in sorted(glob.glob(f"gpurun_out/ab_{name}_*.json")):
        try:
            d = json.loads([l for l in open(path) if l.startswith("{")][0])
        except Exception as e:
            print(name, path, "no line", e); continue
        r = d["roofline"]
        ks = " ".join(f"{k['kernel']}={k['avg_launch_us']:.1f}" for k in r.get("kernels", []) if isinstance(k, dict))
        ph = r.get("phases") or {}
        print(f"{name:12s} {d['ms_per_step']:7.3f} ms  col {r.get('avg_launch_us')}us  phases {ph.get('prefix_pass_and_small_kernels_ms')}/{ph.get('column_launches_ms')}  {ks}  [{d['library']['build'][-40:]}]")
PY
