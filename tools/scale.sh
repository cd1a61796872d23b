#!/bin/bash
# The 1 -> N GPU curve of the headline, in the three forms BASELINE.json quotes (run on a multi-GPU MI355X node, from the repository root):
#   weak    every rank renders its own C5 (8 sources x 16 views = 128 views per GPU per step)          python bench.py --views 128
#   strong  C5 as ONE job of 128 views dealt round-robin over the ranks                                --total-views 128
#   circle  C4: the 64-frame 'C' circle from one source, frames dealt round-robin                      --trajectory circle --frames 64
# One process per GPU under torch.distributed.run (RCCL over xGMI, 127.0.0.1 rendezvous); for every run the frames/s of the whole
# job and the world size the backend itself reports (`collective.world_size` of the bench line).
#   tools/scale.sh [GPU counts, default "1 2 4 8"]       PS_SCALE_STEPS / PS_SCALE_WARMUP: steps per run (20 / 5)
set -u
cd "$(dirname "$0")/.."
NS=${*:-1 2 4 8}
STEPS=${PS_SCALE_STEPS:-20}; WARM=${PS_SCALE_WARMUP:-5}
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/scale
have=$(python -c 'import torch; print(torch.cuda.device_count())')
for form in weak strong circle; do
  case $form in
    weak) extra="--views 128" ;;
    strong) extra="--total-views 128" ;;
    circle) extra="--trajectory circle --frames 64" ;;
  esac
  for n in $NS; do
    if [ "$n" -gt "$have" ]; then echo "$form N=$n: skipped ($have GPU(s) on this node)"; continue; fi
    port=$((29600 + RANDOM % 300))
    log=gpurun_out/scale/${form}_n$n.json
    if [ "$n" -eq 1 ]; then   # a single rank goes through the RCCL group as well: the same code path as N > 1
      PS_BENCH_FORCE_COLLECTIVE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$port python bench.py --gpus 1 --steps $STEPS --warmup $WARM --no-cpu-baseline --no-extra $extra > $log 2> ${log%.json}.err
    else
      python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port bench.py --gpus $n --steps $STEPS --warmup $WARM --no-cpu-baseline --no-extra $extra > $log 2> ${log%.json}.err
    fi
    python - "$form" "$n" "$log" <<'P'
import json, sys
form, n, log = sys.argv[1:4]
try:
    r = json.loads([ln for ln in open(log) if ln.startswith("{")][-1])
    c = r.get("collective", {})
    print("%-6s N=%s: %9.1f frames/s  %8.3f ms/step  scaling=%s  backend=%s world_size(reported)=%s" % (
        form, n, r["value"], r["ms_per_step"], r["scaling"], c.get("backend"), c.get("world_size")))
except Exception as e:
    print("%-6s N=%s: failed (%r) -- see %s" % (form, n, e, log.replace(".json", ".err")))
P
  done
done
