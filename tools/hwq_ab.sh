#!/bin/bash
# Tuning aid: the step with the RCCL path forced on one rank (PS_BENCH_FORCE_COLLECTIVE=1) for the number of hardware queues the HIP
# runtime gives the process (GPU_MAX_HW_QUEUES; default 4) and the number of prefix streams.  Round 5, one box: 4 queues: 19.49 ms with
# two prefix streams / 18.37 with one; 8 queues: 18.15 / 18.39; without the collective 17.9 either way.
run() { "$@" python bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | grep '^{"metric' | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print(r['ms_per_step'])"; }
FC="env PS_BENCH_FORCE_COLLECTIVE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533"
for q in 4 8 16; do for ps in 2 1; do echo -n "forced collective, $ps prefix stream(s), GPU_MAX_HW_QUEUES=$q: "; run $FC GPU_MAX_HW_QUEUES=$q PS_PREFIX_STREAMS=$ps; done; done
echo -n "forced collective, defaults: "; run $FC
echo -n "no collective, defaults: "; run env
