#!/bin/bash
# instruction-mix counters of the splat-only loop (config C2): tools/splat_pmc.sh [B] [tag]
B=${1:-32}; tag=${2:-splat_pmc}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_RD"; do
  name=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d $out/pmc_$name -o pmc -- python tools/splat_bench.py $B 4 > /dev/null 2> $out/pmc_$name.log
done
for f in $(find $out -name "*counter_collection.csv"); do python tools/pmc_summary.py $f; done 2>&1 | grep -E "k_composite|k_sort_small"
