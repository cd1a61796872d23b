#!/bin/bash
# The splat alone (config C2, B clouds per batch): kernel trace + stats, then FETCH_SIZE / WRITE_SIZE and the instruction-mix counters, one
# rocprofv3 --pmc pass per group (never beside a trace):  tools/splat_profile.sh <tag> [B]     -> gpurun_out/<tag>/splat_*.txt
tag=$1; B=${2:-32}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/splat_stats -o splat -- python tools/splat_bench.py $B 20 > $out/splat_bench.txt 2> $out/splat_stats.log
cp $(find $out/splat_stats -name "*kernel_stats.csv" | head -1) $out/splat_kernel_stats.csv
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY"; do
  name=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d $out/splat_pmc_$name -o pmc -- python tools/splat_bench.py $B 4 > /dev/null 2> $out/splat_pmc_$name.log
done
for f in $(find $out -path "*splat_pmc_*" -name "*counter_collection.csv"); do python tools/pmc_summary.py $f; done > $out/splat_pmc_summary.txt 2>&1
tail -1 $out/splat_bench.txt
head -12 $out/splat_kernel_stats.csv | cut -c1-160
grep -E "FETCH_SIZE|WRITE_SIZE" $out/splat_pmc_summary.txt | head -24
