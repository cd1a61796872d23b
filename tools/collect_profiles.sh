#!/bin/bash
# Collect the rocprofv3 summaries kept under profiles/ (run on the GPU box through gpurun, from the repository root):
#   tools/collect_profiles.sh <tag> [bench args...]      e.g.  tools/collect_profiles.sh r02_a --views 128
# kernel trace + stats in one run, then one run per PMC group (counters never share a run with a trace).
set -u
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# (PS_BENCH_PMC_CHILD=1: nothing but the timed steps in the profiled command -- bench.py then skips the roofline's own measurement runs,
# whose two-batch launches of twice the frames used to be averaged into every per-kernel figure)
export PS_BENCH_PMC_CHILD=1
B="python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-extra $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- $B > $out/bench_under_trace.json 2> $out/stats.log
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  name=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d $out/pmc_$name -o pmc -- $B > /dev/null 2> $out/pmc_$name.log
done
find $out -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
for f in $(find $out -name "*counter_collection.csv"); do python tools/pmc_summary.py $f; done > $out/pmc_summary.txt 2>&1
unset PS_BENCH_PMC_CHILD
ls -la $out | head -30
head -12 $out/kernel_stats.csv
cat $out/pmc_summary.txt | head -80
