#!/bin/bash
# The refinement decoder's profile (run on the GPU box through gpurun, from the repository root):
#   tools/collect_decoder_profiles.sh <tag>
# kernel trace + stats of tools/dec_time.py 16 f16x3 (14 decoder passes), then one PMC run per counter group over
# tools/conv_f16x3_time.py (13 launches each of four of the decoder's layers) -- counters never share a run with a trace.
set -u
tag=$1
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o dec -- python tools/dec_time.py 16 f16x3 > $out/dec_time.txt 2> $out/stats.log
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/decoder_kernel_stats.csv
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  name=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d $out/pmc_$name -o pmc -- python tools/conv_f16x3_time.py 16 > /dev/null 2> $out/pmc_$name.log
done
for f in $(find $out -name "*counter_collection.csv"); do python tools/pmc_summary.py $f; done > $out/conv_pmc_summary.txt 2>&1
python tools/kernel_stats_top.py $out/decoder_kernel_stats.csv 14
grep -E "k_conv3x3|^gpurun" $out/conv_pmc_summary.txt | head -40
