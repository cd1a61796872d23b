#!/bin/bash
# builds a tuning copy of the library with k_gemm_wg's stamps and runs tools/wg_trace.py on it (on the GPU box)
set -e
cd "$(dirname "$0")/.."
export PS_HIP_LIB=$PWD/gpurun_out/libps_wgtrace.so PS_OBJ_SUFFIX=_wgtrace PS_EXTRA_HIPCC_FLAGS="-DPS_WG_TRACE_BUILD -DPS_TUNING_BUILD $PS_TRACE_EXTRA"
mkdir -p gpurun_out
python -m pixelsynth_amd.build > /dev/null
python tools/wg_trace.py "$@"
