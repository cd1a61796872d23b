#!/bin/bash
# builds a tuning copy of the library (column_debug exists there only) and times the column launches with a role switched off
set -e
cd "$(dirname "$0")/.."
export PS_HIP_LIB=$PWD/gpurun_out/libps_tuning.so PS_OBJ_SUFFIX=_tuning PS_EXTRA_HIPCC_FLAGS="-DPS_TUNING_BUILD"
mkdir -p gpurun_out
python -m pixelsynth_amd.build > /dev/null
python tools/tp_time.py "$@"
