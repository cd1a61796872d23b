"""Tuning aid: average k_column / k_column_tp launch of a whole AR run of V views (HIP events around every launch), as is and
with a role switched off (tuning value column_debug, which exists in -DPS_TUNING_BUILD builds only: 1 = chains do not wait for the
neighbour slots, 2 = no chains, 3 = no neighbour role and no waiting; the codes of such runs are INVALID).
usage: bash tools/tp_trace.sh is the model for the build; then
   PS_HIP_LIB=gpurun_out/libps_trace.so python tools/tp_time.py [views] [modes e.g. 0,2,3]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pixelsynth_amd import _lib  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 128
modes = [int(m) for m in (sys.argv[2] if len(sys.argv) > 2 else "0,2,3").split(",")]
device = torch.device("cuda", 0)
model = bench.build_model(device)
d, _ = bench.make_inputs(0, V, device)
out = bench.run_step(model, d, 1)
plan = out["plan"]
eng = model.outpaint2.engine(32, 32, V)
cols, wave_start = plan.waves
print(f"V={V}: {cols.shape[0]} columns in {len(wave_start) - 1} waves, first_step {plan.first_step}")
for mode in modes:
    eng.set_tuning(column_debug=mode)
    launches, total_ms, fpc = ctypes.c_int(0), ctypes.c_float(0.0), ctypes.c_double(0.0)
    res = []
    for _ in range(3):
        c32 = d["codes"].reshape(V, 1024).to(torch.int32).contiguous().clone()
        rc = _lib.lib().ps_pixelcnn_time_ar_run_waves(
            eng.handle, _lib.ptr(c32), _lib.ptr(plan.order_loc), _lib.ptr(plan.region), _lib.ptr(plan.mask_init),
            _lib.ptr(plan.mask_undilated), _lib.ptr(plan.mask_dilated), _lib.ptr(d["uniforms"]), 0.7, V, plan.first_step,
            _lib.ptr(cols), _lib.ptr(wave_start), len(wave_start) - 1, ctypes.cast(ctypes.byref(launches), ctypes.c_void_p),
            ctypes.cast(ctypes.byref(total_ms), ctypes.c_void_p), ctypes.cast(ctypes.byref(fpc), ctypes.c_void_p),
            _lib.current_stream())
        _lib.check(rc, "time")
        res.append(total_ms.value * 1e3 / max(1, launches.value))
    us = sorted(res)[1]
    tf = fpc.value * cols.shape[0] / launches.value / (us * 1e-6) / 1e12
    print(f"  debug {mode}: {launches.value} launches, {us:.1f} us each, {tf:.1f} TFLOP/s dense-equivalent ({tf / 157.3:.3f} of peak)")
    try:
        eng.check()
    except RuntimeError as e:
        print("   (status:", str(e)[:80], ")")
eng.set_tuning(column_debug=0)
