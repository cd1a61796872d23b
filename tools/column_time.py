"""Tuning aid: time of one k_column launch with `views` columns (one order position of every view:
ps_pixelcnn_time_column_step).
usage: python tools/column_time.py [views] [reps]
env: PS_CHAIN_TRACE=file (needs a -DPS_CHAIN_TRACE_BUILD build; read with tools/chain_trace.py),
     PS_COLUMN_DEBUG=1|2|3 (chains do not wait / no chains / no neighbour role; a -DPS_TUNING_BUILD build only), PS_COL_CAP,
     PS_CHAIN_XCDS, PS_NBR_GROUPS (tuning values, read when the handle is created)"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pixelsynth_amd import _lib  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
device = torch.device("cuda", 0)
model = bench.build_model(device)
d, _ = bench.make_inputs(0, V, device)
out = bench.run_step(model, d, V)
plan = out["plan"]
eng = model.outpaint2.engine(32, 32, V)
c32 = out["codes"].reshape(V, 1024).to(torch.int32).contiguous()
step = min(1023, plan.first_step + (1024 - plan.first_step) // 2)
launches = (ctypes.c_int * 2)()
total_ms = (ctypes.c_float * 2)()
rc = _lib.lib().ps_pixelcnn_time_column_step(
    eng.handle, _lib.ptr(c32), _lib.ptr(plan.order_loc), _lib.ptr(plan.mask_init), _lib.ptr(plan.mask_undilated),
    _lib.ptr(plan.mask_dilated), V, step, reps, ctypes.cast(launches, ctypes.c_void_p),
    ctypes.cast(total_ms, ctypes.c_void_p), None, None, _lib.current_stream())
_lib.check(rc, "ps_pixelcnn_time_column_step")
for name, i in (("k_nbr", 0), ("chain", 1)):
    if launches[i]:
        print(f"{name}: {total_ms[i] * 1e3 / launches[i]:.2f} us x {launches[i]}")
