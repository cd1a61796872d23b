"""Tuning aid (needs a build with PS_EXTRA_HIPCC_FLAGS=-DPS_TP_TRACE_BUILD): shader-clock stamps of chain_role_tp2, tile 0,
wave PS_COLUMN_DEBUG >> 8, per stage: control + counter wait, operand requests, products (until y is there), to barrier 1,
barrier 1, statistics + barrier 2, finish, barrier 3 (= until the next stage's start)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pixelsynth_amd import _lib  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 128
device = torch.device("cuda", 0)
model = bench.build_model(device)
d, _ = bench.make_inputs(0, V, device)
out = bench.run_step(model, d, 1)
eng = model.outpaint2.engine(32, 32, V)
p = _lib.lib().ps_pixelcnn_debug_cache(eng.handle, 4, 0)
out = bench.run_step(model, d, 1)
torch.cuda.synchronize()


class _Raw:
    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": "<i8", "data": (ptr, False), "version": 2}


st = torch.as_tensor(_Raw(p, (33, 8)), device=device).cpu().numpy()
names = ["ctl+wait", "ops issue", "products", "stat 1", "barrier 1", "stat 2+bar", "finish"]
tot = np.zeros(8)
print("stage  " + "  ".join(f"{n:>10s}" for n in names) + "   barrier 3   total")
for s in range(32):
    dl = [int(st[s, k + 1] - st[s, k]) for k in range(7)]
    nxt = int(st[s + 1, 0] - st[s, 7]) if s < 31 else 0
    tot += np.array(dl + [nxt])
    print(f"{s:5d}  " + "  ".join(f"{v:10d}" for v in dl) + f"   {nxt:9d}   {sum(dl) + nxt}")
print("sum    " + "  ".join(f"{int(v):10d}" for v in tot[:7]) + f"   {int(tot[7]):9d}   {int(tot.sum())}")
