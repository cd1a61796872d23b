"""Tuning aid (needs a build with PS_EXTRA_HIPCC_FLAGS=-DPS_TP_TRACE_BUILD): shader-clock stamps of the chain role of
k_column_tp, tile 0 / wave 0, per stage: wait for the neighbour counter, operand issue, MFMA phase, barrier, post op, barrier."""
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
if len(sys.argv) > 2:   # column_debug of the tuning build (roles / memory operations / phases switched off: results invalid)
    eng.set_tuning(column_debug=int(sys.argv[2]))
p = _lib.lib().ps_pixelcnn_debug_cache(eng.handle, 4, 0)
out = bench.run_step(model, d, 1)
torch.cuda.synchronize()


class _Raw:
    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": "<i8", "data": (ptr, False), "version": 2}


st = torch.as_tensor(_Raw(p, (33, 8)), device=device).cpu().numpy()
names = ["ctl+wait", "ops issue", "B reads", "half 1", "half 2", "acc->LDS", "barrier"]
tot = np.zeros(7)
print("stage  " + "  ".join(f"{n:>13s}" for n in names) + "   total (cycles of the 100 MHz? shader clock)")
for s in range(32):
    dl = [st[s, k + 1] - st[s, k] for k in range(7)]
    tot += dl
    nxt = (st[s + 1, 0] - st[s, 7]) if s < 31 else 0
    print(f"{s:5d}  " + "  ".join(f"{v:13d}" for v in dl) + f"   post+barrier {nxt:6d}   total {st[s, 7] - st[s, 0] + nxt}")
print("sum    " + "  ".join(f"{int(v):13d}" for v in tot) + f"   {int(tot.sum())}   span {st[31, 7] - st[0, 0]}")
