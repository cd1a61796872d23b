"""Tuning aid (build with PS_EXTRA_HIPCC_FLAGS=-DPS_LA_COUNT): how many look-ahead items of the latency form had to wait for `done`."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pixelsynth_amd import _lib  # noqa: E402
V = int(sys.argv[1]) if len(sys.argv) > 1 else 16
device = torch.device("cuda", 0)
model = bench.build_model(device)
d, _ = bench.make_inputs(0, V, device)
out = bench.run_step(model, d, 1)
eng = model.outpaint2.engine(32, 32, V)
_lib.lib().ps_pixelcnn_debug_cache.restype = __import__("ctypes").c_void_p
p = _lib.lib().ps_pixelcnn_debug_cache(eng.handle, 8, 0)
class _Raw:
    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": "<u4", "data": (ptr, False), "version": 2}
t = torch.as_tensor(_Raw(p, (33, 32)), device=device)
before = t.cpu().numpy().copy()
out = bench.run_step(model, d, 1)
torch.cuda.synchronize()
after = t.cpu().numpy()
print("published per stage:", (after[:, 0] - before[:, 0])[:24])
print("waits per stage:    ", (after[:, 1] - before[:, 1])[:24])
