"""Tuning aid: host-side cost of build_ar_plan (D2H mask, ps_ar_plan, uploads, wavefront schedule) for V views."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pixelsynth_amd import _lib  # noqa: E402
from pixelsynth_amd.lmconv.model import wavefronts  # noqa: E402
import ctypes  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 128
device = torch.device("cuda", 0)
model = bench.build_model(device)
d, _ = bench.make_inputs(0, V, device)
gen_fs, bgm = model.pts_transformer.forward_justpts(d["img"], d["depth"], d["K"], d["Kinv"], d["P"], d["Pinv"], d["RT2"], d["RT2inv"])
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    bg = bgm.to(torch.uint8).cpu().contiguous().numpy()
    t1 = time.perf_counter()
    B, S, _ = bg.shape
    L = 1024
    order_loc = np.empty((B, L), np.int32)
    region = np.empty((B, L), np.uint8)
    masks = [np.empty((B, 9, L), np.float32) for _ in range(3)]
    first = ctypes.c_int32(0)
    rc = _lib.lib().ps_ar_plan(_lib.ptr(bg), B, S, 32, _lib.ptr(order_loc), _lib.ptr(region), _lib.ptr(masks[0]),
                               _lib.ptr(masks[1]), _lib.ptr(masks[2]), ctypes.cast(ctypes.byref(first), ctypes.c_void_p))
    t2 = time.perf_counter()
    ups = [torch.from_numpy(a).to(device, non_blocking=True) for a in (order_loc, region, *masks)]
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    w = wavefronts(order_loc, 32, 32, int(first.value), device)
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print(f"V={V}: D2H {1e3 * (t1 - t0):.2f} ms, ps_ar_plan {1e3 * (t2 - t1):.2f} ms, uploads {1e3 * (t3 - t2):.2f} ms, wavefronts {1e3 * (t4 - t3):.2f} ms ({len(w[1]) - 1} waves); cpus {os.cpu_count()}")
