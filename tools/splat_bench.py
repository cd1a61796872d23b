"""Splat-only timing loop (BASELINE config C2: batch-32 256x256 reprojection + soft z-buffer splat), meant to be
run under rocprofv3 --kernel-trace --stats:  python tools/splat_bench.py [B] [iters]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DEBUG", "False")
import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
model = bench.build_model(dev)
d, _ = bench.make_inputs(1, B, dev)
pm = model.pts_transformer
call = lambda: pm.forward_justpts(d["img"], d["depth"], d["K"], d["Kinv"], d["P"], d["Pinv"], d["RT2"], d["RT2inv"])
for _ in range(3):
    call()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    call()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
print(f"B={B}: {dt * 1e3:.3f} ms/batch, {B / dt:.0f} frames/s, {1900544.0 * B / dt / 1e9:.1f} GB/s algorithmic")
