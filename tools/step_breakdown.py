"""Tuning aid: wall-clock split of one bench step (splat / AR plan / AR run), synchronised between parts."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pixelsynth_amd.z_buffermodel import build_ar_plan  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 16
device = torch.device("cuda", 0)
model = bench.build_model(device)
d, _ = bench.make_inputs(0, V, device)
for _ in range(2):
    bench.run_step(model, d, 1)
torch.cuda.synchronize()
acc = [0.0, 0.0, 0.0]
N = 5
for _ in range(N):
    t0 = time.perf_counter()
    gen_fs, bg = model.pts_transformer.forward_justpts(d["img"], d["depth"], d["K"], d["Kinv"], d["P"], d["Pinv"], d["RT2"], d["RT2inv"])
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    plan = build_ar_plan(bg, 32)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    c32 = d["codes"].reshape(V, 1024).to(torch.int32).contiguous().clone()
    eng = model.outpaint2.engine(32, 32, V)
    eng.ar_run(c32, plan.order_loc, plan.region, plan.mask_init, plan.mask_undilated, plan.mask_dilated,
               temperature=0.7, uniforms=d["uniforms"], forced=None, first_step=plan.first_step,
               waves=None if os.environ.get("PS_NO_WAVES") else plan.waves)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    acc[0] += t1 - t0
    acc[1] += t2 - t1
    acc[2] += t3 - t2
print(f"V={V}: splat {acc[0] / N * 1e3:.3f} ms, plan {acc[1] / N * 1e3:.3f} ms, ar_run {acc[2] / N * 1e3:.3f} ms, first_step {plan.first_step}, waves {len(plan.waves[1]) - 1}, columns {plan.waves[0].shape[0]}")
