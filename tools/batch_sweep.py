"""Tuning / documentation aid: frames/s of the pipelined step (bench.run_steps) against the number of views per step -- C5-shaped
batches of ceil(V / 16) sources x 16 views.  The column launches are bound by the latency of their 33 dependent stages, so a step's
time grows much more slowly than its batch.   usage: python tools/batch_sweep.py [V ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

Vs = [int(v) for v in sys.argv[1:]] or [16, 32, 64, 128, 192, 256]
dev = torch.device("cuda", 0)
model = bench.build_model(dev)
side = torch.cuda.Stream()
for V in Vs:
    d, host = bench.make_inputs(0, V, dev)
    bench.run_steps(model, d, 1, 3, side)
    torch.cuda.synchronize()
    n = max(5, min(20, 2560 // V))
    t0 = time.perf_counter()
    out = bench.run_steps(model, d, 1, n, side)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / n
    model.outpaint2.engine(32, 32, V).check()
    r = bench.measure_roofline(model, d, out, V)
    print(f"V={V:4d}: {ms:7.3f} ms per step, {V / ms * 1e3:7.0f} frames/s; column launches: {r['launches_per_ar_run']} x {r['avg_launch_us']:.1f} us, "
          f"{r['columns_per_launch']:.0f} columns each, frac {r['frac']:.3f}", flush=True)
    del d
    torch.cuda.empty_cache()
