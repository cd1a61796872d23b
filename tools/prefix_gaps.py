"""Tuning aid (un-profiled): where the prefix phase of a pipelined step goes.  Events on the streams themselves around every
eng.ar_prefix call and around the column launches of bench.run_steps: columns end -> first prefix kernel (per stream), the two ranges'
passes, join -> first column launch.  usage: python tools/prefix_gaps.py [steps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda", 0)
model = bench.build_model(dev)
d, host = bench.make_inputs(0, 128, dev)
side = bench.side_stream()
bench.run_steps(model, d, 1, 4, side)
torch.cuda.synchronize()
eng = model.outpaint2.engine(32, 32, 256)
marks = []     # (kind, stream id, event)
ev = lambda: torch.cuda.Event(enable_timing=True)
real_prefix, real_cols = eng.ar_prefix, model._pipe_columns


def prefix(*a, **k):
    e0, e1 = ev(), ev()
    e0.record()
    r = real_prefix(*a, **k)
    e1.record()
    marks.append(("prefix", torch.cuda.current_stream().cuda_stream, e0, e1))
    return r


def cols(*a, **k):
    e0, e1 = ev(), ev()
    e0.record()
    real_cols(*a, **k)
    e1.record()
    marks.append(("cols", torch.cuda.current_stream().cuda_stream, e0, e1))


eng.ar_prefix = prefix
model._pipe_columns = cols
import time
t0 = time.perf_counter()
bench.run_steps(model, d, 1, steps, side)
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / steps * 1e3:.3f} ms per step")
# per step: [prefix side, prefix main, cols]
seq = [m for m in marks]
rows = []
i = 0
while i + 2 < len(seq):
    if seq[i][0] == "prefix" and seq[i + 1][0] == "prefix" and seq[i + 2][0] == "cols":
        rows.append((seq[i], seq[i + 1], seq[i + 2]))
        i += 3
    else:
        i += 1
out = []
for k in range(2, len(rows) - 1):
    (ps, pm, c), prev = rows[k], rows[k - 1][2]
    out.append([prev[3].elapsed_time(ps[2]), prev[3].elapsed_time(pm[2]), ps[2].elapsed_time(ps[3]), pm[2].elapsed_time(pm[3]),
                max(prev[3].elapsed_time(ps[3]), prev[3].elapsed_time(pm[3])), pm[3].elapsed_time(c[2]), c[2].elapsed_time(c[3]), prev[3].elapsed_time(c[2])])
a = np.array(out)
names = ["cols end -> side prefix starts", "cols end -> main prefix starts", "side range's pass", "main range's pass", "cols end -> both passes done",
         "main pass done -> first column launch", "column launches", "cols end -> next cols start (prefix phase)"]
for n, col in zip(names, a.T):
    print(f"{n:45s} mean {col.mean():7.3f} ms   min {col.min():7.3f}   max {col.max():7.3f}")
