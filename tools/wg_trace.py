"""Tuning aid (needs a build with PS_EXTRA_HIPCC_FLAGS=-DPS_WG_TRACE_BUILD): shader-clock stamps of wave 0 of the first 32
workgroups of the last k_gemm_wg launch of each variant: set-up, first staging, every open tap."""
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
L = _lib.lib()
p = L.ps_pixelcnn_debug_cache(eng.handle, 6, 0)
out = bench.run_step(model, d, 1)
torch.cuda.synchronize()


class _Raw:
    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": "<i8", "data": (ptr, False), "version": 2}


st = torch.as_tensor(_Raw(p, (3, 32, 16)), device=device).cpu().numpy()
for v, name in enumerate(("conv_out <2,2,2>", "conv_in <1,2,2>", "dilated <1,2,1>")):
    print(name)
    for y in range(0, 32, 2):
        r = st[v, y]
        taps = [int(r[k] - r[k - 1]) for k in range(3, 14) if r[k] > r[2] and r[k] > r[k - 1]]
        last = max(r[2:14])
        print(f"  wg {y * 128:4d}: setup {int(r[1] - r[0]):6d}  first staging {int(r[2] - r[1]):6d}  taps {taps}  sum {sum(taps)}  to post {int(r[14] - last)}  "
              f"post {int(r[15] - r[14])}  total {int(r[15] - r[0])}")

p9 = L.ps_pixelcnn_debug_cache(eng.handle, 9, 0)
ch = torch.as_tensor(_Raw(p9, (3, 32, 4, 24)), device=device).cpu().numpy()
for v, name in enumerate(("conv_out", "conv_in", "dilated")):
    print(name, "-- inside the fourth open tap, per wave: chain j = (MFMA issue, wait for the next chain's weights); then the tap's end")
    for y in (4, 12, 20):
        for w in range(4):
            r = ch[v, y, w]
            if r[16] == 0: continue
            parts = [f"({int(r[3 * j + 1] - r[3 * j])},{int(r[3 * j + 2] - r[3 * j + 1])})" for j in range(5)] if r[0] else []
            gaps = [int(r[3 * (j + 1)] - r[3 * j + 2]) for j in range(4)] if r[0] else []
            print(f"  wg {y * 128:4d} wave {w}: " + " ".join(parts) + f"  between chains {gaps}  total {int(r[15] - r[0])}"
                  f" | stage_load {int(r[17] - r[16])} products {int(r[18] - r[17])} stage_store {int(r[19] - r[18])} barrier {int(r[20] - r[19])}")
p7 = L.ps_pixelcnn_debug_cache(eng.handle, 7, 0)
sp = torch.as_tensor(_Raw(p7, (3, 4096, 2)), device=device).cpu().numpy().view(np.uint64)
for v, name in enumerate(("conv_out <2,2,2>", "conv_in <1,2,2>", "dilated <1,2,1>")):
    t0, t1, hw = sp[v, :, 0].astype(np.int64), (sp[v, :, 1] >> np.uint64(16)).astype(np.int64), (sp[v, :, 1] & np.uint64(0xffff)).astype(np.int64)
    t0 = t0 & ((1 << 48) - 1)
    ok = (t0 > 0) & (t1 > t0) & (t1 - t0 < 10 ** 6)
    # keep the workgroups of the LAST launch: starts within 1 ms of the latest start
    if not ok.any():
        print(name, ": no spans"); continue
    ok &= t0 > t0[ok].max() - 100000
    life = (t1 - t0)[ok] / 100.0
    print(f"{name}: lives (us) percentiles 0/10/50/90/100: {np.percentile(life, [0, 10, 50, 90, 100]).round(1)}; "
          f"{(life < 3).sum()} of {ok.sum()} shorter than 3 us")
    xcd = (hw[ok] >> 0) & 0xffff
    a, b = t0[ok], t1[ok]
    base = a.min()
    span = (b.max() - base) / 100.0
    dur = (b - a) / 100.0
    ev = sorted([(x, 1) for x in a] + [(x, -1) for x in b])
    cur, area, last = 0, 0.0, base
    for x, d in ev:
        area += cur * (x - last)
        last, cur = x, cur + d
    edges = np.linspace(base, b.max(), 11)
    conc = []
    for lo, hi in zip(edges[:-1], edges[1:]):
        conc.append(((np.minimum(b, hi) - np.maximum(a, lo)).clip(0).sum() / (hi - lo)) / 256)
    print("   workgroups per CU in tenths of the launch:", " ".join(f"{c:.2f}" for c in conc))
    print(f"{name}: {ok.sum()} workgroups of the last launch, span {span:.1f} us, mean life {dur.mean():.1f} us (min {dur.min():.1f}, max {dur.max():.1f}), "
          f"mean concurrency {area / (b.max() - base):.0f} workgroups = {area / (b.max() - base) / 256:.2f} per CU; distinct hw ids {len(set(hw[ok]))}")
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/wg_span.npz", sp=sp, st=st)
