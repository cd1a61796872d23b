"""Tuning probe: does a whole-grid pass on a second stream / second engine overlap with the column launches of an AR run?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pixelsynth_amd.z_buffermodel import build_ar_plan

V = 16
dev = torch.device("cuda:0")
mA, mB = bench.build_model(dev), bench.build_model(dev)
d, _ = bench.make_inputs(0, V, dev)
gen_fs, bg = mA.pts_transformer.forward_justpts(d["img"], d["depth"], d["K"], d["Kinv"], d["P"], d["Pinv"], d["RT2"], d["RT2inv"])
plan = build_ar_plan(bg, 32)
engA, engB = mA.outpaint2.engine(32, 32, V), mB.outpaint2.engine(32, 32, V)
codes = d["codes"].reshape(V, 1024).to(torch.int32).contiguous()
sB = torch.cuda.Stream()

def ar():
    c = codes.clone()
    engA.ar_run(c, plan.order_loc, plan.region, plan.mask_init, plan.mask_undilated, plan.mask_dilated, temperature=0.7,
                uniforms=d["uniforms"], first_step=plan.first_step, waves=plan.waves)

def grid():   # the prefix pass of another step: same kernels as a whole-grid forward restricted to the prefix
    c = codes.clone()
    engB.ar_run(c, plan.order_loc, plan.region * 0, plan.mask_init, plan.mask_undilated, plan.mask_dilated, temperature=0.7,
                uniforms=d["uniforms"], first_step=plan.first_step, waves=None) if False else engB.forward(c, plan.mask_init, plan.mask_undilated, plan.mask_dilated)

def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

print("AR alone %.3f ms" % timed(ar))
print("grid(full forward) alone %.3f ms" % timed(grid))
def both():
    with torch.cuda.stream(sB):
        grid()
    ar()
print("AR + grid on a second stream %.3f ms" % timed(both))
for delay in (1.5, 2.5, 3.5):
    def late():
        ar()
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e3 < delay:
            pass
        with torch.cuda.stream(sB):
            grid()
    print("AR, then %.1f ms later a grid on a second stream: %.3f ms" % (delay, timed(late)))
engA.check(); engB.check()
