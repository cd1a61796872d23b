"""Refinement decoder through MIOpen: default (immediate-mode heuristics) against torch.backends.cudnn.benchmark = True
(MIOpen's find mode: every convolution shape is timed once over the applicable solvers and the fastest is kept).
usage: python tools/dec_bench_mode.py [views, default 16]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from pixelsynth_amd import synthetic as syn  # noqa: E402
from pixelsynth_amd.networks import Unet, get_decoder  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")


def build(make):
    mod = make()
    shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
    mod.load_state_dict({k: torch.from_numpy(v) for k, v in syn.fill_state_dict(shapes, 5).items()})
    return mod.to(dev).eval()


dec = build(lambda: get_decoder(syn.network_opts()))
unet = build(lambda: Unet(channels_in=3, channels_out=1, opt=syn.network_opts()))
x = torch.from_numpy(syn.image(1, V, 3, 256)).to(dev)
bg = torch.zeros(V, 256, 256, dtype=torch.bool, device=dev)
bg[:, :, 160:] = True
ref = {}
for mode in (False, True):
    torch.backends.cudnn.benchmark = mode
    with torch.no_grad():
        for name, fn in (("decoder", lambda: dec(x, bg)), ("unet", lambda: unet(x))):
            t0 = time.perf_counter()
            for _ in range(3):
                out = fn()
            torch.cuda.synchronize()
            warm = time.perf_counter() - t0
            t0 = time.perf_counter()
            n = 5
            for _ in range(n):
                out = fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            if not mode:
                ref[name] = out
            err = float((out - ref[name]).abs().max())
            print(f"benchmark={mode} {name}: {dt * 1e3:.2f} ms per {V} views (warm-up incl. search {warm:.1f} s), max |diff| vs default {err:.2e}", flush=True)
