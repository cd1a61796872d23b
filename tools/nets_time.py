"""Tuning aid: time the depth Unet and the refinement decoder on 16 views under different torch settings."""
import sys, time
import torch
sys.path.insert(0, ".")
from pixelsynth_amd import synthetic as syn
from pixelsynth_amd.networks import Unet, get_decoder

dev = torch.device("cuda:0")
def filled(mod):
    shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
    mod.load_state_dict({k: torch.from_numpy(v) for k, v in syn.fill_state_dict(shapes, 5).items()})
    return mod.to(dev).eval()
unet, dec = filled(Unet(channels_in=3, channels_out=1, opt=syn.network_opts())), filled(get_decoder(syn.network_opts()))
x = torch.from_numpy(syn.image(1, 16, 3, 256)).to(dev)
bg = torch.zeros(16, 256, 256, dtype=torch.bool, device=dev); bg[:, :, 160:] = True
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    for bench in (False, True):
        torch.backends.cudnn.benchmark = bench
        for cl in (False, True):
            xx = x.contiguous(memory_format=torch.channels_last) if cl else x
            if cl:
                unet.to(memory_format=torch.channels_last); dec.to(memory_format=torch.channels_last)
            else:
                unet.to(memory_format=torch.contiguous_format); dec.to(memory_format=torch.contiguous_format)
            print(f"benchmark={bench} channels_last={cl}: unet {t(lambda: unet(xx)):.2f} ms, decoder {t(lambda: dec(xx, bg)):.2f} ms", flush=True)
    # hipGraph capture of the Unet (launch-bound at the 1x1..8x8 levels)
    torch.backends.cudnn.benchmark = False
    unet.to(memory_format=torch.contiguous_format)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): unet(x)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = unet(x)
    print(f"unet graph replay {t(g.replay):.2f} ms")
