"""The refinement decoder on V views (default 16): PS_DECODER_CONV=f16x3 against fp32 (MIOpen).  usage: python tools/dec_time.py [V [mode ...]]"""
import sys, time
import torch
sys.path.insert(0, ".")
from pixelsynth_amd import synthetic as syn
from pixelsynth_amd.networks import get_decoder
from pixelsynth_amd.networks import architectures as A
dev = torch.device("cuda:0")
V = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dec = get_decoder(syn.network_opts())
shapes = {k: tuple(v.shape) for k, v in dec.state_dict().items()}
dec.load_state_dict({k: torch.from_numpy(v) for k, v in syn.fill_state_dict(shapes, 5).items()})
dec = dec.to(dev).eval()
x = torch.from_numpy(syn.image(1, V, 3, 256)).to(dev)
bg = torch.zeros(V, 256, 256, dtype=torch.bool, device=dev); bg[:, :, 160:] = True
noise = [torch.randn(V, 20, device=dev) for _ in range(dec.n_noise())]
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    out = {}
    for mode in (sys.argv[2:] or ("fp32", "f16x3")):
        A.DECODER_CONV = mode
        out[mode] = dec(x, bg, noise=noise)
        print(f"{mode}: {t(lambda: dec(x, bg, noise=noise)):.2f} ms per {V} views", flush=True)
    A.check_f16x3_overflow(dev)
    if len(out) == 2:
        print("max |difference| of the two images:", (out["fp32"] - out["f16x3"]).abs().max().item())
