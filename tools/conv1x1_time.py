"""The decoder's / VQ-VAE's 1 x 1 convolutions at V views (default 128): csrc/conv1x1.hip against torch (MIOpen), ms per layer.
usage: python tools/conv1x1_time.py [V]"""
import sys, time
import torch
sys.path.insert(0, ".")
from pixelsynth_amd.networks import architectures as A
dev = torch.device("cuda:0")
V = int(sys.argv[1]) if len(sys.argv) > 1 else 128
LAYERS = [("dec b0 4->64 @256", 4, 64, 256), ("dec b1 64->128 @128", 64, 128, 128), ("dec b2 128->256 @64", 128, 256, 64),
          ("dec b4 256->128 @64", 256, 128, 64), ("dec b5 128->128 @128", 128, 128, 128), ("dec b7 128->3 @256", 128, 3, 256),
          ("vq res 32->128 @64", 32, 128, 64), ("vq res 32->128 @32", 32, 128, 32), ("vq quantize 128->64 @32", 128, 64, 32)]
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
tot = [0.0, 0.0]
with torch.no_grad():
    for name, Ci, Co, S in LAYERS:
        conv = torch.nn.Conv2d(Ci, Co, 1).to(dev).to(memory_format=torch.channels_last)
        x = torch.randn(V, Ci, S, S, device=dev).contiguous(memory_format=torch.channels_last)
        assert A.conv1x1(conv, x) is not None
        a = t(lambda: A.conv1x1(conv, x))
        b = t(lambda: A._conv2d_batches(lambda u: torch.nn.functional.conv2d(u, conv.weight), x, Co))
        gb = V * S * S * (Ci + Co) * 4 / 1e9
        gf = V * S * S * Ci * Co * 2 / 1e9
        print(f"{name:26s} hip {a:7.3f} ms ({gb / a:6.0f} GB/s, {gf / a:6.1f} TFLOP/s)   torch {b:7.3f} ms", flush=True)
        if name.startswith("dec"):
            tot[0] += a; tot[1] += b
        del x
print(f"decoder's six layers: hip {tot[0]:.2f} ms, torch {tot[1]:.2f} ms per {V} views")
