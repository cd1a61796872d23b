"""VQ-VAE top level on V views (default 128): encode_codes and decode_code, for a kernel profile.  usage: python tools/vq_time.py [V]"""
import sys, time
import torch
sys.path.insert(0, ".")
from pixelsynth_amd import synthetic as syn
from pixelsynth_amd.vqvae2.vqvae import VQVAETop
dev = torch.device("cuda:0")
V = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 128
vq = VQVAETop()
vq.load_state_dict({k: torch.from_numpy(v) for k, v in syn.vqvae_state_dict(0).items()}, strict=True)
vq = vq.to(dev).eval()
x = torch.from_numpy(syn.image(1, V, 3, 256)).to(dev)
def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    if "--benchmark" in sys.argv:
        torch.backends.cudnn.benchmark = True
    if "--cl" in sys.argv:
        vq = vq.to(memory_format=torch.channels_last); x = x.contiguous(memory_format=torch.channels_last)
    codes = vq.encode_codes(x)
    print(f"encode_codes {t(lambda: vq.encode_codes(x)):.2f} ms, decode_code {t(lambda: vq.decode_code(codes)):.2f} ms per {V} views")
