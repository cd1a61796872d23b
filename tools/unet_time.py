"""The depth Unet on V source images (default 8), for a kernel profile.  usage: python tools/unet_time.py [V]"""
import sys, time
import torch
sys.path.insert(0, ".")
from pixelsynth_amd import synthetic as syn
from pixelsynth_amd.networks import Unet
dev = torch.device("cuda:0")
V = int(sys.argv[1]) if len(sys.argv) > 1 else 8
unet = Unet(channels_in=3, channels_out=1, opt=syn.network_opts())
shapes = {k: tuple(v.shape) for k, v in unet.state_dict().items()}
unet.load_state_dict({k: torch.from_numpy(v) for k, v in syn.fill_state_dict(shapes, 5).items()})
unet = unet.to(dev).eval()
x = torch.from_numpy(syn.image(1, V, 3, 256)).to(dev)
with torch.no_grad():
    for _ in range(3): unet(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): unet(x)
    torch.cuda.synchronize()
print(f"unet {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms per {V} images")
