import sys, torch
sys.path.insert(0, ".")
from pixelsynth_amd import synthetic as syn
from pixelsynth_amd.networks import get_decoder
dev = torch.device("cuda:0")
dec = get_decoder(syn.network_opts())
shapes = {k: tuple(v.shape) for k, v in dec.state_dict().items()}
dec.load_state_dict({k: torch.from_numpy(v) for k, v in syn.fill_state_dict(shapes, 5).items()})
dec = dec.to(dev).eval().to(memory_format=torch.channels_last)
x = torch.from_numpy(syn.image(1, 16, 3, 256)).to(dev).contiguous(memory_format=torch.channels_last)
bg = torch.zeros(16, 256, 256, dtype=torch.bool, device=dev); bg[:, :, 160:] = True
with torch.no_grad():
    for _ in range(5): dec(x, bg)
torch.cuda.synchronize()
