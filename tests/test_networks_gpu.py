"""GPU: the dense-network mirrors (pixelsynth_amd/networks) on the MI355X against the reference's golden
outputs (tests/golden/networks.npz, see tests/test_networks_cpu.py).  fp32 through MIOpen: tolerance 1e-3 relative /
1e-4 absolute -- convolution algorithms differ from the CPU's in summation order over up to 4096 products."""
import os

import numpy as np
import pytest
import torch

from pixelsynth_amd import synthetic as syn
from pixelsynth_amd.networks import Unet, get_decoder

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _filled(mod, seed):
    shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
    mod.load_state_dict({k: torch.from_numpy(v) for k, v in syn.fill_state_dict(shapes, seed).items()}, strict=True)
    return mod.to(DEV).eval()


def test_unet_and_decoder_on_the_gpu_match_the_reference(golden_dir):
    fx = np.load(os.path.join(golden_dir, "networks.npz"))
    seed = int(fx["weight_seed"])
    unet = _filled(Unet(channels_in=3, channels_out=1, opt=syn.network_opts()), seed)
    dec = _filled(get_decoder(syn.network_opts()), seed)
    img = torch.from_numpy(syn.image(int(fx["image_seeds"][0]), 1, 3, 256)).to(DEV)
    x = torch.from_numpy(syn.image(int(fx["image_seeds"][1]), 1, 3, 256)).to(DEV)
    bgm = torch.from_numpy(syn.background_masks(256)["ragged"])[None].to(DEV)
    noise = [torch.from_numpy(n).to(DEV) for n in fx["noise"]]
    with torch.no_grad():
        depth = unet(img)
        refined = dec(x, bgm, noise=noise)
    np.testing.assert_allclose(depth.cpu().numpy()[:, :, ::2, ::2], fx["unet_out_sub"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(refined.cpu().numpy()[:, :, ::4, ::4], fx["decoder_out_sub"], rtol=1e-3, atol=1e-4)
    # a batch of views gives the same images as one at a time (per-sample noise affine, no cross-batch statistics)
    with torch.no_grad():
        both = dec(torch.cat([x, img]), torch.cat([bgm, ~bgm]), noise=[n.expand(2, -1) for n in noise])
    np.testing.assert_allclose(both[0].cpu().numpy(), refined[0].cpu().numpy(), rtol=1e-3, atol=1e-4)
