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


def test_block_elementwise_kernels_against_torch():
    """csrc/nets.hip: the one-pass forms of a ResNet_Block's elementwise work (blocks.py:41-47, :61-73) against the torch
    ops they replace, on channels-last tensors; the torch ops run on the CPU (the definition), fp32, 1e-6."""
    from pixelsynth_amd import _lib
    from pixelsynth_amd.networks.architectures import ResNet_Block, _resample, _resample_sum
    g = torch.Generator().manual_seed(0)
    st = torch.cuda.current_stream().cuda_stream
    for B, C, H, W in ((2, 64, 16, 24), (3, 8, 10, 6), (1, 128, 2, 2)):
        a = torch.randn(B, C, H, W, generator=g)
        b = torch.randn(B, C, H, W, generator=g)
        ad, bd = (t.to(DEV).contiguous(memory_format=torch.channels_last) for t in (a, b))
        bias = torch.randn(C, generator=g)
        for kind in ("Down", "Up", None):
            want = _resample(kind, a) + _resample(kind, b)
            got = _resample_sum(kind, ad, bd)
            assert got.is_contiguous(memory_format=torch.channels_last) and got.shape == want.shape
            np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=1e-6, atol=1e-6, err_msg=f"{kind} {B, C, H, W}")
            # a convolution bias still missing from the inputs (zero padding of the pooling windows stays zero)
            want = _resample(kind, a + bias.view(1, -1, 1, 1)) + _resample(kind, b)
            got = _resample_sum(kind, ad, bd, bias.to(DEV))
            np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=1e-6, atol=2e-6, err_msg=f"{kind} + bias {B, C, H, W}")
        # single branch through the C ABI (b = NULL)
        out = torch.empty((B, C, 2 * H, 2 * W), device=DEV).contiguous(memory_format=torch.channels_last)
        _lib.check(_lib.lib().ps_upsample_add_nhwc_f32(ad.data_ptr(), None, None, B, H, W, C, out.data_ptr(), st), "upsample")
        np.testing.assert_allclose(out.cpu().numpy(), _resample("Up", a).numpy(), rtol=1e-6, atol=1e-6)
        out = torch.empty((B, C, H // 2, W // 2), device=DEV).contiguous(memory_format=torch.channels_last)
        _lib.check(_lib.lib().ps_pool_add_nhwc_f32(ad.data_ptr(), None, None, B, H, W, C, out.data_ptr(), st), "pool")
        np.testing.assert_allclose(out.cpu().numpy(), _resample("Down", a).numpy(), rtol=1e-6, atol=1e-6)
        # norm + ReLU, per-sample and shared (1, C) affine
        for rows in (B, 1):
            scale, shift = torch.randn(rows, C, 1, 1, generator=g), torch.randn(rows, C, 1, 1, generator=g)
            layer = type("L", (), {"affine": staticmethod(lambda x, n, s=scale, h=shift: (s.to(x.device), h.to(x.device)))})
            with torch.no_grad():
                got = ResNet_Block._noise_affine(layer, ad, None)
                want = ResNet_Block._noise_affine(layer, a, None)
                gotb = ResNet_Block._noise_affine(layer, ad, None, bias.to(DEV))
                wantb = ResNet_Block._noise_affine(layer, a + bias.view(1, -1, 1, 1), None)
            np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=1e-6, atol=1e-6)
            np.testing.assert_allclose(gotb.cpu().numpy(), wantb.numpy(), rtol=1e-5, atol=1e-5)
    # channel counts the kernels do not take (the 3-channel last block) fall back to torch
    a3 = torch.randn(1, 3, 4, 4, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    np.testing.assert_allclose(_resample_sum("Up", a3, a3).cpu().numpy(), (2 * _resample("Up", a3)).cpu().numpy(), rtol=1e-6)
    rc = _lib.lib().ps_pool_add_nhwc_f32(a3.data_ptr(), None, None, 1, 4, 4, 3, a3.data_ptr(), st)
    assert rc != 0 and b"multiple of 4" in _lib.lib().ps_last_error()
