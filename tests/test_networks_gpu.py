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


def _f16x3(x_nchw, w, scale=None, shift=None, bias=None, res=None):
    """ps_conv3x3_f16x3_pack + ps_conv3x3_f16x3_nhwc through the C ABI; returns (y as NCHW view, overflow flag tensor).
    res: (B, H, W, Co) contiguous."""
    from pixelsynth_amd import _lib
    L, st = _lib.lib(), torch.cuda.current_stream().cuda_stream
    B, Ci, H, W = x_nchw.shape
    Co = w.shape[0]
    xl, wl = x_nchw.permute(0, 2, 3, 1).contiguous(), w.permute(0, 2, 3, 1).contiguous()
    packed = torch.empty(L.ps_conv3x3_f16x3_packed_bytes(Co, Ci), dtype=torch.uint8, device=DEV)
    assert packed.numel() == 9 * ((Co + 127) // 128 * 128) * Ci * 4
    _lib.check(L.ps_conv3x3_f16x3_pack(wl.data_ptr(), Co, Ci, packed.data_ptr(), st), "pack")
    y = torch.empty(B, H, W, Co, device=DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    _lib.check(L.ps_conv3x3_f16x3_nhwc(xl.data_ptr(), None if scale is None else scale.data_ptr(), None if shift is None else shift.data_ptr(),
                                       packed.data_ptr(), None if bias is None else bias.data_ptr(), None if res is None else res.data_ptr(),
                                       B, H, W, Ci, Co, y.data_ptr(), flag.data_ptr(), st), "conv")
    return y.permute(0, 3, 1, 2), flag


@pytest.mark.parametrize("B,H,W,Ci,Co,fuse", [(2, 32, 32, 64, 128, False), (2, 32, 48, 128, 128, True), (1, 16, 16, 256, 256, True),
                                               (3, 48, 16, 32, 128, False), (5, 16, 32, 96, 256, True), (2, 32, 32, 64, 64, True),
                                               (37, 16, 16, 32, 192, False)])
def test_conv3x3_on_the_fp16_pipe_against_an_fp64_convolution(B, H, W, Ci, Co, fuse):
    """csrc/conv_f16x3.hip: fp32 in and out, every product three fp16 MFMAs on split operands.  Against torch's convolution in
    fp64 (the definition; F.conv2d(relu(norm(x)), w, None, 1, 1), models/layers/blocks.py:41-47): the error stays within 3e-6 of the output's
    largest magnitude -- an fp32 convolution (MIOpen, same inputs) sits at 2-4e-7 -- and within ten times the fp32 convolution's own.
    Tiles on every border (one-tile images, 1 x 3 and 3 x 1 tile grids), more and fewer (tile, channel block) items than persistent
    workgroups and counts the eight XCDs do not divide, 64 output channels (a half-empty channel block), the norm + ReLU on the way in with per-sample (B, C) scale / shift."""
    g = torch.Generator().manual_seed(Ci + Co + H)
    x = (torch.randn(B, Ci, H, W, generator=g) * 1.5).to(DEV)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / (3 * Ci ** 0.5)).to(DEV)
    sc = (torch.rand(B, Ci, generator=g) + 0.5).to(DEV) if fuse else None
    sh = (torch.randn(B, Ci, generator=g) * 0.3).to(DEV) if fuse else None
    xa = torch.clamp_min(x * sc.view(B, Ci, 1, 1) - sh.view(B, Ci, 1, 1), 0) if fuse else x
    ref = torch.nn.functional.conv2d(xa.double(), w.double(), None, 1, 1)
    y32 = torch.nn.functional.conv2d(xa, w, None, 1, 1)
    y, flag = _f16x3(x, w, sc, sh)
    top = ref.abs().max().item()
    e16, e32 = (y.double() - ref).abs().max().item() / top, (y32.double() - ref).abs().max().item() / top
    assert int(flag.item()) == 0
    assert e16 < 3e-6 and e16 < 10 * e32, (e16, e32)


@pytest.mark.parametrize("B,H,W,C,Co,fuse", [(2, 16, 32, 64, 128, True), (1, 32, 16, 128, 64, True), (3, 16, 16, 32, 128, False)])
def test_conv3x3_on_the_fp16_pipe_reads_space_to_depth_blocks_in_place(B, H, W, C, Co, fuse):
    """ps_conv3x3_f16x3_ex_nhwc, in_s2d: x (B, 2 H, 2 W, C) read as its space-to-depth form (B, H, W, 4 C) -- a 4 x 4 stride-2 convolution as the
    VQ-VAE runs it (vqvae.s2d_weight) against torch's strided layer in fp64, ReLU on the way in."""
    from pixelsynth_amd import _lib
    from pixelsynth_amd.vqvae2.vqvae import s2d_weight
    L, st = _lib.lib(), torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(C + Co)
    x = torch.randn(B, C, 2 * H, 2 * W, generator=g).to(DEV)
    w = (torch.randn(Co, C, 4, 4, generator=g) / (4 * C ** 0.5)).to(DEV)
    bias = torch.randn(Co, generator=g).to(DEV)
    ref = torch.nn.functional.conv2d((torch.relu(x) if fuse else x).double(), w.double(), bias.double(), 2, 1)
    wl = s2d_weight(w).permute(0, 2, 3, 1).contiguous()
    packed = torch.empty(L.ps_conv3x3_f16x3_packed_bytes(Co, 4 * C), dtype=torch.uint8, device=DEV)
    _lib.check(L.ps_conv3x3_f16x3_pack(wl.data_ptr(), Co, 4 * C, packed.data_ptr(), st), "pack")
    xl = x.permute(0, 2, 3, 1).contiguous()
    sc, sh = (torch.ones(B, 4 * C, device=DEV), torch.zeros(B, 4 * C, device=DEV)) if fuse else (None, None)
    p = lambda t: None if t is None else t.data_ptr()
    y = torch.full((B, H, W, Co), float("nan"), device=DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    _lib.check(L.ps_conv3x3_f16x3_ex_nhwc(xl.data_ptr(), p(sc), p(sh), packed.data_ptr(), bias.data_ptr(), None, B, H, W, 4 * C, Co, 0, 1, 0,
                                          y.data_ptr(), flag.data_ptr(), st), "conv")
    err = (y.permute(0, 3, 1, 2).double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 3e-6 and int(flag.item()) == 0, err
    assert L.ps_conv3x3_f16x3_ex_nhwc(xl.data_ptr(), None, None, packed.data_ptr(), None, None, B, H, W, 64, Co, 0, 1, 0, y.data_ptr(),
                                      flag.data_ptr(), st) != 0                       # (Ci / 4 = 16: not a multiple of 32)


@pytest.mark.parametrize("Co,live", [(64, 32), (64, 0), (128, 96), (128, 40)])
def test_conv3x3_on_the_fp16_pipe_skips_channel_tiles_that_are_padding(Co, live):
    """co_live: the output channels from there on carry all-zero weights (the VQ-VAE's 128 -> 32 layers packed as 64; 64 of a 128-channel block
    by themselves) and their MFMAs are not issued.  Same results as without the hint -- the live channels against an fp64 convolution, the
    others exactly bias -- whichever tiles of a wave are live (both, the lower one, none)."""
    from pixelsynth_amd import _lib
    L, st = _lib.lib(), torch.cuda.current_stream().cuda_stream
    B, H, W, Ci = 3, 32, 16, 128
    n = live or Co
    g = torch.Generator().manual_seed(Co + live)
    x = torch.randn(B, Ci, H, W, generator=g).to(DEV)
    w = torch.zeros(Co, Ci, 3, 3)
    w[:n] = torch.randn(n, Ci, 3, 3, generator=g) / (3 * Ci ** 0.5)
    w = w.to(DEV)
    bias = torch.randn(Co, generator=g).to(DEV)
    ref = torch.nn.functional.conv2d(torch.relu(x).double(), w.double(), bias.double(), 1, 1)
    wl = w.permute(0, 2, 3, 1).contiguous()
    packed = torch.empty(L.ps_conv3x3_f16x3_packed_bytes(Co, Ci), dtype=torch.uint8, device=DEV)
    _lib.check(L.ps_conv3x3_f16x3_pack(wl.data_ptr(), Co, Ci, packed.data_ptr(), st), "pack")
    xl = x.permute(0, 2, 3, 1).contiguous()
    sc, sh = torch.ones(B, Ci, device=DEV), torch.zeros(B, Ci, device=DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    outs = []
    for hint in (live, 0):
        y = torch.full((B, H, W, Co), float("nan"), device=DEV)
        _lib.check(L.ps_conv3x3_f16x3_ex_nhwc(xl.data_ptr(), sc.data_ptr(), sh.data_ptr(), packed.data_ptr(), bias.data_ptr(), None, B, H, W, Ci, Co,
                                              hint, 0, 0, y.data_ptr(), flag.data_ptr(), st), "conv")
        outs.append(y)
    assert torch.equal(outs[0], outs[1])
    got = outs[0].permute(0, 3, 1, 2).double()
    assert (got - ref).abs().max().item() / ref.abs().max().item() < 3e-6
    assert torch.equal(outs[0][..., n:], bias[n:].expand(B, H, W, Co - n))
    assert L.ps_conv3x3_f16x3_ex_nhwc(xl.data_ptr(), None, None, packed.data_ptr(), None, None, B, H, W, Ci, Co, Co + 1, 0, 0, y.data_ptr(),
                                      flag.data_ptr(), st) != 0


@pytest.mark.parametrize("B,H,W,Ci,C,fuse", [(2, 16, 32, 64, 64, False), (1, 32, 32, 128, 64, True), (3, 16, 16, 32, 128, True)])
def test_conv3x3_on_the_fp16_pipe_stores_depth_to_space(B, H, W, Ci, C, fuse):
    """ps_conv3x3_f16x3_ex_nhwc, out_d2s: the (B, H, W, 4 C) result stored as (B, 2 H, 2 W, C) -- a 4 x 4 stride-2 TRANSPOSED convolution as the
    VQ-VAE runs it (vqvae.convt_weight) against torch's layer in fp64."""
    from pixelsynth_amd import _lib
    from pixelsynth_amd.vqvae2.vqvae import convt_weight
    L, st = _lib.lib(), torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(Ci + C)
    x = torch.randn(B, Ci, H, W, generator=g).to(DEV)
    wt = (torch.randn(Ci, C, 4, 4, generator=g) / (2 * Ci ** 0.5)).to(DEV)
    bias = torch.randn(C, generator=g).to(DEV)
    ref = torch.nn.functional.conv_transpose2d((torch.relu(x) if fuse else x).double(), wt.double(), bias.double(), 2, 1)
    wl = convt_weight(wt).permute(0, 2, 3, 1).contiguous()
    packed = torch.empty(L.ps_conv3x3_f16x3_packed_bytes(4 * C, Ci), dtype=torch.uint8, device=DEV)
    _lib.check(L.ps_conv3x3_f16x3_pack(wl.data_ptr(), 4 * C, Ci, packed.data_ptr(), st), "pack")
    xl = x.permute(0, 2, 3, 1).contiguous()
    sc, sh = (torch.ones(B, Ci, device=DEV), torch.zeros(B, Ci, device=DEV)) if fuse else (None, None)
    p = lambda t: None if t is None else t.data_ptr()
    y = torch.full((B, 2 * H, 2 * W, C), float("nan"), device=DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    b4 = bias.repeat(4).contiguous()
    _lib.check(L.ps_conv3x3_f16x3_ex_nhwc(xl.data_ptr(), p(sc), p(sh), packed.data_ptr(), b4.data_ptr(), None, B, H, W, Ci, 4 * C, 0, 0, 1,
                                          y.data_ptr(), flag.data_ptr(), st), "conv")
    err = (y.permute(0, 3, 1, 2).double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 3e-6 and int(flag.item()) == 0, err
    assert L.ps_conv3x3_f16x3_ex_nhwc(xl.data_ptr(), None, None, packed.data_ptr(), None, y.data_ptr(), B, H, W, Ci, 4 * C, 0, 0, 1, y.data_ptr(),
                                      flag.data_ptr(), st) != 0                       # (res does not go with depth-to-space)


@pytest.mark.parametrize("Co", [64, 128, 256])
def test_conv3x3_on_the_fp16_pipe_adds_bias_and_the_other_branch_on_the_way_out(Co):
    """bias (Co) and res (B, H, W, Co): y = conv + bias + res, what a ResNet_Block's second convolution hands on (blocks.py:61-73);
    each alone and both, several items per workgroup (B = 70: 280 tiles), against the plain result plus the torch ops."""
    g = torch.Generator().manual_seed(Co)
    B, H, Ci = 70, 32, 32
    x = torch.randn(B, Ci, H, H, generator=g).to(DEV)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) * 0.1).to(DEV)
    bias, res = torch.randn(Co, generator=g).to(DEV), torch.randn(B, H, H, Co, generator=g).to(DEV)
    plain, _ = _f16x3(x, w)
    for bb, rr in ((bias, None), (None, res), (bias, res)):
        got, _ = _f16x3(x, w, None, None, bb, rr)
        want = plain.clone()
        if bb is not None:
            want = want + bb.view(1, -1, 1, 1)
        if rr is not None:
            want = want + rr.permute(0, 3, 1, 2)
        assert (got - want).abs().max().item() < 1e-5


def test_conv3x3_on_the_fp16_pipe_flags_what_fp16_cannot_hold_and_rejects_what_it_does_not_take():
    from pixelsynth_amd import _lib
    x = torch.ones(1, 32, 16, 16, device=DEV)
    w = torch.full((128, 32, 3, 3), 0.01, device=DEV)
    y, flag = _f16x3(x, w)
    assert int(flag.item()) == 0 and abs(y[0, 0, 8, 8].item() - 2.88) < 1e-5 and abs(y[0, 0, 0, 0].item() - 1.28) < 1e-5
    x[0, 5, 3, 3] = 7.0e4                      # beyond fp16's largest finite value
    _, flag = _f16x3(x, w)
    assert int(flag.item()) == 1
    x[0, 5, 3, 3] = float("nan")
    _, flag = _f16x3(x, w)
    assert int(flag.item()) == 1
    L, st = _lib.lib(), torch.cuda.current_stream().cuda_stream
    buf = torch.empty(1 << 20, dtype=torch.uint8, device=DEV)
    for Co, Ci, H in ((96, 32, 16), (128, 24, 16), (128, 32, 24)):
        rc = (L.ps_conv3x3_f16x3_pack(buf.data_ptr(), Co, Ci, buf.data_ptr(), st) if H == 16 else
              L.ps_conv3x3_f16x3_nhwc(buf.data_ptr(), None, None, buf.data_ptr(), None, None, 1, H, H, Ci, Co, buf.data_ptr(), buf.data_ptr(), st))
        assert rc != 0 and b"multiple" in L.ps_last_error()


def test_decoder_with_split_fp16_convolutions_equals_the_fp32_decoder(monkeypatch):
    """The refinement decoder's wide 3 x 3 layers run through csrc/conv_f16x3.hip by default (norm + ReLU fused into the staging);
    PS_DECODER_CONV=fp32 / opt.decoder_conv = "fp32" sends everything through torch (MIOpen fp32).  Same weights, image and noise:
    the two images agree to 2e-5 (the output is a tanh in [-1, 1]); the default path really takes the kernel; an overflowing
    activation is reported by check_f16x3_overflow."""
    from pixelsynth_amd.networks import architectures as A
    dec = _filled(get_decoder(syn.network_opts()), 7)
    x = torch.from_numpy(syn.image(3, 2, 3, 256)).to(DEV)
    bgm = torch.from_numpy(syn.background_masks(256)["ragged"])[None].expand(2, -1, -1).to(DEV)
    noise = [torch.randn(2, 20, generator=torch.Generator().manual_seed(i)).to(DEV) for i in range(dec.n_noise())]
    calls = []
    real = A._f16x3_conv
    monkeypatch.setattr(A, "_f16x3_conv", lambda *a, **k: (calls.append(a[0].in_channels), real(*a, **k))[1])
    with torch.no_grad():
        got = dec(x, bgm, noise=noise)
        assert len(calls) == 13 and A.DECODER_CONV == "f16x3"        # every 3 x 3 layer with Ci % 32 == 0 and Co % 64 == 0 (4 -> 64 and 128 -> 3: conv_thin)
        A.check_f16x3_overflow(x.device)
        monkeypatch.setattr(A, "DECODER_CONV", "fp32")
        del calls[:]
        want = dec(x, bgm, noise=noise)
        assert not calls
    assert (got - want).abs().max().item() < 2e-5
    with torch.no_grad():
        monkeypatch.setattr(A, "DECODER_CONV", "f16x3")
        dec(x * 1e6, bgm, noise=noise)
    with pytest.raises(RuntimeError, match="fp16's range"):
        A.check_f16x3_overflow(x.device)
    A.check_f16x3_overflow(x.device)   # (the flag is cleared by the report)
    # a WEIGHT fp16 cannot hold: that layer goes through torch (decided once, when the weight is packed), the others stay
    conv = dec.eblocks[6].ch_a[5]
    with torch.no_grad():
        conv.weight_orig[0, 0, 0, 0] = 3.0e9      # (spectral norm divides by sigma ~ 3e9 * |u0 v0|: still a huge entry against the rest)
        conv.weight_u.zero_(); conv.weight_u[0] = 1e-6; conv.weight_v.zero_(); conv.weight_v[0] = 1e-6
        del calls[:]
        out = dec(x, bgm, noise=noise)
        A.check_f16x3_overflow(x.device)
        assert len(calls) == 14 and conv.__dict__["_ps_f16x3_cache"][1] is None and out.shape == want.shape   # (asked twice: with and without the branch sum)


def test_conv3x3_on_the_fp16_pipe_indexes_activations_beyond_4_gib():
    """C5's 128 views put 4 GiB through the decoder's 128-channel layers at 256 x 256.  MIOpen's fp32 NHWC kernels come back wrong
    there (32-bit indexing; tools/conv_f16x3_big_check.py: 8e-2 of the output's scale) -- which is why networks/architectures.py
    cuts the batch for what still goes through torch (next test); the split-fp16 kernel indexes frames with 64 bits: frames 0, 63,
    64 and 127 of a 128-view batch against an fp64 convolution of those frames alone."""
    g = torch.Generator().manual_seed(11)
    V, H, C = 128, 256, 128
    x = torch.empty(V, C, H, H, device=DEV).contiguous(memory_format=torch.channels_last)
    for v in range(0, V, 16):
        x[v:v + 16] = torch.randn(16, C, H, H, generator=g).to(DEV)
    w = (torch.randn(C, C, 3, 3, generator=g) * 0.03).to(DEV)
    y, flag = _f16x3(x, w)
    assert x.numel() * 4 == 2 ** 32 and int(flag.item()) == 0
    for v in (0, 63, 64, 127):
        ref = torch.nn.functional.conv2d(x[v:v + 1].double(), w.double(), None, 1, 1)
        err = (y[v:v + 1].double() - ref).abs().max().item() / ref.abs().max().item()
        assert err < 3e-6, (v, err)


def test_convolutions_through_torch_are_cut_below_2_gib_per_call(monkeypatch):
    """_conv2d_batches: the decoder's convolutions that still go through torch (MIOpen) see at most _MIOPEN_SAFE_BYTES of input or
    output per call.  With the limit lowered so that a 5-view batch is cut into calls of 2, 2 and 1 views: the same image as uncut (1e-5),
    channels_last preserved, and the calls are the expected ones."""
    from pixelsynth_amd.networks import architectures as A
    monkeypatch.setattr(A, "DECODER_CONV", "fp32")
    dec = _filled(get_decoder(syn.network_opts()), 9)
    x = torch.from_numpy(syn.image(5, 5, 3, 256)).to(DEV)
    bgm = torch.from_numpy(syn.background_masks(256)["ragged"])[None].expand(5, -1, -1).to(DEV)
    noise = [torch.randn(5, 20, generator=torch.Generator().manual_seed(i)).to(DEV) for i in range(dec.n_noise())]
    with torch.no_grad():
        want = dec(x, bgm, noise=noise)
        sizes = []
        real = A._conv2d_batches

        def spy(fn, t, *a, **k):
            return real(lambda u: (sizes.append(u.size(0)), fn(u))[1], t, *a, **k)
        monkeypatch.setattr(A, "_conv2d_batches", spy)
        monkeypatch.setattr(A, "_MIOPEN_SAFE_BYTES", 2 * 128 * 256 * 256 * 4 + 1)   # two views of the widest layer
        got = dec(x, bgm, noise=noise)
    assert 2 in sizes and 1 in sizes and max(sizes) <= 5 and sizes.count(5) > 0    # (narrow layers still go in one call)
    assert (got - want).abs().max().item() < 1e-5      # (MIOpen may pick another algorithm for another batch size: not bit-equal)
    t = torch.randn(5, 8, 16, 16, device=DEV).contiguous(memory_format=torch.channels_last)
    monkeypatch.setattr(A, "_MIOPEN_SAFE_BYTES", 2 * 8 * 16 * 16 * 4 + 1)
    out = real(lambda u: u * 2, t, 8)
    assert out.is_contiguous(memory_format=torch.channels_last) and torch.equal(out, t * 2)


@pytest.mark.parametrize("B,H,W,Ci,Co,fuse", [(2, 16, 64, 4, 64, True), (3, 8, 128, 4, 8, False), (2, 16, 64, 128, 3, True), (1, 24, 128, 32, 4, False),
                                               (5, 8, 32, 64, 1, True), (1, 8, 96, 256, 2, True)])
def test_thin_convolutions_against_an_fp64_convolution(B, H, W, Ci, Co, fuse):
    """csrc/conv_thin.hip: the decoder's 4 -> 64 and 128 -> 3 layers (and their neighbours in shape) as fp32 FMA passes, norm + ReLU
    on the way in, against torch's convolution in fp64: within 2e-6 of the output's largest magnitude (fp32 summation noise)."""
    from pixelsynth_amd import _lib
    L, st = _lib.lib(), torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(3 * Ci + Co)
    x = torch.randn(B, Ci, H, W, generator=g).to(DEV)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / (3 * Ci ** 0.5)).to(DEV)
    sc = (torch.rand(B, Ci, generator=g) + 0.5).to(DEV) if fuse else None
    sh = (torch.randn(B, Ci, generator=g) * 0.3).to(DEV) if fuse else None
    xa = torch.clamp_min(x * sc.view(B, Ci, 1, 1) - sh.view(B, Ci, 1, 1), 0) if fuse else x
    ref = torch.nn.functional.conv2d(xa.double(), w.double(), None, 1, 1)
    xl, wl = x.permute(0, 2, 3, 1).contiguous(), w.permute(2, 3, 1, 0).contiguous()
    y = torch.empty(B, H, W, Co, device=DEV)
    p = lambda t: None if t is None else t.data_ptr()
    if Ci == 4:
        _lib.check(L.ps_conv3x3_thin_in_nhwc_f32(xl.data_ptr(), p(sc), p(sh), wl.data_ptr(), B, H, W, Co, y.data_ptr(), st), "thin_in")
    else:
        _lib.check(L.ps_conv3x3_thin_out_nhwc_f32(xl.data_ptr(), p(sc), p(sh), wl.data_ptr(), B, H, W, Ci, Co, y.data_ptr(), st), "thin_out")
    err = (y.permute(0, 3, 1, 2).double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 2e-6, err
    assert L.ps_conv3x3_thin_out_nhwc_f32(xl.data_ptr(), None, None, wl.data_ptr(), B, H, W, 32, 5, y.data_ptr(), st) != 0
    assert L.ps_conv3x3_thin_in_nhwc_f32(xl.data_ptr(), None, None, wl.data_ptr(), B, 12, W, 8, y.data_ptr(), st) != 0


@pytest.mark.parametrize("B,H,W,Ci,Co", [(2, 16, 64, 4, 64), (3, 9, 31, 64, 128), (1, 64, 64, 128, 256), (2, 33, 17, 256, 128), (1, 128, 128, 128, 128),
                                          (2, 40, 24, 128, 3), (3, 7, 5, 32, 128), (1, 32, 32, 128, 64), (1, 3, 3, 64, 1), (2, 5, 7, 32, 20)])
def test_conv1x1_on_the_fp32_matrix_pipe_against_an_fp64_convolution(B, H, W, Ci, Co):
    """csrc/conv1x1.hip: the projection branches of the decoder's blocks and the VQ-VAE's 1 x 1 layers (and neighbours in shape: pixel counts
    that do not fill a workgroup's trip, output channels that do not fill a tile) against torch's convolution in fp64: within 2e-6 of the
    output's largest magnitude (exact fp32 products, fp32 accumulation); through the module-level wrapper the decoder uses as well."""
    from pixelsynth_amd import _lib
    from pixelsynth_amd.networks import architectures as A
    L, st = _lib.lib(), torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(5 * Ci + Co)
    x = torch.randn(B, Ci, H, W, generator=g).to(DEV)
    w = (torch.randn(Co, Ci, 1, 1, generator=g) / Ci ** 0.5).to(DEV)
    ref = torch.nn.functional.conv2d(x.double(), w.double())
    xl, wl = x.permute(0, 2, 3, 1).contiguous(), w.reshape(Co, Ci).contiguous()
    y = torch.full((B, H, W, Co), float("nan"), device=DEV)
    assert L.ps_conv1x1_takes(Ci, Co) == 1
    _lib.check(L.ps_conv1x1_nhwc_f32(xl.data_ptr(), wl.data_ptr(), B * H * W, Ci, Co, y.data_ptr(), st), "conv1x1")
    err = (y.permute(0, 3, 1, 2).double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 2e-6, err
    if Ci % 4 == 0:
        conv = torch.nn.Conv2d(Ci, Co, 1).to(DEV)
        with torch.no_grad():
            conv.weight.copy_(w)
            got = A.conv1x1(conv, x.contiguous(memory_format=torch.channels_last))
        assert got is not None and got.shape == (B, Co, H, W) and (got.double() - ref).abs().max().item() / ref.abs().max().item() < 2e-6
        with torch.no_grad():
            assert A.conv1x1(conv, x) is None                                  # (NCHW storage: not taken)
            assert A.conv1x1(torch.nn.Conv2d(Ci, Co, 1, stride=2).to(DEV), x.contiguous(memory_format=torch.channels_last)) is None
    assert L.ps_conv1x1_takes(48, 64) == 0 and L.ps_conv1x1_takes(256, 256) == 0 and L.ps_conv1x1_takes(128, 0) == 0
    assert L.ps_conv1x1_nhwc_f32(xl.data_ptr(), wl.data_ptr(), B * H * W, 48, Co, y.data_ptr(), st) != 0
    assert L.ps_conv1x1_nhwc_f32(xl.data_ptr(), wl.data_ptr(), 0, Ci, Co, y.data_ptr(), st) != 0


def test_noise_affine_in_one_launch_equals_the_composed_form():
    """ps_noise_affine_f32 (LinearNoiseLayer.affine_bc on the GPU): the (B, C) scale / shift of a noise-conditioned norm layer, a
    pending convolution bias folded in, against affine() composed from torch ops on the CPU (normalization.py:21-47, :170-184)."""
    from pixelsynth_amd.networks.architectures import LinearNoiseLayer
    g = torch.Generator().manual_seed(5)
    torch.manual_seed(5)     # (spectral_norm draws its u, v from the global generator: sigma, hence the size of the numbers, depends on it)
    for C, B in ((64, 3), (4, 1), (256, 16)):
        layer = LinearNoiseLayer(syn.network_opts(), output_sz=C).eval()
        with torch.no_grad():
            layer.bn.stored_mean.copy_(torch.randn(C, generator=g))
            layer.bn.stored_var.copy_(torch.rand(C, generator=g) + 0.2)
            for lin in (layer.gain, layer.bias):
                lin.weight_orig.copy_(torch.randn(C, 20, generator=g) * 0.3)
        noise, pend = torch.randn(B, 20, generator=g), torch.randn(C, generator=g)
        x = torch.zeros(B, C, 8, 8)
        with torch.no_grad():
            sc, sh = layer.affine(x, noise)
            want = sc.reshape(B, C), (sh - pend.view(1, -1, 1, 1) * sc).reshape(B, C)
            layer = layer.to(DEV)
            got = layer.affine_bc(x.to(DEV), noise.to(DEV), pend.to(DEV))
            plain = layer.affine_bc(x.to(DEV), noise.to(DEV))
        assert got[0].shape == (B, C) and got[0].is_contiguous()
        top = max(float(want[0].abs().max()), float(want[1].abs().max()), float(sh.abs().max()), 1.0)    # (shift is a difference of products of scale)
        np.testing.assert_allclose(got[0].cpu().numpy(), want[0].numpy(), rtol=1e-5, atol=1e-5 * top)
        np.testing.assert_allclose(got[1].cpu().numpy(), want[1].numpy(), rtol=1e-5, atol=2e-5 * top)
        np.testing.assert_allclose(plain[1].cpu().numpy(), sh.reshape(B, C).numpy(), rtol=1e-5, atol=2e-5 * top)


def test_decoder_input_is_packed_in_one_pass():
    """ps_cat_mask_nhwc_f32: torch.cat((x, ~background_mask), 1) (models/networks/architectures.py:153-156) of an NCHW image and a
    bool mask, written as the NHWC tensor the first block reads -- bit for bit the cat, in channels_last storage."""
    from pixelsynth_amd import _lib
    g = torch.Generator().manual_seed(2)
    for B, H, W in ((3, 8, 40), (1, 256, 256)):
        x = torch.randn(B, 3, H, W, generator=g).to(DEV)
        bg = (torch.rand(B, H, W, generator=g) > 0.4).to(DEV)
        out = torch.empty(B, H, W, 4, device=DEV)
        _lib.check(_lib.lib().ps_cat_mask_nhwc_f32(x.data_ptr(), bg.data_ptr(), B, H, W, out.data_ptr(), torch.cuda.current_stream().cuda_stream), "cat")
        want = torch.cat((x, (~bg).unsqueeze(1).float()), 1)
        assert torch.equal(out.permute(0, 3, 1, 2), want)


@pytest.mark.parametrize("B,H,W,fuse", [(2, 16, 64, True), (3, 9, 48, False), (1, 256, 256, True)])
def test_thin_in_on_the_fp16_pipe_against_an_fp64_convolution(B, H, W, fuse):
    """ps_conv3x3_thin_in_f16x3_nhwc: the decoder's 4 -> 64 layer as split-fp16 MFMAs (K = 36 in two steps of v_mfma_f32_16x16x32_f16,
    16 pixels of a row per tile) against torch's convolution in fp64: within 2e-6 of the output's largest magnitude, every border."""
    from pixelsynth_amd import _lib
    L, st = _lib.lib(), torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(H + W)
    x = torch.randn(B, 4, H, W, generator=g).to(DEV)
    w = (torch.randn(64, 4, 3, 3, generator=g) / 6).to(DEV)
    sc = (torch.rand(B, 4, generator=g) + 0.5).to(DEV) if fuse else None
    sh = (torch.randn(B, 4, generator=g) * 0.3).to(DEV) if fuse else None
    xa = torch.clamp_min(x * sc.view(B, 4, 1, 1) - sh.view(B, 4, 1, 1), 0) if fuse else x
    ref = torch.nn.functional.conv2d(xa.double(), w.double(), None, 1, 1)
    xl, wl = x.permute(0, 2, 3, 1).contiguous(), w.permute(2, 3, 1, 0).contiguous()
    y = torch.empty(B, H, W, 64, device=DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    p = lambda t: None if t is None else t.data_ptr()
    _lib.check(L.ps_conv3x3_thin_in_f16x3_nhwc(xl.data_ptr(), p(sc), p(sh), wl.data_ptr(), B, H, W, 64, y.data_ptr(), flag.data_ptr(), st), "thin_in_f16x3")
    err = (y.permute(0, 3, 1, 2).double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 2e-6 and int(flag.item()) == 0, err
    xl[0, 0, 0, 0] = 1e5
    _lib.check(L.ps_conv3x3_thin_in_f16x3_nhwc(xl.data_ptr(), None, None, wl.data_ptr(), B, H, W, 64, y.data_ptr(), flag.data_ptr(), st), "thin_in_f16x3")
    assert int(flag.item()) == 1
    assert L.ps_conv3x3_thin_in_f16x3_nhwc(xl.data_ptr(), None, None, wl.data_ptr(), B, H, W, 32, y.data_ptr(), flag.data_ptr(), st) != 0


def test_affine_bc_refuses_pointers_it_cannot_hand_to_the_kernel():
    """LinearNoiseLayer.affine_bc hands raw device pointers to ps_noise_affine_f32: a caller-supplied noise tensor that lives on the host
    (the documented noise= argument) must take the torch path -- where it fails with torch's device-mismatch error, as it always did --
    not reach the kernel as a host address (a GPU memory fault).  Device noise takes the kernel and equals the torch composition."""
    from pixelsynth_amd.networks import architectures as A
    opt = syn.network_opts()
    layer = A.LinearNoiseLayer(opt, output_sz=64).to(DEV).eval()
    x = torch.randn(3, 64, 8, 8, device=DEV).contiguous(memory_format=torch.channels_last)
    noise = torch.randn(3, A.NOISE_SZ)
    with torch.no_grad():
        with pytest.raises(RuntimeError):
            layer.affine_bc(x, noise)                       # host noise: torch's own error, no kernel launch
        torch.cuda.synchronize()                            # (the device is still alive)
        sc, sh = layer.affine_bc(x, noise.to(DEV))
        sc_ref, sh_ref = layer.affine(x, noise.to(DEV))
    torch.testing.assert_close(sc, sc_ref.reshape(3, 64), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(sh, sh_ref.reshape(3, 64).expand(3, 64), rtol=1e-5, atol=1e-6)
