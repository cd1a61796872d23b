"""GPU: VQ kernels (ps_vq_nearest_f32 / ps_vq_embed_f32 through the C ABI) and the VQVAETop mirror against the
oracle and the reference's golden outputs (tests/golden/vqvae.npz).

Index parity: the nearest-code index is an arg-min over float32 distances whose summation order the reference
leaves to its BLAS, so it is exact wherever the two smallest distances differ by more than the float noise
(TIE = 1e-4 relative); at the few near-ties either candidate is accepted (and checked to be one of the two)."""
import os

import numpy as np
import pytest
import torch

from oracle import vqvae_oracle as vo
from pixelsynth_amd import _lib, synthetic as syn

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "vqvae.npz"))
TIE = 1e-4


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def nearest(z, layout, embed, hw=1, want_dist=False):
    n = z.numel() // embed.shape[0]
    idx = torch.empty(n, dtype=torch.int32, device="cuda")
    md = torch.empty(n, dtype=torch.float32, device="cuda") if want_dist else None
    rc = _lib.lib().ps_vq_nearest_f32(_lib.ptr(z), layout, _lib.ptr(embed), n, embed.shape[0], embed.shape[1], hw, _lib.ptr(idx),
                                      _lib.ptr(md) if want_dist else None, _lib.current_stream())
    _lib.check(rc, "ps_vq_nearest_f32")
    return (idx, md) if want_dist else idx


def check_indices(got, z, embed):
    """got (N,) vs the float64 distances of the same inputs: exact except at near-ties."""
    d = ((z.astype(np.float64)[:, :, None] - embed.astype(np.float64)[None]) ** 2).sum(1)
    order = np.argsort(d, 1, kind="stable")
    best, second = order[:, 0], order[:, 1]
    rows = np.arange(len(d))
    gap = (d[rows, second] - d[rows, best]) / np.maximum(d[rows, best], 1e-12)
    ok = got == best
    assert np.all(ok | (gap < TIE)), f"{(~ok & (gap >= TIE)).sum()} wrong indices away from ties"
    near = ~ok
    assert np.all(d[rows[near], got[near]] <= d[rows[near], best[near]] * (1 + TIE) + 1e-12)
    return int(near.sum())


@pytest.mark.parametrize("N,D,K", [(1, 64, 512), (16, 64, 512), (1000, 64, 512), (37, 8, 5), (300, 33, 700)])
def test_nearest_random(N, D, K):
    rs = np.random.RandomState(N + D + K)
    z, emb = rs.randn(N, D).astype(np.float32), rs.randn(D, K).astype(np.float32)
    idx, md = nearest(dev(z), 0, dev(emb), want_dist=True)
    got = idx.cpu().numpy()
    assert got.min() >= 0 and got.max() < K
    check_indices(got, z, emb)
    d = ((z.astype(np.float64)[:, :, None] - emb.astype(np.float64)[None]) ** 2).sum(1)
    np.testing.assert_allclose(md.cpu().numpy(), d[np.arange(N), got], rtol=1e-4, atol=1e-4)


def test_nearest_exact_ties_take_the_smallest_index():
    z = np.zeros((4, 8), np.float32)
    emb = np.ones((8, 6), np.float32)
    emb[:, 3] = 0.5
    emb[:, 5] = 0.5            # columns 3 and 5 are identical minimisers
    assert nearest(dev(z), 0, dev(emb)).cpu().tolist() == [3, 3, 3, 3]


def test_nearest_nchw_layout_equals_flat_layout():
    lat = G["latB"]                                              # (1,64,32,32)
    emb = dev(G["embed"])
    flat = np.ascontiguousarray(lat.transpose(0, 2, 3, 1).reshape(-1, 64))
    a = nearest(dev(flat), 0, emb).cpu().numpy()
    b = nearest(dev(lat), 1, emb, hw=1024).cpu().numpy()
    assert np.array_equal(a, b)
    near = check_indices(a, flat, G["embed"])
    ref = G["codes"].reshape(-1)
    two = G["two_smallest"]
    gap = (two[:, 1] - two[:, 0]) / np.maximum(np.abs(two[:, 0]), 1e-12)
    assert np.all((a == ref) | (gap < TIE)) and near <= 8          # the reference's codes, up to its own near-ties


def test_embed_gather():
    emb = G["embed"]
    codes = G["codes"].copy()
    codes[0, 0, 0], codes[0, 0, 1] = -1, 512                      # out of range -> zeros
    out = torch.empty(1, 64, 32, 32, device="cuda")
    d_codes, d_emb = dev(codes), dev(emb)                          # (kept alive across the asynchronous launch)
    rc = _lib.lib().ps_vq_embed_f32(_lib.ptr(d_codes), _lib.ptr(d_emb), 1, 1024, 64, 512, _lib.ptr(out), _lib.current_stream())
    _lib.check(rc, "ps_vq_embed_f32")
    want = emb[:, np.clip(codes.reshape(-1), 0, 511)].reshape(64, 32, 32)
    want[:, 0, 0] = 0
    want[:, 0, 1] = 0
    assert np.array_equal(out.cpu().numpy()[0], want)


def test_errors():
    L = _lib.lib()
    z = dev(np.zeros((4, 65), np.float32))
    assert L.ps_vq_nearest_f32(_lib.ptr(z), 0, _lib.ptr(z), 4, 65, 4, 1, _lib.ptr(z), None, None) < 0   # D > 64
    assert L.ps_vq_nearest_f32(None, 0, _lib.ptr(z), 4, 8, 4, 1, _lib.ptr(z), None, None) < 0


def test_module_encode_decode_against_reference_outputs():
    from pixelsynth_amd.vqvae2 import VQVAETop
    sd = {k: torch.from_numpy(v) for k, v in syn.vqvae_state_dict(0).items()}
    sd["quantize_t.embed"] = torch.from_numpy(G["embed"])
    m = VQVAETop().eval()
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    img = dev(syn.image(int(G["image_seeds"][1]), 1, 3, 256))
    # The dense convolutions around the quantiser go through MIOpen, which picks its algorithm per shape from what it has
    # benchmarked in this process so far (run alone this test never failed, inside the whole suite about one run in six did):
    # the latent in front of the arg-min moves by a few 1e-4 relative with the algorithm, so the near-tie margin here is wider
    # than for the quantiser kernel alone (tests above: TIE), and the two encode surfaces may disagree at such a near-tie.
    margin = 10 * TIE
    codes = m.encode_codes(img)
    assert codes.dtype == torch.int32 and codes.is_cuda and tuple(codes.shape) == (1, 32, 32)
    got, ref = codes.cpu().numpy().reshape(-1), G["codes"].reshape(-1)
    two = G["two_smallest"]
    gap = (two[:, 1] - two[:, 0]) / np.maximum(np.abs(two[:, 0]), 1e-12)
    assert np.all((got == ref) | (gap < margin)) and (got != ref).sum() <= 16
    with torch.no_grad():
        full = m.encode(img)                                       # the reference-shaped surface agrees with the fast path
    other = full[3].to(torch.int32).cpu().numpy().reshape(-1)
    assert np.all((other == got) | (gap < margin)) and (other != got).sum() <= 16
    dec = m.decode_code(dev(G["codes"]))                           # the reference's codes -> the reference's image
    np.testing.assert_allclose(dec.cpu().numpy()[:, :, ::4, ::4], G["dec_sub"], rtol=1e-3, atol=1e-4)
    sd_cpu = {k: v for k, v in sd.items()}
    with torch.no_grad():
        want = vo.decode_code(sd_cpu, torch.from_numpy(G["codes"]).long())
    np.testing.assert_allclose(dec.cpu().numpy(), want.numpy(), rtol=1e-3, atol=1e-4)


def _module(seed=0):
    from pixelsynth_amd.vqvae2 import VQVAETop
    m = VQVAETop().eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in syn.vqvae_state_dict(seed).items()}, strict=True)
    return m.cuda()


def test_hand_written_convolutions_against_the_fp32_module_and_an_fp64_one(monkeypatch):
    """The inference GPU path of encode_codes / decode_code (vqvae._FastPath: the 3 x 3, 4 x 4 stride-2 and transposed convolutions through
    csrc/conv_f16x3.hip, the ResBlock tails through csrc/conv1x1.hip) against the same module in fp64 on the host: the latent in front of
    the arg-min within 3e-5 of its scale -- closer than the MIOpen fp32 path gets --, the codes equal except at near-ties, the decoded image within
    1e-5 of the fp64 one's scale; PS_VQVAE_CONV=fp32 and a decoder_conv("fp32") block take the torch path (no split-fp16 launch)."""
    from pixelsynth_amd.networks import architectures as A
    from pixelsynth_amd.vqvae2 import vqvae as V
    m = _module(1)
    img = dev(syn.image(11, 3, 3, 256))
    with torch.no_grad():
        fast = m._fast(img, 256, 256)
        assert fast is not None
        lat = fast.encode_latent(m, img)                                   # (B, 32, 32, 64)
        m64 = _module(1).cpu().double()
        ref = m64.quantize_conv_t(m64.enc_t(m64.enc_b(img.cpu().double()))).permute(0, 2, 3, 1)
        scale = ref.abs().max().item()
        err = (lat.cpu().double() - ref).abs().max().item() / scale
        assert err < 3e-5, err
        codes = m.encode_codes(img)
        emb = m64.quantize_t.embed
        z = ref.reshape(-1, emb.shape[0])
        d = (z * z).sum(1, keepdim=True) - 2 * z @ emb + (emb * emb).sum(0, keepdim=True)
        two = torch.topk(d, 2, dim=1, largest=False)
        gap = ((two.values[:, 1] - two.values[:, 0]) / two.values[:, 0].abs().clamp_min(1e-12)).numpy()
        got = codes.cpu().numpy().reshape(-1)
        assert np.all((got == two.indices[:, 0].numpy()) | (gap < 10 * TIE)) and (got != two.indices[:, 0].numpy()).sum() <= 8
        image = m.decode_code(codes)
        want = m64.decode(m64.quantize_t.embed_code(codes.cpu().long()).permute(0, 3, 1, 2))
        assert image.shape == (3, 3, 256, 256) and image.is_contiguous()
        assert (image.cpu().double() - want).abs().max().item() / want.abs().max().item() < 1e-5
        A.check_f16x3_overflow(img.device)
        # the switches
        calls = []
        real = V._FastPath.conv3
        monkeypatch.setattr(V._FastPath, "conv3", lambda self, *a, **k: (calls.append(1), real(self, *a, **k))[1])
        m.encode_codes(img)
        assert calls
        del calls[:]
        with A.decoder_conv("fp32"):
            slow = m.encode_codes(img)
            m.decode_code(slow)
        monkeypatch.setattr(V, "VQVAE_CONV", "fp32")
        m.decode_code(m.encode_codes(img))
        assert not calls
        assert (slow.cpu().numpy().reshape(-1) != got).sum() <= 8


@pytest.mark.parametrize("B,H,W", [(2, 64, 64), (1, 12, 48), (3, 256, 256)])
def test_three_channel_ends_against_fp64_layers(B, H, W):
    """csrc/vq_ends.hip: the VQ-VAE's first convolution (3 -> 64, 4 x 4, stride 2; output in the space-to-depth blocks the next layer reads)
    and its last transposed convolution (ReLU, 64 -> 3, 4 x 4, stride 2; the NCHW image) on the fp32 matrix pipe against torch's layers
    in fp64: 2e-6 of the output's scale; borders (zero padding) included, a non-square image."""
    from pixelsynth_amd.vqvae2.vqvae import _s2d
    L, st = _lib.lib(), _lib.current_stream()
    g = torch.Generator().manual_seed(H + W)
    x = torch.randn(B, 3, H, W, generator=g).cuda()
    w = (torch.randn(64, 3, 4, 4, generator=g) / 7).cuda()
    b = torch.randn(64, generator=g).cuda()
    y = torch.full((B, H // 4, W // 4, 256), float("nan"), device="cuda")
    _lib.check(L.ps_vq_stem_s2d_f32(x.data_ptr(), w.data_ptr(), b.data_ptr(), B, H, W, y.data_ptr(), st), "ps_vq_stem_s2d_f32")
    ref = _s2d(torch.nn.functional.conv2d(x.double(), w.double(), b.double(), 2, 1)).permute(0, 2, 3, 1)
    assert (y.double() - ref).abs().max().item() / ref.abs().max().item() < 2e-6
    Hh, Wh = H // 2, W // 2 if (W // 2) % 16 == 0 else 16
    h = torch.randn(B, Hh, Wh, 64, generator=g).cuda()
    wt = (torch.randn(64, 3, 4, 4, generator=g) / 16).cuda()
    bt = torch.randn(3, generator=g).cuda()
    img = torch.full((B, 3, 2 * Hh, 2 * Wh), float("nan"), device="cuda")
    _lib.check(L.ps_vq_head_f32(h.data_ptr(), wt.data_ptr(), bt.data_ptr(), B, Hh, Wh, img.data_ptr(), st), "ps_vq_head_f32")
    ref = torch.nn.functional.conv_transpose2d(torch.relu(h.double().permute(0, 3, 1, 2)), wt.double(), bt.double(), 2, 1)
    assert (img.double() - ref).abs().max().item() / ref.abs().max().item() < 2e-6
    assert L.ps_vq_stem_s2d_f32(x.data_ptr(), w.data_ptr(), b.data_ptr(), B, H, 40, y.data_ptr(), st) != 0
    assert L.ps_vq_head_f32(h.data_ptr(), wt.data_ptr(), bt.data_ptr(), B, Hh, 24, img.data_ptr(), st) != 0


def test_encoder_runs_again_in_fp32_when_an_activation_leaves_fp16s_range():
    """An image scaled so that the first activations pass 65 000: the split-fp16 convolutions raise the device flag, encode_codes warns and
    returns what the torch path computes."""
    from pixelsynth_amd.networks import architectures as A
    m = _module(2)
    img = dev(syn.image(5, 1, 3, 256)) * 3.0e6
    with torch.no_grad():
        with pytest.warns(UserWarning, match="run again in fp32"):
            got = m.encode_codes(img)
        with A.decoder_conv("fp32"):
            want = m.encode_codes(img)
    assert (got != want).sum().item() <= 8
    A.check_f16x3_overflow(img.device)          # (left clear)


def test_large_batches_are_cut_per_call(monkeypatch):
    """VQVAETop.encode_codes / decode_code and Unet.forward cut a batch beyond MAX_BATCH into several calls (MIOpen's fp32 kernels
    index wrongly from 2 GiB activations on: networks/architectures.py:_conv2d_batches).  With MAX_BATCH lowered to 2, five views
    give the codes and images of one call -- codes equal except where two distances tie within float noise, images to 1e-5."""
    from pixelsynth_amd.networks import Unet
    from pixelsynth_amd.vqvae2.vqvae import VQVAETop
    vq = VQVAETop()
    vq.load_state_dict({k: torch.from_numpy(v) for k, v in syn.vqvae_state_dict(0).items()}, strict=True)
    vq = vq.cuda().eval()
    x = torch.from_numpy(syn.image(21, 5, 3, 256)).cuda()
    with torch.no_grad():
        codes, img = vq.encode_codes(x), None
        img = vq.decode_code(codes)
        monkeypatch.setattr(VQVAETop, "MAX_BATCH", 2)
        codes2 = vq.encode_codes(x)
        img2 = vq.decode_code(codes)
    assert codes2.shape == codes.shape and float((codes2 == codes).float().mean()) > 0.999
    assert img2.shape == img.shape and float((img2 - img).abs().max()) < 1e-5
    unet = Unet(channels_in=3, channels_out=1, opt=syn.network_opts())
    shapes = {k: tuple(v.shape) for k, v in unet.state_dict().items()}
    unet.load_state_dict({k: torch.from_numpy(v) for k, v in syn.fill_state_dict(shapes, 5).items()}, strict=True)
    unet = unet.cuda().eval()
    with torch.no_grad():
        d1 = unet(x)
        monkeypatch.setattr(Unet, "MAX_BATCH", 2)
        d2 = unet(x)
    assert d2.shape == d1.shape and float((d2 - d1).abs().max()) < 1e-4 * max(1.0, float(d1.abs().max()))
