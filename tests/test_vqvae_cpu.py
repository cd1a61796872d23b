"""CPU: the VQ-VAE-2 top-level oracle (oracle/vqvae_oracle.py) against the fixture produced by the reference's own
VQVAETop (tests/golden/vqvae.npz, make_golden.py:gen_vqvae), and name / shape compatibility of the mirror module."""
import os

import numpy as np
import torch

from oracle import vqvae_oracle as vo
from pixelsynth_amd import synthetic as syn

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "vqvae.npz"))


def state_dict():
    sd = {k: torch.from_numpy(v) for k, v in syn.vqvae_state_dict(0).items()}
    sd["quantize_t.embed"] = torch.from_numpy(G["embed"])
    return sd


def test_state_dict_names_and_shapes_match_the_reference():
    from pixelsynth_amd.vqvae2 import VQVAETop
    m = VQVAETop()
    ours = [f"{k}:{','.join(map(str, v.shape))}" for k, v in m.state_dict().items()]
    assert ours == list(G["keys"])
    m.load_state_dict({k: torch.from_numpy(v) for k, v in syn.vqvae_state_dict(0).items()}, strict=True)


def test_codebook_generator_is_reproducible():
    sd = state_dict()
    with torch.no_grad():
        latA = vo.top_latent(sd, torch.from_numpy(syn.image(int(G["image_seeds"][0]), 1, 3, 256)))
    emb = syn.codebook_from_latents(latA.numpy(), 0)
    np.testing.assert_allclose(emb, G["embed"], rtol=0, atol=2e-6)


def test_oracle_encode_decode_against_reference_outputs():
    sd = state_dict()
    img = torch.from_numpy(syn.image(int(G["image_seeds"][1]), 1, 3, 256))
    with torch.no_grad():
        codes, dist, lat = vo.encode_codes(sd, img)
        dec = vo.decode_code(sd, torch.from_numpy(G["codes"]).long())
    np.testing.assert_allclose(lat.numpy(), G["latB"], rtol=1e-5, atol=1e-5)
    assert np.array_equal(codes.numpy().astype(np.int32), G["codes"])          # same library, same arithmetic: exact
    two = np.sort(dist.reshape(-1, 512).numpy(), 1)[:, :2]
    np.testing.assert_allclose(two, G["two_smallest"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(dec.numpy()[:, :, ::4, ::4], G["dec_sub"], rtol=1e-5, atol=1e-6)
    assert len(np.unique(G["codes"])) > 200                                      # the fixture is not degenerate


def test_stride_two_layers_as_3x3_convolutions_over_blocks():
    """vqvae.s2d_weight / convt_weight: Conv2d(4, stride 2, padding 1) is a 3 x 3 convolution over the space-to-depth blocks of its
    input, ConvTranspose2d(4, stride 2, padding 1) a 3 x 3 convolution at the input's resolution towards the four output parities --
    the forms csrc/conv_f16x3.hip runs them in.  Against torch's own layers in fp64, odd channel counts, a non-square image."""
    import torch
    import torch.nn.functional as F
    from pixelsynth_amd.vqvae2.vqvae import _d2s, _s2d, convt_weight, s2d_weight
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 5, 8, 12, generator=g, dtype=torch.float64)
    w = torch.randn(7, 5, 4, 4, generator=g, dtype=torch.float64)
    got = F.conv2d(_s2d(x), s2d_weight(w), None, 1, 1)
    assert got.shape == (2, 7, 4, 6) and (got - F.conv2d(x, w, None, 2, 1)).abs().max() < 1e-12
    assert int((s2d_weight(torch.ones(1, 1, 4, 4)) != 0).sum()) == 16                  # 16 of the 36 (block, sub-position) pairs
    wt = torch.randn(5, 3, 4, 4, generator=g, dtype=torch.float64)
    y = F.conv2d(x, convt_weight(wt), None, 1, 1).contiguous(memory_format=torch.channels_last)
    got = _d2s(y, 3)
    assert got.shape == (2, 3, 16, 24) and (got - F.conv_transpose2d(x, wt, None, 2, 1)).abs().max() < 1e-12
    assert int((convt_weight(torch.ones(1, 1, 4, 4)) != 0).sum()) == 16
    back = _d2s(_s2d(x).contiguous(memory_format=torch.channels_last), 5)              # the two permutations are inverses
    assert torch.equal(back, x)
