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
