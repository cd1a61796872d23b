"""CPU restatement (torch fp32, functional, straight from a state_dict) of the VQ-VAE-2 top level pieces on the
novel-view path -- TEST INFRASTRUCTURE ONLY (tests/, smoke, bench cpu leg); the product never imports it.

Follows the reference's models/vqvae2/vqvae.py: Encoder :100-126, ResBlock :81-97 (in-place first ReLU: the skip
adds relu(x)), Quantize.forward :41-51 (distance = |z|^2 - 2 z.E + |E|^2, first arg-max of -dist), Decoder :129-161,
VQVAETop.encode / decode_code :262-311.  Pinned against tests/golden/vqvae.npz (outputs of the reference itself).
"""
import torch
import torch.nn.functional as F


def _res(sd, p, x):
    r = F.relu(x)
    h = F.conv2d(r, sd[p + ".conv.1.weight"], sd[p + ".conv.1.bias"], padding=1)
    return F.conv2d(F.relu(h), sd[p + ".conv.3.weight"], sd[p + ".conv.3.bias"]) + r


def encoder_b(sd, x, p="enc_b.blocks"):
    x = F.relu(F.conv2d(x, sd[p + ".0.weight"], sd[p + ".0.bias"], stride=2, padding=1))
    x = F.relu(F.conv2d(x, sd[p + ".2.weight"], sd[p + ".2.bias"], stride=2, padding=1))
    x = F.conv2d(x, sd[p + ".4.weight"], sd[p + ".4.bias"], padding=1)
    return F.relu(_res(sd, p + ".6", _res(sd, p + ".5", x)))


def encoder_t(sd, x, p="enc_t.blocks"):
    x = F.relu(F.conv2d(x, sd[p + ".0.weight"], sd[p + ".0.bias"], stride=2, padding=1))
    x = F.conv2d(x, sd[p + ".2.weight"], sd[p + ".2.bias"], padding=1)
    return F.relu(_res(sd, p + ".4", _res(sd, p + ".3", x)))


def top_latent(sd, img):
    """(B,3,S,S) -> pre-quantisation latent (B,64,S/8,S/8)."""
    return F.conv2d(encoder_t(sd, encoder_b(sd, img)), sd["quantize_conv_t.weight"], sd["quantize_conv_t.bias"])


def nearest(z, embed):
    """z (N,D), embed (D,K) -> (idx (N,) int64, dist (N,K)) with the reference's expression."""
    dist = z.pow(2).sum(1, keepdim=True) - 2 * z @ embed + embed.pow(2).sum(0, keepdim=True)
    return (-dist).max(1)[1], dist


def encode_codes(sd, img):
    lat = top_latent(sd, img)
    B, D, H, W = lat.shape
    idx, dist = nearest(lat.permute(0, 2, 3, 1).reshape(-1, D), sd["quantize_t.embed"])
    return idx.view(B, H, W), dist.view(B, H, W, -1), lat


def decode_code(sd, codes):
    q = F.embedding(codes.long(), sd["quantize_t.embed"].t()).permute(0, 3, 1, 2)
    x = F.conv_transpose2d(q, sd["upsample_t.weight"], sd["upsample_t.bias"], stride=2, padding=1)
    p = "dec.blocks"
    x = F.conv2d(x, sd[p + ".0.weight"], sd[p + ".0.bias"], padding=1)
    x = F.relu(_res(sd, p + ".2", _res(sd, p + ".1", x)))
    x = F.relu(F.conv_transpose2d(x, sd[p + ".4.weight"], sd[p + ".4.bias"], stride=2, padding=1))
    return F.conv_transpose2d(x, sd[p + ".6.weight"], sd[p + ".6.bias"], stride=2, padding=1)
