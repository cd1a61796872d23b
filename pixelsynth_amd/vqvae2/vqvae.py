"""VQ-VAE-2 top level around the AR loop -- inference mirror of the reference's
models/vqvae2/vqvae.py (VQVAETop :229-311, Quantize :28-78, ResBlock :81-97, Encoder :100-126, Decoder :129-161).

Same class names, attribute / parameter / buffer names and shapes (a reference ``state_dict`` loads with
``strict=True``), same call surface (``encode(x)[3]`` are the top codes, ``decode_code(codes)`` the image).
What is native here: the quantiser (nearest codebook entry, ps_vq_nearest_f32) and the code -> latent
gather (ps_vq_embed_f32) are HIP kernels, so codes stay on the device as int32 and no (N,512) one-hot / distance
matrix is built; ``encode_codes`` computes only what the top codes depend on (the reference's ``encode`` also runs
dec_t and the bottom quantiser, whose results the novel-view path discards).  The dense convolutions go through
torch (MIOpen) for now -- "Torch/MIOpen first" in SURVEY 8f.  Training (the EMA codebook update, :53-70) is out of scope.

One reference quirk is part of the numerics: ResBlock starts with an *in-place* ReLU, so the residual it adds is
relu(x), not x (:93-95).
"""
import torch
from torch import nn
from torch.nn import functional as F

from .. import _lib


class Quantize(nn.Module):
    def __init__(self, dim, n_embed, decay=0.99, eps=1e-5):
        super().__init__()
        self.dim, self.n_embed, self.decay, self.eps = dim, n_embed, decay, eps
        embed = torch.randn(dim, n_embed)
        self.register_buffer("embed", embed)
        self.register_buffer("cluster_size", torch.zeros(n_embed))
        self.register_buffer("embed_avg", embed.clone())

    def nearest(self, z, layout, hw=1):
        """z: (N,dim) [layout 0] or (B,dim,HW) [layout 1], float32 CUDA -> int32 codes (N,)."""
        _lib.require_cuda(z, self.embed)
        z = z.contiguous()
        n = z.numel() // self.dim
        idx = torch.empty(n, dtype=torch.int32, device=z.device)
        rc = _lib.lib().ps_vq_nearest_f32(_lib.ptr(z), layout, _lib.ptr(self.embed.contiguous()), n, self.dim, self.n_embed, hw,
                                          _lib.ptr(idx), None, _lib.current_stream())
        _lib.check(rc, "ps_vq_nearest_f32")
        return idx

    def forward(self, input):
        """input (..., dim) -> (quantize, diff, embed_ind) like the reference (inference)."""
        if self.training:
            raise RuntimeError("Quantize: the EMA codebook update (training) is not part of the novel-view path")
        flat = input.reshape(-1, self.dim).float()
        ind = self.nearest(flat, 0).to(torch.int64).view(*input.shape[:-1])
        quantize = self.embed_code(ind)
        diff = (quantize - input).pow(2).mean()
        return quantize, diff, ind

    def embed_code(self, embed_id):
        return F.embedding(embed_id.to(torch.int64), self.embed.transpose(0, 1))

    def embed_grid(self, codes):
        """codes (B,H,W) int -> (B,dim,H,W): embed_code + permute(0,3,1,2) in one gather kernel."""
        _lib.require_cuda(codes, self.embed)
        B, H, W = codes.shape
        c32 = codes.to(torch.int32).contiguous()
        out = torch.empty(B, self.dim, H, W, dtype=torch.float32, device=codes.device)
        rc = _lib.lib().ps_vq_embed_f32(_lib.ptr(c32), _lib.ptr(self.embed.contiguous()), B, H * W, self.dim, self.n_embed,
                                        _lib.ptr(out), _lib.current_stream())
        _lib.check(rc, "ps_vq_embed_f32")
        return out


class ResBlock(nn.Module):
    def __init__(self, in_channel, channel):
        super().__init__()
        self.conv = nn.Sequential(nn.ReLU(), nn.Conv2d(in_channel, channel, 3, padding=1), nn.ReLU(),
                                  nn.Conv2d(channel, in_channel, 1))

    def forward(self, input):
        r = F.relu(input)  # the reference's first ReLU is in place: the skip branch sees relu(x)
        return self.conv[3](F.relu(self.conv[1](r))) + r


def _res_stack(channel, n_res_block, n_res_channel):
    return [ResBlock(channel, n_res_channel) for _ in range(n_res_block)]


class Encoder(nn.Module):
    def __init__(self, in_channel, channel, n_res_block, n_res_channel, stride):
        super().__init__()
        half = channel // 2
        if stride == 4:
            head = [nn.Conv2d(in_channel, half, 4, stride=2, padding=1), nn.ReLU(),
                    nn.Conv2d(half, channel, 4, stride=2, padding=1), nn.ReLU(), nn.Conv2d(channel, channel, 3, padding=1)]
        elif stride == 2:
            head = [nn.Conv2d(in_channel, half, 4, stride=2, padding=1), nn.ReLU(), nn.Conv2d(half, channel, 3, padding=1)]
        else:
            raise ValueError("Encoder: stride must be 2 or 4")
        self.blocks = nn.Sequential(*head, *_res_stack(channel, n_res_block, n_res_channel), nn.ReLU())

    def forward(self, input):
        return self.blocks(input)


class Decoder(nn.Module):
    def __init__(self, in_channel, out_channel, channel, n_res_block, n_res_channel, stride):
        super().__init__()
        body = [nn.Conv2d(in_channel, channel, 3, padding=1), *_res_stack(channel, n_res_block, n_res_channel), nn.ReLU()]
        if stride == 4:
            tail = [nn.ConvTranspose2d(channel, channel // 2, 4, stride=2, padding=1), nn.ReLU(),
                    nn.ConvTranspose2d(channel // 2, out_channel, 4, stride=2, padding=1)]
        elif stride == 2:
            tail = [nn.ConvTranspose2d(channel, out_channel, 4, stride=2, padding=1)]
        else:
            raise ValueError("Decoder: stride must be 2 or 4")
        self.blocks = nn.Sequential(*body, *tail)

    def forward(self, input):
        return self.blocks(input)


class VQVAETop(nn.Module):
    def __init__(self, in_channel=3, channel=128, n_res_block=2, n_res_channel=32, embed_dim=64, n_embed=512, decay=0.99):
        super().__init__()
        self.enc_b = Encoder(in_channel, channel, n_res_block, n_res_channel, stride=4)
        self.enc_t = Encoder(channel, channel, n_res_block, n_res_channel, stride=2)
        self.quantize_conv_t = nn.Conv2d(channel, embed_dim, 1)
        self.quantize_t = Quantize(embed_dim, n_embed)
        self.dec_t = Decoder(embed_dim, embed_dim, channel, n_res_block, n_res_channel, stride=2)
        self.quantize_conv_b = nn.Conv2d(embed_dim + channel, embed_dim, 1)
        self.quantize_b = Quantize(embed_dim, n_embed)
        self.upsample_t = nn.ConvTranspose2d(embed_dim, embed_dim, 4, stride=2, padding=1)
        self.dec = Decoder(embed_dim, in_channel, channel, n_res_block, n_res_channel, stride=4)

    # ---------------------------------------------------------------- the novel-view path
    MAX_BATCH = 256   # views per call on the GPU, at most
    _MIOPEN_SAFE_BYTES = 2 ** 31   # where MIOpen's fp32 kernels start to index wrongly (networks/architectures.py:_conv2d_batches)

    def _views_per_call(self, H, W):
        """Views per call on the GPU for (H, W) IMAGES: the widest activation on either side of the codes -- channel / 2 maps at half the
        image's size (enc_b's first convolution, dec's last but one), channel maps at a quarter -- stays below the 2 GiB at which MIOpen's
        fp32 kernels index wrongly, whatever the image size (256 x 256, 64 maps: 512 views; capped at MAX_BATCH)."""
        c = self.enc_b.blocks[0].out_channels          # channel // 2
        per = max(c * (H // 2) * (W // 2), 2 * c * (H // 4) * (W // 4), 3 * H * W) * 4
        return max(1, min(self.MAX_BATCH, (self._MIOPEN_SAFE_BYTES - 1) // per))

    @torch.no_grad()
    def encode_codes(self, input):
        """(B,3,256,256) -> top codes (B,32,32) int32, on the device (= ``encode(input)[3]``, z_buffermodel.py:345)."""
        n = self._views_per_call(input.size(2), input.size(3)) if input.is_cuda else input.size(0)
        if input.size(0) > n:
            return torch.cat([self.encode_codes(input[i:i + n]) for i in range(0, input.size(0), n)])
        lat = self.quantize_conv_t(self.enc_t(self.enc_b(input)))  # (B,64,32,32)
        B, _, H, W = lat.shape
        return self.quantize_t.nearest(lat.float(), 1, H * W).view(B, H, W)

    @torch.no_grad()
    def decode_code(self, code_t):
        """codes (B,32,32) int -> image (B,3,256,256) (vqvae.py:305-311, z_buffermodel.py:250)."""
        n = self._views_per_call(8 * code_t.size(-2), 8 * code_t.size(-1)) if code_t.is_cuda else code_t.size(0)
        if code_t.size(0) > n:
            return torch.cat([self.decode_code(code_t[i:i + n]) for i in range(0, code_t.size(0), n)])
        return self.decode(self.quantize_t.embed_grid(code_t))

    # ---------------------------------------------------------------- reference-shaped surface
    def _quantise(self, quantizer, features):
        """(B,C,H,W) features -> (quantised (B,C,H,W), diff (1,), codes (B,H,W)) through a Quantize module."""
        q, diff, ids = quantizer(features.movedim(1, -1))
        return q.movedim(-1, 1), diff.reshape(1), ids

    def encode(self, input):
        """-> (quant_t, quant_b, diff, id_t, id_b) like the reference; only id_t matters to the novel-view path."""
        bottom = self.enc_b(input)
        quant_t, diff_t, id_t = self._quantise(self.quantize_t, self.quantize_conv_t(self.enc_t(bottom)))
        fused = torch.cat([self.dec_t(quant_t), bottom], dim=1)
        quant_b, diff_b, id_b = self._quantise(self.quantize_b, self.quantize_conv_b(fused))
        return quant_t, quant_b, diff_t + diff_b, id_t, id_b

    def forward(self, input):
        quant_t, _, diff, _, _ = self.encode(input)
        return self.decode(quant_t), diff

    def decode(self, quant_t):
        return self.dec(self.upsample_t(quant_t))
