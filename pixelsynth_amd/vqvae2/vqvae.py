"""VQ-VAE-2 top level around the AR loop -- inference mirror of the reference's
models/vqvae2/vqvae.py (VQVAETop :229-311, Quantize :28-78, ResBlock :81-97, Encoder :100-126, Decoder :129-161).

Same class names, attribute / parameter / buffer names and shapes (a reference ``state_dict`` loads with
``strict=True``), same call surface (``encode(x)[3]`` are the top codes, ``decode_code(codes)`` the image).
What is native here: the quantiser (nearest codebook entry, ps_vq_nearest_f32) and the code -> latent
gather (ps_vq_embed_f32) are HIP kernels, so codes stay on the device as int32 and no (N,512) one-hot / distance
matrix is built; ``encode_codes`` computes only what the top codes depend on (the reference's ``encode`` also runs
dec_t and the bottom quantiser, whose results the novel-view path discards).  On the inference GPU path the convolutions of
``encode_codes`` / ``decode_code`` are hand-written too (``_FastPath`` below, round 6): the 3 x 3 layers and -- rewritten as 3 x 3
layers over space-to-depth blocks / towards depth-to-space blocks -- the 4 x 4 stride-2 convolutions and transposed convolutions run
through csrc/conv_f16x3.hip (split-fp16 MFMA, fp32 in / out, ReLU applied as the patch is staged, bias on the way out), a ResBlock's
tail (ReLU, 1 x 1, + the ReLU'd input) is one launch of csrc/conv1x1.hip; the 3-channel ends (3 -> 64, 64 -> 3) are csrc/vq_ends.hip.
PS_VQVAE_CONV=fp32 (or networks.architectures.decoder_conv("fp32") in effect) sends everything through torch (MIOpen).
Training (the EMA codebook update, :53-70) is out of scope.

One reference quirk is part of the numerics: ResBlock starts with an *in-place* ReLU, so the residual it adds is
relu(x), not x (:93-95).
"""
import os

import torch
from torch import nn
from torch.nn import functional as F

from .. import _lib

VQVAE_CONV = os.environ.get("PS_VQVAE_CONV", "f16x3")   # "f16x3" | "fp32" (every convolution through torch / MIOpen)


def _s2d(x):
    """(B, C, H, W) -> (B, 4 C, H / 2, W / 2) channels_last, channel (sy, sx, c) = pixel (2 y + sy, 2 x + sx) of channel c."""
    B, C, H, W = x.shape
    t = x.permute(0, 2, 3, 1).reshape(B, H // 2, 2, W // 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H // 2, W // 2, 4 * C)
    return t.contiguous().permute(0, 3, 1, 2)


def _d2s(y, C):
    """(B, 4 C, H, W) channels_last with channel (py, px, c) -> (B, C, 2 H, 2 W) channels_last: pixel (2 y + py, 2 x + px)."""
    B, _, H, W = y.shape
    t = y.permute(0, 2, 3, 1).reshape(B, H, W, 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(B, 2 * H, 2 * W, C)
    return t.contiguous().permute(0, 3, 1, 2)


def s2d_weight(w):
    """Conv2d(Ci, Co, 4, stride 2, padding 1) as a 3 x 3 convolution (padding 1) over the space-to-depth blocks of its input:
    (Co, Ci, 4, 4) -> (Co, 4 Ci, 3, 3).  Output (y, x) reads input rows 2 y - 1 + ky; row 2 (y + by) + sy of block row by is tap
    ky = 2 by + sy + 1 where that lies in 0 .. 3 -- 16 of the 36 (block, sub-position) pairs; the others are zero weights."""
    Co, Ci = w.shape[:2]
    out = w.new_zeros(Co, 2, 2, Ci, 3, 3)
    for by in (-1, 0, 1):
        for sy in (0, 1):
            ky = 2 * by + sy + 1
            if not 0 <= ky <= 3:
                continue
            for bx in (-1, 0, 1):
                for sx in (0, 1):
                    kx = 2 * bx + sx + 1
                    if 0 <= kx <= 3:
                        out[:, sy, sx, :, by + 1, bx + 1] = w[:, :, ky, kx]
    return out.reshape(Co, 4 * Ci, 3, 3)


def convt_weight(wt):
    """ConvTranspose2d(Ci, Co, 4, stride 2, padding 1) as a 3 x 3 convolution (padding 1) at the INPUT's resolution towards the four
    output parities: (Ci, Co, 4, 4) -> (4 Co, Ci, 3, 3), output channel (py, px, co) = output pixel (2 y + py, 2 x + px).  Output row
    2 y + py takes input row y + d through tap ky = py + 1 - 2 d where that lies in 0 .. 3 -- two of the three d per parity."""
    Ci, Co = wt.shape[:2]
    out = wt.new_zeros(2, 2, Co, Ci, 3, 3)
    for py in (0, 1):
        for dy in (-1, 0, 1):
            ky = py + 1 - 2 * dy
            if not 0 <= ky <= 3:
                continue
            for px in (0, 1):
                for dx in (-1, 0, 1):
                    kx = px + 1 - 2 * dx
                    if 0 <= kx <= 3:
                        out[py, px, :, :, dy + 1, dx + 1] = wt[:, :, ky, kx].transpose(0, 1)
    return out.reshape(4 * Co, Ci, 3, 3)


class _FastPath:
    """encode_codes / decode_code of a VQVAETop with the convolutions through csrc/conv_f16x3.hip and csrc/conv1x1.hip (module docstring).
    Built per (module, device) from the module's parameters as they are -- packed weights, padded biases -- and rebuilt when one of
    them changes (storage or version counter)."""

    def __init__(self, m, device):
        self.device = device
        self.key = self.key_of(m)
        L = _lib.lib()
        self.L = L

        def layer3(w3, bias):
            """(Co, Ci, 3, 3) fp32 + bias -> packed for ps_conv3x3_f16x3_nhwc, Co padded to a multiple of 64 with zero channels."""
            Co, Ci = w3.shape[:2]
            Cop = -(-Co // 64) * 64
            if Ci % 32:
                raise ValueError("VQ-VAE fast path: %d input channels" % Ci)
            w = w3.new_zeros(Cop, Ci, 3, 3)
            w[:Co] = w3
            b = w3.new_zeros(Cop)
            b[:Co] = bias
            wl = w.permute(0, 2, 3, 1).contiguous()
            top = float(wl.abs().max())
            if not (top == top and top < 6.0e4):
                raise ValueError("VQ-VAE fast path: a weight fp16 cannot hold")
            packed = torch.empty(L.ps_conv3x3_f16x3_packed_bytes(Cop, Ci), dtype=torch.uint8, device=device)
            _lib.check(L.ps_conv3x3_f16x3_pack(wl.data_ptr(), Cop, Ci, packed.data_ptr(), _lib.current_stream()), "ps_conv3x3_f16x3_pack")
            return dict(packed=packed, Ci=Ci, Co=Cop, live=Co, bias=b.contiguous())

        def conv(c):
            return c.weight.detach().float(), c.bias.detach().float()

        def res(block):
            c3, c1 = block.conv[1], block.conv[3]
            w1, b1 = conv(c1)
            return dict(c3=layer3(*conv(c3)), w1=w1.reshape(w1.size(0), w1.size(1)).contiguous(), b1=b1.contiguous(), mid=c3.out_channels)

        with torch.no_grad():
            eb, et, dc = m.enc_b.blocks, m.enc_t.blocks, m.dec.blocks
            w, b = conv(eb[2])
            self.e2 = layer3(s2d_weight(w), b)
            self.e3 = layer3(*conv(eb[4]))
            self.eb_res = [res(eb[5]), res(eb[6])]
            w, b = conv(et[0])
            self.t1 = layer3(s2d_weight(w), b)
            self.t1_out = et[0].out_channels
            self.t2 = layer3(*conv(et[2]))
            self.et_res = [res(et[3]), res(et[4])]
            wq, bq = conv(m.quantize_conv_t)
            self.wq, self.bq = wq.reshape(wq.size(0), wq.size(1)).contiguous(), bq.contiguous()
            w, b = conv(m.upsample_t)
            self.up = layer3(convt_weight(w), b.repeat(4))
            self.up_out = m.upsample_t.out_channels
            self.d1 = layer3(*conv(dc[0]))
            self.dc_res = [res(dc[1]), res(dc[2])]
            w, b = conv(dc[4])
            self.d2 = layer3(convt_weight(w), b.repeat(4))
            self.d2_out = dc[4].out_channels
            # the 3-channel ends (csrc/vq_ends.hip) where they are the shapes it is written for; otherwise torch
            self.ends = (tuple(eb[0].weight.shape) == (64, 3, 4, 4) and tuple(dc[6].weight.shape) == (64, 3, 4, 4)
                         and eb[0].bias is not None and dc[6].bias is not None)
            if self.ends:
                self.stem_w, self.stem_b = (t.contiguous() for t in conv(eb[0]))
                self.head_w, self.head_b = (t.contiguous() for t in conv(dc[6]))
        self.ones, self.zeros = {}, {}

    @staticmethod
    def key_of(m):
        return tuple((p.data_ptr(), p._version) for p in m.parameters()) + ((m.quantize_t.embed.data_ptr(), m.quantize_t.embed._version),)

    @staticmethod
    def takes(m, H, W):
        """256 x 256-like images: every layer's output a multiple of 16 pixels a side, the module as VQVAETop builds it."""
        try:
            eb, et, dc = m.enc_b.blocks, m.enc_t.blocks, m.dec.blocks
            ok = (len(eb) == 8 and len(et) == 6 and len(dc) == 7 and H % 128 == 0 and W % 128 == 0
                  and eb[2].in_channels % 8 == 0 and et[0].in_channels % 8 == 0 and eb[5].conv[1].out_channels in (32, 64, 128, 256)
                  and all(isinstance(c, nn.Conv2d) for c in (eb[0], eb[2], eb[4], et[0], et[2]))
                  and all(isinstance(c, nn.ConvTranspose2d) for c in (m.upsample_t, dc[4], dc[6])))
            return bool(ok)
        except (AttributeError, IndexError, TypeError):
            return False

    def _act(self, B, C):
        """scale = 1, shift = 0 per (frame, channel): the kernel's norm + ReLU on the way in as a plain ReLU."""
        if (B, C) not in self.ones:
            self.ones[(B, C)] = torch.ones(B, C, device=self.device)
            self.zeros[(B, C)] = torch.zeros(B, C, device=self.device)
        return self.ones[(B, C)], self.zeros[(B, C)]

    def conv3(self, x, layer, relu_in, s2d=False, d2s=False):
        """The layer on x (B, C, H, W) channels_last.  s2d: x is the tensor BEFORE the space-to-depth step -- (B, Ci / 4, 2 H, 2 W) -- and the
        kernel reads it in that form; d2s: the result leaves as (B, Co / 4, 2 H, 2 W), the depth-to-space step done by the stores."""
        from ..networks.architectures import _overflow_flag, _empty_nhwc
        B, C, H, W = x.shape
        if s2d:
            C, H, W = 4 * C, H // 2, W // 2
        assert C == layer["Ci"] and x.is_contiguous(memory_format=torch.channels_last)
        y = _empty_nhwc(B, layer["Co"] // 4, 2 * H, 2 * W, x) if d2s else _empty_nhwc(B, layer["Co"], H, W, x)
        sc, sh = self._act(B, C) if relu_in else (None, None)
        p = lambda t: None if t is None else t.data_ptr()
        _lib.check(self.L.ps_conv3x3_f16x3_ex_nhwc(x.data_ptr(), p(sc), p(sh), layer["packed"].data_ptr(), layer["bias"].data_ptr(), None, B, H, W,
                                                   C, layer["Co"], layer["live"], int(s2d), int(d2s), y.data_ptr(), _overflow_flag(x.device).data_ptr(),
                                                   _lib.current_stream()), "ps_conv3x3_f16x3_ex_nhwc")
        return y

    def conv1(self, x, ldx, w, bias, res, flags):
        from ..networks.architectures import _empty_nhwc
        B, _, H, W = x.shape
        Co, Ci = w.shape
        y = _empty_nhwc(B, Co, H, W, x)
        _lib.check(self.L.ps_conv1x1_ex_nhwc_f32(x.data_ptr(), ldx, w.data_ptr(), bias.data_ptr(), None if res is None else res.data_ptr(), flags,
                                                 B * H * W, Ci, Co, y.data_ptr(), _lib.current_stream()), "ps_conv1x1_ex_nhwc_f32")
        return y

    def res(self, x, r):
        """ResBlock (vqvae.py:81-97): conv1x1(relu(conv3x3(relu(x)))) + relu(x) -- two launches."""
        h = self.conv3(x, r["c3"], True)                       # (B, 64-padded, H, W): the first `mid` channels are the layer's
        return self.conv1(h, h.size(1), r["w1"], r["b1"], x, 3)

    def encode_latent(self, m, input):
        """-> quantize_conv_t(enc_t(enc_b(input))) as (B, H / 8, W / 8, embed_dim) contiguous."""
        eb = m.enc_b.blocks
        B, _, H, W = input.shape
        if self.ends:                                                  # 3 -> 64 at half size, written as the 2 x 2 blocks the next layer reads
            s = torch.empty(B, H // 4, W // 4, 4 * eb[0].out_channels, device=input.device)
            x = input.contiguous()
            _lib.check(self.L.ps_vq_stem_s2d_f32(x.data_ptr(), self.stem_w.data_ptr(), self.stem_b.data_ptr(), B, H, W, s.data_ptr(),
                                                 _lib.current_stream()), "ps_vq_stem_s2d_f32")
            s = s.permute(0, 3, 1, 2)
        else:
            s = _s2d(F.conv2d(input, eb[0].weight, eb[0].bias, 2, 1))
        h = self.conv3(s, self.e2, True)
        h = self.conv3(h, self.e3, True)
        for r in self.eb_res:
            h = self.res(h, r)
        h = self.conv3(h, self.t1, True, s2d=True)                     # (the trailing ReLU of enc_b on the way in; 2 x 2 blocks read in place)
        if h.size(1) != self.t1_out:
            h = h[:, :self.t1_out].contiguous(memory_format=torch.channels_last)
        h = self.conv3(h, self.t2, True)
        for r in self.et_res:
            h = self.res(h, r)
        lat = self.conv1(h, h.size(1), self.wq, self.bq, None, 1)
        return lat.permute(0, 2, 3, 1)

    def decode(self, m, quant_nhwc):
        """quant (B, H, W, embed_dim) contiguous -> image (B, 3, 8 H, 8 W)."""
        dc = m.dec.blocks
        q = quant_nhwc.permute(0, 3, 1, 2)
        h = self.conv3(q, self.up, False, d2s=True)                    # (the four output parities stored where they belong)
        h = self.conv3(h, self.d1, False)
        for r in self.dc_res:
            h = self.res(h, r)
        h = self.conv3(h, self.d2, True, d2s=True)
        if self.ends:                                                  # ReLU, 64 -> 3 at twice the size: the image, NCHW
            B, _, Hh, Wh = h.shape
            img = torch.empty(B, 3, 2 * Hh, 2 * Wh, device=h.device)
            _lib.check(self.L.ps_vq_head_f32(h.data_ptr(), self.head_w.data_ptr(), self.head_b.data_ptr(), B, Hh, Wh, img.data_ptr(),
                                             _lib.current_stream()), "ps_vq_head_f32")
            return img
        return F.conv_transpose2d(F.relu(h), dc[6].weight, dc[6].bias, 2, 1).contiguous()


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
        fast = self._fast(input, input.size(2), input.size(3))
        if fast is not None:
            from ..networks.architectures import check_f16x3_overflow, clear_f16x3_overflow, decoder_conv
            clear_f16x3_overflow(input.device)
            lat = fast.encode_latent(self, input.float())            # (B, 32, 32, 64)
            B, H, W, D = lat.shape
            codes = self.quantize_t.nearest(lat.reshape(-1, D), 0).view(B, H, W)
            try:
                check_f16x3_overflow(input.device)                   # (synchronises: the codes steer everything that follows)
                return codes
            except RuntimeError as err:
                import warnings
                warnings.warn(f"VQ-VAE encoder: {err}: run again in fp32")
                with decoder_conv("fp32"):
                    return self.encode_codes(input)
        lat = self.quantize_conv_t(self.enc_t(self.enc_b(input)))  # (B,64,32,32)
        B, _, H, W = lat.shape
        return self.quantize_t.nearest(lat.float(), 1, H * W).view(B, H, W)

    def _fast(self, t, H, W):
        """The hand-written path for a float32 CUDA call on (H, W) images in eval mode, or None (module docstring)."""
        from ..networks import architectures as A
        mode = A._FORCED_CONV[-1] if A._FORCED_CONV else VQVAE_CONV
        if (mode != "f16x3" or not t.is_cuda or self.training or torch.is_grad_enabled() or not _FastPath.takes(self, H, W)
                or any(p.dtype != torch.float32 or p.device != t.device for p in self.parameters())):
            return None
        key = (str(t.device), _FastPath.key_of(self))
        cache = self.__dict__.get("_ps_fast")
        if cache is None or cache[0] != key:
            try:
                cache = (key, _FastPath(self, t.device))
            except ValueError:          # (a weight fp16 cannot hold, channel counts the kernels do not take: torch)
                cache = (key, None)
            self.__dict__["_ps_fast"] = cache
        return cache[1]

    @torch.no_grad()
    def decode_code(self, code_t):
        """codes (B,32,32) int -> image (B,3,256,256) (vqvae.py:305-311, z_buffermodel.py:250)."""
        n = self._views_per_call(8 * code_t.size(-2), 8 * code_t.size(-1)) if code_t.is_cuda else code_t.size(0)
        if code_t.size(0) > n:
            return torch.cat([self.decode_code(code_t[i:i + n]) for i in range(0, code_t.size(0), n)])
        fast = self._fast(code_t, 8 * code_t.size(-2), 8 * code_t.size(-1))
        if fast is not None:    # (split-fp16 convolutions: the caller's overflow check -- z_buffermodel._decode_checked -- covers them)
            return fast.decode(self, self.quantize_t.embed_code(code_t))
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
