"""Locally masked convolution on MI355X -- drop-in for the reference's
models/lmconv/locally_masked_convolution.py (forward only; the custom backward :52-93 is training code).

The reference builds im2col (F.unfold), multiplies by the per-location 3x3 mask and calls matmul
(:25-42).  Here one C-ABI call (ps_lmconv_forward_f32 -> csrc/lmconv.hip:k_gemm) gathers the masked
taps straight from a channels-last copy of x into v_mfma_f32_16x16x4_f32 tiles.  No CPU fallback.
"""
import math

import torch
import torch.nn as nn
from torch.nn.parameter import Parameter

from .. import _lib

_WS = {}


def _workspace(device, nbytes):
    t = _WS.get(device)
    if t is None or t.numel() < nbytes:
        t = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _WS[device] = t
    return t


def compact_mask(mask, B, C_in):
    """The reference hands the mask repeated C_in times, (B*C_in, 9, L), identical across channels
    (models/z_buffermodel.py:697-699).  Returns one copy per image (B|1, 9, L), contiguous f32."""
    if mask.dim() != 3:
        raise ValueError("mask must be (B*C_in, k1*k2, L), (B, k1*k2, L) or (1, k1*k2, L)")
    if mask.size(0) == B * C_in and C_in > 1:
        mask = mask.view(B, C_in, mask.size(1), mask.size(2))[:, 0]
    elif mask.size(0) not in (1, B):
        raise ValueError(f"mask batch {mask.size(0)} matches neither B={B} nor B*C_in={B * C_in}")
    return mask.float().contiguous()


def lmconv_forward(x, mask, weight, bias=None, dilation=1):
    """y[b,o,l] = bias[o] + sum_{c,t} W[o,c,t] * mask[b,t,l] * xpad[b,c,l+dil*off(t)]  (reference :25-49)."""
    if x.dim() != 4:
        raise AssertionError("lmconv expects a 4D (B, C, H, W) input")
    out_channels, in_channels, k1, k2 = weight.shape
    if x.size(1) != in_channels or mask.size(1) != k1 * k2:
        raise AssertionError(f"lmconv: input has {x.size(1)} channels / mask {mask.size(1)} taps, "
                             f"weight wants {in_channels} / {k1 * k2}")
    if (k1, k2) != (3, 3):
        raise NotImplementedError("the HIP lmconv kernel implements the 3x3 kernels PixelSynth uses")
    _lib.require_cuda(x, mask, weight, bias)
    B, _, H, W = x.shape
    m = compact_mask(mask, B, in_channels)
    stride = 0 if m.size(0) == 1 and B > 1 else 9 * H * W
    xc = x.float().contiguous()
    wc = weight.float().contiguous()
    bc = None if bias is None else bias.float().contiguous()
    y = torch.empty(B, out_channels, H, W, dtype=torch.float32, device=x.device)
    L = _lib.lib()
    ws = _workspace(x.device, L.ps_lmconv_workspace_bytes(B, in_channels, out_channels, H, W))
    rc = L.ps_lmconv_forward_f32(_lib.ptr(xc), _lib.ptr(m), stride, _lib.ptr(wc), _lib.ptr(bc), B, in_channels,
                                 out_channels, H, W, int(dilation), _lib.ptr(y), _lib.ptr(ws), ws.numel(),
                                 _lib.current_stream())
    _lib.check(rc, "ps_lmconv_forward_f32")
    return y


class _locally_masked_conv2d:
    """Same call surface as the reference autograd.Function (inference only)."""

    @staticmethod
    def apply(x, mask, weight, mask_weight=None, bias=None, dilation=1, padding=1):
        if mask_weight is not None:
            raise NotImplementedError("conv_mask_weight=True is not used by PixelSynth (z_buffermodel.py:71)")
        return lmconv_forward(x, mask, weight, bias, dilation)

    forward = apply


class locally_masked_conv2d(nn.Module):
    """Module form: parameters `weight (Co,Ci,k,k)`, optional `mask_weight (Co,k,k)` and `bias (Co)` under the
    reference's names, default-initialised like a torch conv (uniform with bound 1/sqrt(fan_in))."""

    def __init__(self, in_channels, out_channels, kernel_size=(3, 3), dilation=1, bias=True, mask_weight=False):
        super(locally_masked_conv2d, self).__init__()
        kh, kw = kernel_size
        self.in_channels, self.out_channels, self.dilation = in_channels, out_channels, dilation
        self.padding = tuple(dilation * (k - 1) // 2 for k in (kh, kw))
        self.weight = Parameter(torch.empty(out_channels, in_channels, kh, kw))
        self.mask_weight = Parameter(torch.empty(out_channels, kh, kw)) if mask_weight else None
        self.bias = Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        bound = 1.0 / math.sqrt(self.in_channels * self.weight.shape[2] * self.weight.shape[3])
        with torch.no_grad():
            for p_ in (self.weight, self.mask_weight):
                if p_ is not None:
                    nn.init.kaiming_uniform_(p_, a=math.sqrt(5))      # = U(-1/sqrt(fan_in), +1/sqrt(fan_in))
            if self.bias is not None:
                self.bias.uniform_(-bound, bound)

    def forward(self, x, mask=None):
        return _locally_masked_conv2d.apply(x, mask, self.weight, self.mask_weight, self.bias, self.dilation,
                                            self.padding)
