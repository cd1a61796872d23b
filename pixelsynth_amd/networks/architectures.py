"""The dense networks either side of the hot path: the depth regressor `Unet`
(models/networks/architectures.py:174-279) and the refinement `ResNetDecoder` (:126-167) with its blocks
(models/layers/blocks.py:34-73) and noise-conditioned normalisation (models/layers/normalization.py:21-47, :97-200).

Same constructor arguments, attribute / parameter names and shapes as the reference, so its checkpoints load with
`load_state_dict(strict=True)`; the bodies are written for inference on one MI355X:

  * NHWC (channels_last) on the GPU; inputs are converted on entry, results handed back NCHW-contiguous because the HIP kernels
    downstream take raw NCHW pointers;
  * the DECODER's 3 x 3 convolutions are hand-written (csrc/conv_f16x3.hip: split-fp16 MFMA, fp32 in / out, the norm + ReLU in
    front of them applied as the input is staged; csrc/conv_thin.hip for the 4 -> 64 and 128 -> 3 layers): 8.3 instead of 20 ms
    per 16 views; its 1 x 1 projections run on the fp32 matrix pipe (csrc/conv1x1.hip, round 6).  PS_DECODER_CONV=fp32 /
    opt.decoder_conv = "fp32" sends them through torch.  The Unet's convolutions and the decoder's 3 -> 3 layer run through torch (MIOpen), the batch cut so that no call sees 2 GiB (MIOpen's fp32 NHWC
    kernels are silently wrong on 4 GiB activations -- 128 views of the decoder's widest layer);
  * spectral-normalised weights are computed once per checkpoint in eval mode, not at every forward (_normalised_weight);
  * `LinearNoiseLayer` + stored-statistics batch norm + ReLU is ONE per-(sample, channel) affine and a clamp
    (`_noise_affine`), not four elementwise passes;
  * `ResNetDecoder.forward(..., noise=)` takes the noise draws explicitly (a list of (B,20) tensors, two per block) so
    a run is reproducible and comparable with the reference; without it the draws come from torch.randn as there.

Training-mode batch statistics are supported for the plain BatchNorm2d layers (torch's own); the noise-conditioned
layers implement the stored-statistics (eval) form only -- this repository does not train.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

NOISE_SZ = 20

# decoder tables of get_resnet_arch (models/networks/configs.py): first width, then (width, resample) per block.
_DEC_WIDTHS = lambda g: [g, 2 * g, 4 * g, 4 * g, 2 * g, 2 * g, 2 * g, 3]
_DEC_RESAMPLE = [None, "Down", "Down", None, "Up", "Up", None, None]
_DEC_FIRST = {"256W8UpDown": lambda c: 128, "256W8UpDown64": lambda c: 64, "256W8UpDownDV": lambda c: 64,
              "256W8UpDownRGB": lambda c: 3, "256W8UpDown3": lambda c: c, "256W8UpDown3SuperRes": lambda c: c}


def _spectral(opt):
    return "spectral" in opt.norm_G


def _conv(opt, cin, cout, k, stride, pad):
    conv = nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=pad)
    return nn.utils.spectral_norm(conv) if _spectral(opt) else conv


def _norm_layer(opt):
    kind = opt.norm_G.split(":")[1]
    if kind in ("batch", "spectral_batch"):
        return nn.BatchNorm2d
    if kind == "spectral_instance":
        return nn.InstanceNorm2d
    if kind == "spectral_batchstanding":
        return BatchNorm_StandingStats
    raise ValueError("unknown norm_G %r" % opt.norm_G)


class bn(nn.Module):
    """Stored statistics of the BigGAN-style batch norm (normalization.py:117-166); buffers only."""

    def __init__(self, num_channels, eps=1e-5, momentum=0.1):
        super().__init__()
        self.eps, self.momentum, self.accumulate_standing = eps, momentum, False
        self.register_buffer("stored_mean", torch.zeros(num_channels))
        self.register_buffer("stored_var", torch.ones(num_channels))
        self.register_buffer("accumulation_counter", torch.zeros(1))

    def scale_shift(self, gain, bias):
        """y = x * scale - shift with gain/bias (B or 1, C, 1, 1) folded in (fused_bn, normalization.py:170-184)."""
        if self.training:
            raise RuntimeError("noise-conditioned batch norm: only the stored-statistics (eval) form is built here")
        mean, var = self.stored_mean.view(1, -1, 1, 1), self.stored_var.view(1, -1, 1, 1)
        if self.accumulate_standing:
            mean, var = mean / self.accumulation_counter, var / self.accumulation_counter
        scale = torch.rsqrt(var + self.eps) * gain
        return scale, mean * scale - bias

    def forward(self, x, gain, bias):
        scale, shift = self.scale_shift(gain, bias)
        return x * scale - shift


class BatchNorm_StandingStats(nn.Module):
    def __init__(self, output_size, eps=1e-5, momentum=0.1):
        super().__init__()
        self.output_size, self.eps, self.momentum = output_size, eps, momentum
        self.gain = nn.Parameter(torch.ones(output_size))
        self.bias = nn.Parameter(torch.zeros(output_size))
        self.bn = bn(output_size, eps, momentum)

    def forward(self, x, y=None):
        return self.bn(x, self.gain.view(1, -1, 1, 1), self.bias.view(1, -1, 1, 1))


class LinearNoiseLayer(nn.Module):
    """Gain and bias of a batch norm predicted from a noise vector (normalization.py:21-47)."""

    def __init__(self, opt, noise_sz=NOISE_SZ, output_sz=32):
        super().__init__()
        self.noise_sz = noise_sz
        lin = lambda: nn.Linear(noise_sz, output_sz, bias=False)
        self.gain = nn.utils.spectral_norm(lin()) if _spectral(opt) else lin()
        self.bias = nn.utils.spectral_norm(lin()) if _spectral(opt) else lin()
        self.bn = bn(output_sz)

    def affine(self, x, noise=None):
        if noise is None:
            noise = torch.randn(x.size(0), self.noise_sz).to(x.device)
        B = noise.size(0)
        return self.bn.scale_shift((1 + _sn_linear(self.gain, noise)).view(B, -1, 1, 1), _sn_linear(self.bias, noise).view(B, -1, 1, 1))

    def affine_bc(self, x, noise=None, pend=None):
        """affine() as contiguous (B, C) scale / shift -- what the HIP passes take -- with `pend` (C), a convolution bias still missing
        from x, folded into shift.  On the inference GPU path ONE launch (ps_noise_affine_f32) instead of two matrix products and six
        elementwise kernels; elsewhere composed from affine()."""
        if noise is None:
            noise = torch.randn(x.size(0), self.noise_sz).to(x.device)
        B, C = noise.size(0), self.bn.stored_mean.numel()
        bufs = (self.bn.stored_mean, self.bn.stored_var)
        if (x.is_cuda and not torch.is_grad_enabled() and not self.training and not self.bn.accumulate_standing and B == x.size(0)
                and noise.dtype == torch.float32 and noise.is_cuda and noise.device == x.device   # (the kernel takes raw pointers: a host
                # tensor or one of another device must go the torch way, where it raises the usual device-mismatch error)
                and all(t.is_cuda and t.device == x.device and t.dtype == torch.float32 and t.is_contiguous() for t in bufs)
                and (pend is None or (pend.is_cuda and pend.device == x.device and pend.dtype == torch.float32 and pend.is_contiguous()))
                and all(type(m) is nn.Linear and m.bias is None for m in (self.gain, self.bias))):
            from torch.nn.utils.spectral_norm import SpectralNorm
            ws = []
            for lin in (self.gain, self.bias):
                pre = list(lin._forward_pre_hooks.values())
                if lin._forward_hooks or getattr(lin, "parametrizations", None) or not all(isinstance(h, SpectralNorm) for h in pre):
                    ws = None
                    break
                ws.append(_normalised_weight(lin, pre, noise).contiguous())
            if ws is not None:
                from .. import _lib
                noise = noise.contiguous()
                scale = torch.empty(B, C, dtype=torch.float32, device=x.device)
                shift = torch.empty_like(scale)
                _lib.check(_lib.lib().ps_noise_affine_f32(noise.data_ptr(), ws[0].data_ptr(), ws[1].data_ptr(), self.bn.stored_mean.data_ptr(),
                                                          self.bn.stored_var.data_ptr(), _ptr(pend), float(self.bn.eps), B, C, noise.size(1),
                                                          scale.data_ptr(), shift.data_ptr(), _stream()), "ps_noise_affine_f32")
                return scale, shift
        scale, shift = self.affine(x, noise)
        if pend is not None:
            shift = shift - pend.view(1, -1, 1, 1) * scale
        Bx = x.size(0)
        return scale.reshape(-1, C).expand(Bx, C).contiguous(), shift.reshape(-1, C).expand(Bx, C).contiguous()

    def forward(self, x, noise=None):
        scale, shift = self.affine(x, noise)
        return x * scale - shift


class _Slots(nn.Module):
    """Sub-modules registered under the indices the reference's nn.Sequential gave them (checkpoint key names)."""

    def __init__(self, **at):
        super().__init__()
        for idx, mod in at.items():
            self.add_module(idx.lstrip("_"), mod)

    def __getitem__(self, idx):
        return self._modules[str(idx)]


def _nhwc(module, x):
    """On the GPU: weights (once) and input to channels_last."""
    if not x.is_cuda:
        return x
    if not getattr(module, "_nhwc_ready", False):
        module.to(memory_format=torch.channels_last)
        module._nhwc_ready = True
    return x.contiguous(memory_format=torch.channels_last)


def _is_nhwc_cuda(*ts):
    """CUDA fp32 tensors in channels_last storage with C a multiple of 4: what csrc/nets.hip takes."""
    return all(t is not None and t.is_cuda and t.dtype == torch.float32 and t.dim() == 4 and t.size(1) % 4 == 0
               and t.is_contiguous(memory_format=torch.channels_last) for t in ts)


def _empty_nhwc(B, C, H, W, like):
    return torch.empty((B, C, H, W), dtype=like.dtype, device=like.device, memory_format=torch.channels_last)


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return None if t is None else t.data_ptr()


def _plain_conv_weight(conv, x):
    """The weight of `conv` as its forward would use it on `x`, for callers that take the convolution apart -- or None when that
    is not safe.  Only a plain Conv2d, or one whose single extra is the legacy torch.nn.utils.spectral_norm pre-hook (it recomputes
    .weight from weight_orig / u / v and returns nothing), is taken apart; anything else -- forward hooks, the parametrizations
    API, a pre-hook that edits the input, another padding mode -- goes through Module.__call__ untouched."""
    from torch.nn.utils.spectral_norm import SpectralNorm
    pre = list(conv._forward_pre_hooks.values())
    plain = (type(conv) is nn.Conv2d and not conv._forward_hooks and not getattr(conv, "parametrizations", None)
             and conv.padding_mode == "zeros" and all(isinstance(h, SpectralNorm) for h in pre))
    if not plain:
        return None
    return _normalised_weight(conv, pre, x)


def _normalised_weight(mod, pre, x):
    """mod.weight as mod's spectral-norm pre-hooks leave it.  In eval mode the hook does no power iteration -- weight_orig / sigma with
    sigma from the stored u, v is a constant of the checkpoint, yet torch recomputes it at every forward (a matrix-vector product, a
    dot and a division: ~270 launches of a few microseconds per decoder pass, a tenth of its time once the convolutions are
    fast).  Here it is computed once and kept until weight_orig / u / v change (their storage or version counters)."""
    if not pre:
        return mod.weight
    if mod.training or torch.is_grad_enabled():
        for hook in pre:
            hook(mod, (x,))
        return mod.weight
    key = tuple((t.data_ptr(), t._version) for t in (mod.weight_orig, mod.weight_u, mod.weight_v))
    cache = mod.__dict__.get("_ps_sn_cache")
    if cache is None or cache[0] != key:
        for hook in pre:
            hook(mod, (x,))
        cache = (key, mod.weight.detach())
        mod.__dict__["_ps_sn_cache"] = cache
    return cache[1]


def _sn_linear(lin, x):
    """lin(x) for the bias-free, possibly spectral-normalised Linear layers of LinearNoiseLayer, the normalised weight cached as above."""
    from torch.nn.utils.spectral_norm import SpectralNorm
    pre = list(lin._forward_pre_hooks.values())
    if (type(lin) is not nn.Linear or lin.bias is not None or lin._forward_hooks or getattr(lin, "parametrizations", None)
            or not all(isinstance(h, SpectralNorm) for h in pre)):
        return lin(x)
    return F.linear(x, _normalised_weight(lin, pre, x))


_MIOPEN_SAFE_BYTES = 2 ** 31


def _conv2d_batches(fn, x, out_channels, stride=1):
    """fn(x) -- a convolution through torch (MIOpen) -- with the batch cut so that neither the input nor the output of a call
    reaches 2 GiB: MIOpen's fp32 NHWC kernels index with 32 bits, and a (128, 128, 256, 256) activation -- 4 GiB, the decoder's
    widest layer at C5's 128 views -- comes back WRONG, silently (8e-2 of the output's scale against an fp64 convolution,
    tools/conv_f16x3_big_check.py; at 64 views, 2 GiB, it is right).  The split-fp16 kernel indexes frames with 64 bits."""
    B = x.size(0)
    per = max(x[0].numel(), out_channels * (x.size(2) // stride) * (x.size(3) // stride)) * x.element_size()
    n = max(1, min(B, (_MIOPEN_SAFE_BYTES - 1) // max(per, 1)))
    if not x.is_cuda or n >= B:
        return fn(x)
    return torch.cat([fn(x[i:i + n]) for i in range(0, B, n)])


def conv1x1(conv, x):
    """conv(x) WITHOUT the bias for a 1 x 1 convolution (stride 1, no padding) on a channels-last CUDA fp32 activation, through
    csrc/conv1x1.hip (fp32 matrix pipe: exact fp32 products) -- or None when `conv` / `x` are not that (the caller goes through torch)."""
    if not (type(conv) is nn.Conv2d and conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0)
            and conv.dilation == (1, 1) and conv.groups == 1 and _is_nhwc_cuda(x) and not torch.is_grad_enabled()):
        return None
    from .. import _lib
    L = _lib.lib()
    Ci, Co = conv.in_channels, conv.out_channels
    if not L.ps_conv1x1_takes(Ci, Co):
        return None
    weight = _plain_conv_weight(conv, x)
    if weight is None or weight.dtype != torch.float32 or not weight.is_cuda or weight.device != x.device:
        return None
    key = (weight.data_ptr(), weight._version, str(weight.device))
    cache = conv.__dict__.get("_ps_1x1_cache")
    if cache is None or cache[0] != key:
        cache = (key, weight.detach().reshape(Co, Ci).contiguous(), weight)       # (the weight is kept alive: its address is the key)
        conv.__dict__["_ps_1x1_cache"] = cache
    B, _, H, W = x.shape
    y = _empty_nhwc(B, Co, H, W, x)
    if Co == 1:   # (channels_last of one channel is ambiguous to torch; the kernel writes (B, H, W, Co))
        y = torch.empty((B, H, W, Co), dtype=x.dtype, device=x.device).permute(0, 3, 1, 2)
    _lib.check(L.ps_conv1x1_nhwc_f32(x.data_ptr(), cache[1].data_ptr(), B * H * W, Ci, Co, y.data_ptr(), _stream()), "ps_conv1x1_nhwc_f32")
    return y


def _conv_split(conv, x, mode=None):
    """conv(x) as (output WITHOUT the bias, bias) on the inference GPU path -- torch adds a convolution's bias in a pass
    of its own; here the per-channel constant rides along in whatever pass consumes the output (csrc/nets.hip).  Elsewhere
    (CPU, autograd, channel counts the kernels do not take): (conv(x), None).  mode "f16x3": a 1 x 1 convolution through
    csrc/conv1x1.hip instead of torch (MIOpen)."""
    if mode == "f16x3":
        y = conv1x1(conv, x)
        if y is not None:
            return y, conv.bias
    if conv.bias is None or torch.is_grad_enabled() or not _is_nhwc_cuda(x) or conv.out_channels % 4:
        return _conv2d_batches(conv, x, conv.out_channels, conv.stride[0]), None
    weight = _plain_conv_weight(conv, x)
    if weight is None:
        return _conv2d_batches(conv, x, conv.out_channels, conv.stride[0]), None
    return _conv2d_batches(lambda t: F.conv2d(t, weight, None, conv.stride, conv.padding, conv.dilation, conv.groups), x,
                           conv.out_channels, conv.stride[0]), conv.bias


# ---- the wide 3 x 3 convolutions on the fp16 matrix pipe (csrc/conv_f16x3.hip) ------------------------------------------------
DECODER_CONV = os.environ.get("PS_DECODER_CONV", "f16x3")   # "f16x3" | "fp32" (everything through torch / MIOpen)
_FORCED_CONV = []    # decoder_conv(mode) in effect
_overflow_flags = {}


def _conv_mode(opt):
    """Which convolutions a decoder block takes: decoder_conv(...) in effect, else opt.decoder_conv, else PS_DECODER_CONV."""
    return _FORCED_CONV[-1] if _FORCED_CONV else (getattr(opt, "decoder_conv", None) or DECODER_CONV)


class decoder_conv:
    """with decoder_conv("fp32"): ... -- every decoder convolution inside through torch (MIOpen fp32), whatever the options say: how the
    model reruns a pass whose split-fp16 convolutions met an activation beyond fp16's range."""

    def __init__(self, mode):
        if mode not in ("f16x3", "fp32"):
            raise ValueError("decoder_conv: 'f16x3' or 'fp32'")
        self.mode = mode

    def __enter__(self):
        _FORCED_CONV.append(self.mode)
        return self

    def __exit__(self, *exc):
        _FORCED_CONV.pop()
        return False


def _overflow_flag(device):
    key = str(device)
    if key not in _overflow_flags:
        _overflow_flags[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return _overflow_flags[key]


def clear_f16x3_overflow(device):
    """Asynchronous: forget what earlier passes left in the flag, so that the next check_f16x3_overflow speaks of the pass in between only."""
    flag = _overflow_flags.get(str(device))
    if flag is not None:
        flag.zero_()


def check_f16x3_overflow(device):
    """Synchronises.  Raises if a split-fp16 convolution met an activation beyond fp16's range since the last call (its output is
    then wrong); callers that can sit such activations out set PS_DECODER_CONV=fp32 / opt.decoder_conv = "fp32"."""
    flag = _overflow_flags.get(str(device))
    if flag is not None and int(flag.item()):
        flag.zero_()
        raise RuntimeError("refinement decoder: an activation beyond fp16's range (|v| > 65000, or not a number) reached a split-fp16 "
                           "convolution; rerun with PS_DECODER_CONV=fp32")


def _f16x3_takes(conv, x):
    return (conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1)
            and conv.groups == 1 and conv.in_channels % 32 == 0 and conv.out_channels % 64 == 0
            and _is_nhwc_cuda(x) and x.size(2) % 16 == 0 and x.size(3) % 16 == 0 and x.size(2) * x.size(3) * x.size(1) < 2 ** 31
            and not torch.is_grad_enabled())


def _thin_conv(conv, x, scale=None, shift=None):
    """conv(act(x)) WITHOUT the bias for the decoder's two thin 3 x 3 layers (4 -> Co, Ci -> <= 4 channels) through csrc/conv_thin.hip
    (fp32 FMAs), or None when `conv` is not one of them."""
    Ci, Co = conv.in_channels, conv.out_channels
    thin_in, thin_out = Ci == 4 and Co % 4 == 0 and Co <= 256, Ci % 32 == 0 and 1 <= Co <= 4
    if not (conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1
            and (thin_in or thin_out) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
            and x.is_contiguous(memory_format=torch.channels_last)
            and ((thin_in and Co == 64 and x.size(3) % 16 == 0) or (x.size(2) % 8 == 0 and x.size(3) % (64 if thin_in else 32) == 0))
            and not torch.is_grad_enabled()):
        return None
    weight = _plain_conv_weight(conv, x)
    if weight is None:
        return None
    from .. import _lib
    L = _lib.lib()
    key = (weight.data_ptr(), weight._version, str(weight.device))
    cache = conv.__dict__.get("_ps_thin_cache")
    if cache is None or cache[0] != key:
        cache = (key, weight.detach().permute(2, 3, 1, 0).contiguous(), weight)      # [ky][kx][ci][co]
        conv.__dict__["_ps_thin_cache"] = cache
    B, _, H, W = x.shape
    y = _empty_nhwc(B, Co, H, W, x)
    if Co == 1:   # (channels_last of one channel is ambiguous to torch; the kernel writes (B, H, W, Co))
        y = torch.empty((B, H, W, Co), dtype=x.dtype, device=x.device).permute(0, 3, 1, 2)
    if thin_in and Co == 64 and W % 16 == 0:   # on the fp16 matrix pipe (split operands): 82 instead of 270 us per 16 views
        _lib.check(L.ps_conv3x3_thin_in_f16x3_nhwc(x.data_ptr(), _ptr(scale), _ptr(shift), cache[1].data_ptr(), B, H, W, Co, y.data_ptr(),
                                                   _overflow_flag(x.device).data_ptr(), _stream()), "ps_conv3x3_thin_in_f16x3_nhwc")
    elif thin_in:
        _lib.check(L.ps_conv3x3_thin_in_nhwc_f32(x.data_ptr(), _ptr(scale), _ptr(shift), cache[1].data_ptr(), B, H, W, Co, y.data_ptr(),
                                                 _stream()), "ps_conv3x3_thin_in_nhwc_f32")
    else:
        _lib.check(L.ps_conv3x3_thin_out_nhwc_f32(x.data_ptr(), _ptr(scale), _ptr(shift), cache[1].data_ptr(), B, H, W, Ci, Co,
                                                  y.data_ptr(), _stream()), "ps_conv3x3_thin_out_nhwc_f32")
    return y


def _f16x3_conv(conv, x, scale=None, shift=None, bias=None, res=None):
    """conv(act(x)) WITHOUT the convolution's own bias through ps_conv3x3_f16x3_nhwc, act = max(x * scale - shift, 0) with scale /
    shift (B, C) contiguous, or the identity; `bias` (Co) and `res` (an NHWC tensor of the output's shape) are added on the way out.
    None when the kernel does not take this convolution (the caller then goes through torch)."""
    if not _f16x3_takes(conv, x):
        return None
    weight = _plain_conv_weight(conv, x)
    if weight is None:
        return None
    from .. import _lib
    L = _lib.lib()
    Co, Ci = conv.out_channels, conv.in_channels
    key = (weight.data_ptr(), weight._version, str(weight.device))
    cache = conv.__dict__.get("_ps_f16x3_cache")
    if cache is None or cache[0] != key:     # packed once per weight (with spectral norm in eval mode: per checkpoint, see _normalised_weight)
        wl = weight.detach().permute(0, 2, 3, 1).contiguous()          # (Co, 3, 3, Ci): no copy for a channels_last weight
        top = float(wl.abs().max())       # (synchronises -- once per weight) a weight fp16 cannot hold: this layer stays on torch
        packed = None
        if top == top and top < 6.0e4:
            packed = torch.empty(L.ps_conv3x3_f16x3_packed_bytes(Co, Ci), dtype=torch.uint8, device=x.device)
            _lib.check(L.ps_conv3x3_f16x3_pack(wl.data_ptr(), Co, Ci, packed.data_ptr(), _stream()), "ps_conv3x3_f16x3_pack")
        cache = (key, packed, weight)      # (the weight is kept alive: its address is the key)
        conv.__dict__["_ps_f16x3_cache"] = cache
    packed = cache[1]
    if packed is None:
        return None
    B, _, H, W = x.shape
    y = _empty_nhwc(B, Co, H, W, x)
    if res is not None and not (_is_nhwc_cuda(res) and res.shape == y.shape):
        return None
    _lib.check(L.ps_conv3x3_f16x3_nhwc(x.data_ptr(), _ptr(scale), _ptr(shift), packed.data_ptr(), _ptr(bias), _ptr(res), B, H, W, Ci, Co,
                                       y.data_ptr(), _overflow_flag(x.device).data_ptr(), _stream()), "ps_conv3x3_f16x3_nhwc")
    return y


def _sum_bias(*bs):
    bs = [b for b in bs if b is not None]
    return None if not bs else bs[0] if len(bs) == 1 else (bs[0] + bs[1]).contiguous()


def _resample_sum(kind, a, b, bias=None, post=None):
    """_resample(kind, a + bias) + _resample(kind, b) -- blocks.py:61-73 -- as ONE pass through csrc/nets.hip on the GPU
    (`bias`: per-channel constant still missing from the inputs, see _conv_split).  b = None: a alone -- resampling is linear, so
    a caller that already holds the SUM of the two branches resamples once.  post ('Down' only): a tensor of the output's shape added
    after the pooling."""
    from .. import _lib
    if (not (_is_nhwc_cuda(a) if b is None else _is_nhwc_cuda(a, b)) or (b is not None and a.shape != b.shape)
            or (kind and kind != "Up" and (a.size(2) % 2 or a.size(3) % 2)) or (b is None and not kind)):
        if bias is not None:
            a = a + bias.view(1, -1, 1, 1)
        out = _resample(kind, a) if b is None else _resample(kind, a) + _resample(kind, b)
        return out if post is None else out + post
    B, C, H, W = a.shape
    if not kind:
        out = torch.empty_like(a)
        _lib.check(_lib.lib().ps_add_bias_nhwc_f32(a.data_ptr(), b.data_ptr(), _ptr(bias), B, H * W, C, out.data_ptr(), _stream()),
                   "ps_add_bias_nhwc_f32")
    elif kind == "Up":
        out = _empty_nhwc(B, C, 2 * H, 2 * W, a)
        _lib.check(_lib.lib().ps_upsample_add_nhwc_f32(a.data_ptr(), _ptr(b), _ptr(bias), B, H, W, C, out.data_ptr(), _stream()),
                   "ps_upsample_add_nhwc_f32")
    else:
        out = _empty_nhwc(B, C, H // 2, W // 2, a)
        if post is not None and not (_is_nhwc_cuda(post) and post.shape == out.shape):
            raise ValueError("_resample_sum: post must be a channels_last tensor of the pooled shape")
        _lib.check(_lib.lib().ps_pool_add_post_nhwc_f32(a.data_ptr(), _ptr(b), _ptr(bias), _ptr(post), B, H, W, C, out.data_ptr(), _stream()),
                   "ps_pool_add_post_nhwc_f32")
        return out
    if post is not None:
        raise ValueError("_resample_sum: post goes with 'Down' only")
    return out


def _resample(kind, x):
    if kind == "Up":
        return F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
    if kind:            # "Down" or True
        return F.avg_pool2d(x, kernel_size=3, stride=2, padding=1)
    return x


class ResNet_Block(nn.Module):
    """(norm, relu, 3x3) x 2 (+ resample) on one branch, 1x1 (+ resample) or identity on the other (blocks.py:34-73)."""

    def __init__(self, in_c, in_o, opt, downsample=None):
        super().__init__()
        self.opt = opt
        self.resample = downsample
        self.ch_a = _Slots(_0=LinearNoiseLayer(opt, output_sz=in_c), _2=_conv(opt, in_c, in_o, 3, 1, 1),
                           _3=LinearNoiseLayer(opt, output_sz=in_o), _5=_conv(opt, in_o, in_o, 3, 1, 1))
        self.projected = bool(downsample) or in_c != in_o
        if self.projected:
            self.ch_b = _Slots(_0=_conv(opt, in_c, in_o, 1, 1, 0))

    @staticmethod
    def _noise_affine(layer, x, noise, bias=None, affine=None):
        """norm + ReLU of (x + bias): y = max(x * scale[b][c] - shift[b][c], 0), a pending conv bias folded into shift --
        one HIP pass on the GPU (ps_affine_relu_nhwc_f32)."""
        if affine is None and _is_nhwc_cuda(x) and not torch.is_grad_enabled() and hasattr(layer, "affine_bc"):
            affine = layer.affine_bc(x, noise, bias)
        if affine is not None:   # (bias already folded in; as (B, C) or broadcastable to (B, C, 1, 1))
            scale, shift = (t.view(t.size(0), -1, 1, 1) if t.dim() == 2 else t for t in affine)
        else:
            scale, shift = layer.affine(x, noise)
            if bias is not None:
                shift = shift - bias.view(1, -1, 1, 1) * scale
        B, C = x.size(0), x.size(1)
        if _is_nhwc_cuda(x) and scale.numel() in (C, B * C) and not torch.is_grad_enabled():
            from .. import _lib
            sc = scale.reshape(-1, C).expand(B, C).contiguous()
            sh = shift.reshape(-1, C).expand(B, C).contiguous()
            y = torch.empty_like(x)   # (preserves channels_last)
            _lib.check(_lib.lib().ps_affine_relu_nhwc_f32(x.data_ptr(), sc.data_ptr(), sh.data_ptr(), B, x.size(2) * x.size(3), C,
                                                          y.data_ptr(), _stream()), "ps_affine_relu_nhwc_f32")
            return y
        return torch.clamp_min(torch.addcmul(-shift, x, scale), 0)

    def _norm_relu_conv(self, layer, conv, x, noise, bias=None, res=None, out_bias=None):
        """conv(relu(norm(x + bias))) as (output without the convolution's own bias, that bias, fused).  The decoder's 3 x 3 layers:
        ONE kernel, norm + ReLU applied as the patch is staged (csrc/conv_f16x3.hip, conv_thin.hip); the others: the affine pass, then
        torch.  res / out_bias: the block's other branch and the biases still pending, which the split-fp16 kernel adds on its way out
        (`fused` says whether it did: the output is then conv + res + out_bias)."""
        mode = _conv_mode(self.opt)
        if mode == "f16x3" and conv.bias is not None and x.is_cuda and not torch.is_grad_enabled():
            scale, shift = layer.affine_bc(x, noise, bias)
            if _f16x3_takes(conv, x):
                y = _f16x3_conv(conv, x, scale, shift, out_bias, res)
                if y is not None:
                    return y, conv.bias, res is not None or out_bias is not None
                y = _f16x3_conv(conv, x, scale, shift)
            else:
                y = _thin_conv(conv, x, scale, shift)
            if y is not None:
                return y, conv.bias, False
            return _conv_split(conv, self._noise_affine(layer, x, None, None, affine=(scale, shift))) + (False,)
        return _conv_split(conv, self._noise_affine(layer, x, noise, bias)) + (False,)

    def forward(self, x, noise=(None, None)):
        a, ba, _ = self._norm_relu_conv(self.ch_a[0], self.ch_a[2], x, noise[0])
        mode = _conv_mode(self.opt)
        if (self.resample and self.resample != "Up" and mode == "f16x3" and _is_nhwc_cuda(x) and not torch.is_grad_enabled()
                and x.size(2) % 2 == 0 and x.size(3) % 2 == 0 and self.ch_b[0].kernel_size == (1, 1) and self.ch_b[0].stride == (1, 1)
                # (the 1 x 1 conv's bias must come back SEPARATE from _conv_split, to go through the pool's bias * inside / 9 term: a
                # convolution that cannot be taken apart returns conv(x) + bias, whose border pixels would then differ from the reference's)
                and (self.ch_b[0].bias is None or (self.ch_b[0].out_channels % 4 == 0 and _plain_conv_weight(self.ch_b[0], x) is not None))):
            # Down: avg_pool2d and the 1 x 1 convolution of the other branch commute -- pool first, convolve a quarter of the pixels,
            # and hand the result to the pooling of this branch as its `post` term (both biases ride through the pooling: bias=)
            b, bb = _conv_split(self.ch_b[0], _resample_sum(self.resample, x, None), mode)
            a, ba2, _ = self._norm_relu_conv(self.ch_a[3], self.ch_a[5], a, noise[1], ba)
            return _resample_sum(self.resample, a, None, _sum_bias(ba2, bb), post=b)
        b, bb = _conv_split(self.ch_b[0], x, mode) if self.projected else (x, None)
        # The second convolution adds the other branch on its way out where it can (resampling is linear: resample(a) + resample(b) =
        # resample(a + b)); without resampling the biases go in as well and its output is the block's.
        conv2 = self.ch_a[5]
        own = conv2.bias if conv2.bias is not None and not self.resample else None
        a, ba2, fused = self._norm_relu_conv(self.ch_a[3], conv2, a, noise[1], ba, res=b,
                                             out_bias=None if self.resample else _sum_bias(own, bb))
        if fused:
            return a if not self.resample else _resample_sum(self.resample, a, None, _sum_bias(ba2, bb))
        return _resample_sum(self.resample if self.projected else None, a, b, _sum_bias(ba2, bb))


class ResNetDecoder(nn.Module):
    """The refinement network (architectures.py:126-167): eight blocks, tanh, optional residual around them."""

    def __init__(self, opt, channels_in=64, channels_out=3, use_tanh=True):
        super().__init__()
        self.opt = opt
        setup = opt.refine_model_type.split("_")[1]
        if setup not in _DEC_FIRST:
            raise ValueError("refine_model_type %r: decoder table not built" % opt.refine_model_type)
        widths = [_DEC_FIRST[setup](channels_in)] + _DEC_WIDTHS(opt.ngf)
        self.eblocks = nn.ModuleList([ResNet_Block(widths[i], widths[i + 1], opt, _DEC_RESAMPLE[i]) for i in range(8)])
        if use_tanh:
            self.norm = nn.Tanh()

    def n_noise(self):
        return 2 * len(self.eblocks)

    def forward(self, x, background_mask=None, noise=None):
        if (background_mask is not None and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.size(1) == 3 and x.is_contiguous()
                and background_mask.dtype == torch.bool and background_mask.is_contiguous()
                and background_mask.shape == (x.size(0), x.size(2), x.size(3)) and not torch.is_grad_enabled()):
            from .. import _lib
            B, _, H, W = x.shape
            h = _empty_nhwc(B, 4, H, W, x)          # cat + NCHW -> NHWC in one pass
            _lib.check(_lib.lib().ps_cat_mask_nhwc_f32(x.data_ptr(), background_mask.data_ptr(), B, H, W, h.data_ptr(), _stream()),
                       "ps_cat_mask_nhwc_f32")
            _nhwc(self, h)                            # (the weights, once)
        else:
            h = x if background_mask is None else torch.cat((x, (~background_mask).unsqueeze(1).float()), 1)
            h = _nhwc(self, h)
        if noise is None and h.is_cuda:
            # the reference draws each layer's noise on the host and sends it over (normalization.py:36-37): sixteen small blocking
            # copies per pass.  The same sixteen draws, in the same order from the same generator, in ONE copy.
            noise = torch.stack([torch.randn(h.size(0), NOISE_SZ) for _ in range(self.n_noise())]).to(h.device).unbind(0)
        for i, blk in enumerate(self.eblocks):
            h = blk(h, (None, None) if noise is None else (noise[2 * i], noise[2 * i + 1]))
        if getattr(self.opt, "predict_residual", False):
            if background_mask is not None and getattr(self.opt, "normalize_before_residual", False):
                return (self.norm(h) + x).contiguous()
            return self.norm(h + x).contiguous()
        return self.norm(h).contiguous()


class Unet(nn.Module):
    """Depth regressor (architectures.py:174-279): 8 stride-2 4x4 convolutions down to 1x1, 8 x (bilinear x2, 3x3)
    back up with skip concatenations; leaky ReLU on the way down, ReLU on the way up, no norm on conv1 / conv8 / dconv8."""

    _ENC_NORM = [None, "batch_norm2_0", "batch_norm4_0", "batch_norm8_0", "batch_norm8_1", "batch_norm8_2", "batch_norm8_3", None]
    _DEC_NORM = ["batch_norm8_4", "batch_norm8_5", "batch_norm8_6", "batch_norm8_7", "batch_norm4_1", "batch_norm2_1", "batch_norm", None]

    def __init__(self, num_filters=32, channels_in=3, channels_out=3, use_tanh=False, use_3D=False, opt=None):
        super().__init__()
        if use_3D:
            raise NotImplementedError("3-D Unet is not part of the novel-view path")
        f = num_filters
        enc = [channels_in, f, 2 * f, 4 * f, 8 * f, 8 * f, 8 * f, 8 * f, 8 * f]
        dec_out = [8 * f, 8 * f, 8 * f, 8 * f, 4 * f, 2 * f, f, channels_out]
        norm = _norm_layer(opt)
        for i in range(8):
            setattr(self, f"conv{i + 1}", _conv(opt, enc[i], enc[i + 1], 4, 2, 1))
            if self._ENC_NORM[i]:
                setattr(self, self._ENC_NORM[i], norm(enc[i + 1]))
        for i in range(8):
            cin = enc[8] if i == 0 else dec_out[i - 1] + enc[8 - i]          # previous output ++ the mirrored encoder level
            setattr(self, f"dconv{i + 1}", _conv(opt, cin, dec_out[i], 3, 1, 1))
            if self._DEC_NORM[i]:
                setattr(self, self._DEC_NORM[i], norm(dec_out[i]))

    MAX_BATCH = 64   # images per call on the GPU, at most

    def _images_per_call(self, H, W):
        """dconv8 reads 2 f channels at the input's size: the call's batch keeps that activation below the 2 GiB at which MIOpen's fp32
        kernels start to index wrongly (_conv2d_batches) -- by bytes, so at 512 x 512 it is a quarter of what it is at 256 x 256."""
        per = self.dconv8.in_channels * H * W * 4
        return max(1, min(self.MAX_BATCH, (_MIOPEN_SAFE_BYTES - 1) // per))

    @staticmethod
    def _conv(conv, x):
        """conv(x); in eval mode without autograd with the spectral-normalised weight computed once per checkpoint (_normalised_weight)
        instead of at every forward -- torch's hook costs five small launches per layer, a fifth of this network's time at 8 images."""
        if x.is_cuda and not torch.is_grad_enabled() and not conv.training:
            weight = _plain_conv_weight(conv, x)
            if weight is not None:
                return F.conv2d(x, weight, conv.bias, conv.stride, conv.padding, conv.dilation, conv.groups)
        return conv(x)

    def forward(self, input):
        if input.is_cuda and not torch.is_grad_enabled() and not self.training:
            n = self._images_per_call(input.size(2), input.size(3))
            if input.size(0) > n:
                return torch.cat([self.forward(input[i:i + n]) for i in range(0, input.size(0), n)])
        skips, h = [], _nhwc(self, input)
        for i in range(8):
            h = self._conv(getattr(self, f"conv{i + 1}"), h if i == 0 else F.leaky_relu(h, 0.2))
            if self._ENC_NORM[i]:
                h = getattr(self, self._ENC_NORM[i])(h)
            skips.append(h)
        for i in range(8):
            if h.size(2) == 1 and h.size(3) == 1 and h.is_cuda:
                # bilinear x 2 of a single pixel is that pixel four times (torch takes its NCHW kernel for a 1 x 1 map: 0.26 ms of the
                # network's 1.7 per 8 images)
                h = F.relu(h).expand(-1, -1, 2, 2).contiguous(memory_format=torch.channels_last)
            else:
                h = F.interpolate(F.relu(h), scale_factor=2, mode="bilinear", align_corners=False)
            h = self._conv(getattr(self, f"dconv{i + 1}"), h)
            if self._DEC_NORM[i]:
                h = torch.cat((getattr(self, self._DEC_NORM[i])(h), skips[6 - i]), 1)
        return h.contiguous()


def get_decoder(opt):
    """models/networks/utilities.py:26-36 for the resnet decoders (the mask channel is there unless no_outpainting)."""
    if "resnet" not in opt.refine_model_type:
        raise ValueError("refine_model_type %r: only the resnet decoders are built" % opt.refine_model_type)
    return ResNetDecoder(opt, channels_in=3 + (0 if getattr(opt, "no_outpainting", False) else 1), channels_out=3)
