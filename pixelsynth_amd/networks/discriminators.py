"""Multiscale PatchGAN discriminator of the sample ranking -- inference mirror of the reference's
models/networks/discriminators.py:78-216 (the SPADE / pix2pixHD discriminator PixelSynth trains with), same module tree and
state_dict keys (spectral-norm triples `weight_orig` / `weight_u` / `weight_v` included), so a reference checkpoint loads.

Inference form: the spectral normalisation is a fixed scale at test time (sigma = u . W v from the stored vectors, no power
iteration in eval mode) and is folded into the convolution weights once; instance normalisation (affine=False) and LeakyReLU
stay torch ops on MIOpen's convolutions -- this net scores a handful of candidates per view, it is not on the hot path."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class _SNConv(nn.Module):
    """Conv2d under spectral normalisation, eval-mode arithmetic of torch.nn.utils.spectral_norm: W = weight_orig / sigma with
    sigma = u . (W_mat v); parameters named like the hook's (the conv itself has no bias: a normalisation follows)."""

    def __init__(self, cin, cout, k, stride, pad):
        super().__init__()
        self.stride, self.pad = stride, pad
        self.weight_orig = nn.Parameter(torch.empty(cout, cin, k, k))
        self.register_buffer("weight_u", torch.empty(cout))
        self.register_buffer("weight_v", torch.empty(cin * k * k))
        nn.init.xavier_normal_(self.weight_orig, gain=0.02)
        with torch.no_grad():
            self.weight_u.copy_(F.normalize(torch.randn(cout), dim=0))
            self.weight_v.copy_(F.normalize(torch.randn(cin * k * k), dim=0))

    def weight(self):
        wm = self.weight_orig.reshape(self.weight_orig.size(0), -1)
        sigma = torch.dot(self.weight_u, torch.mv(wm, self.weight_v))
        return self.weight_orig / sigma

    def forward(self, x):
        return F.conv2d(x, self.weight(), None, self.stride, self.pad)


class NLayerDiscriminator(nn.Module):
    """discriminators.py:78-139 with n_layers_D = 4: model0 conv+LReLU, model1..3 SN-conv + InstanceNorm + LReLU (the last with
    stride 1), model4 conv to one channel.  forward -> the five intermediate outputs (or the last one with no_ganFeat_loss)."""

    def __init__(self, opt):
        super().__init__()
        opt.n_layers_D = 4
        self.opt = opt
        kw, padw = 4, int(np.ceil((4 - 1.0) / 2))
        nf = opt.ndf
        norm = getattr(opt, "norm_D", "spectralinstance")
        if norm != "spectralinstance":
            raise NotImplementedError("the inference mirror implements norm_D = spectralinstance (what PixelSynth trains with)")
        self.model0 = nn.Sequential(nn.Conv2d(opt.output_nc, nf, kernel_size=kw, stride=2, padding=padw), nn.LeakyReLU(0.2, False))
        for n in range(1, opt.n_layers_D):
            nf_prev, nf = nf, min(nf * 2, 512)
            stride = 1 if n == opt.n_layers_D - 1 else 2
            block = nn.Sequential(nn.Sequential(_SNConv(nf_prev, nf, kw, stride, padw), nn.InstanceNorm2d(nf, affine=False)),
                                  nn.LeakyReLU(0.2, False))
            self.add_module("model" + str(n), block)
        self.add_module("model" + str(opt.n_layers_D), nn.Sequential(nn.Conv2d(nf, 1, kernel_size=kw, stride=1, padding=padw)))

    def forward(self, input):
        results = [input]
        for submodel in self.children():
            results.append(submodel(results[-1]))
        return results[1:] if not self.opt.no_ganFeat_loss else results[-1]


class MultiscaleDiscriminator(nn.Module):
    """discriminators.py:142-208: num_D = 2 NLayerDiscriminators, the second on the 3x3 / stride-2 average-pooled image."""

    def __init__(self, opt):
        super().__init__()
        opt.netD_subarch, opt.num_D = "n_layer", 2
        self.opt = opt
        for i in range(opt.num_D):
            self.add_module("discriminator_%d" % i, NLayerDiscriminator(opt))

    def downsample(self, input):
        return F.avg_pool2d(input, kernel_size=3, stride=2, padding=[1, 1], count_include_pad=False)

    def forward(self, input):
        result = []
        for _, D in self.named_children():
            out = D(input)
            result.append(out if not self.opt.no_ganFeat_loss else [out])
            input = self.downsample(input)
        return result


def define_D(opt):
    return MultiscaleDiscriminator(opt)
