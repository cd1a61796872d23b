"""nin / gated_resnet / PONO with the reference's constructors and parameter names
(models/lmconv/layers.py:20-38, 136-163, 231-243), so that reference state_dicts load.  These torch modules only
serve the generic layer-by-layer path (any OurPixelCNN configuration, every masked conv on the HIP lmconv
kernel); PixelSynth's configuration runs fused inside csrc/lmconv.hip.  The reference file's other variants
(shifted convs, OrderRescale, masked_conv2d) are out of scope."""
import torch
import torch.nn as nn
from torch.nn.utils import weight_norm as wn

from .utils import concat_elu


def identity(x, *extra_args, **extra_kwargs):
    return x


class nin(nn.Module):
    """1x1 'network in network' layer: a (weight-normed) Linear applied over the channel axis.  Parameters live in
    `lin_a` (weight_g / weight_v / bias under weight norm), as in the reference."""

    def __init__(self, dim_in, dim_out, weight_norm=True):
        super(nin, self).__init__()
        linear = nn.Linear(dim_in, dim_out)
        self.lin_a = wn(linear) if weight_norm else linear
        self.dim_out = dim_out

    def forward(self, x):
        # channels last, Linear on the trailing axis, channels first again
        return self.lin_a(x.movedim(1, -1)).movedim(-1, 1)


class gated_resnet(nn.Module):
    """u -> u + PONO(p) * sigmoid(g), (p, g) = conv_out(celu(PONO(conv_input(celu(u))) [+ nin_skip(celu(a))]))."""

    def __init__(self, num_filters, conv_op, feature_norm_op=None, nonlinearity=concat_elu, skip_connection=0,
                 dropout_prob=0.5):
        super(gated_resnet, self).__init__()
        norm = (lambda: feature_norm_op(num_filters)) if feature_norm_op else (lambda: identity)
        wide = 2 * num_filters                                   # concat_elu doubles the channels
        self.skip_connection = skip_connection
        self.nonlinearity = nonlinearity
        self.conv_input = conv_op(wide, num_filters)
        self.norm_input = norm()
        if skip_connection != 0:
            self.nin_skip = nin(skip_connection * wide, num_filters)
        self.dropout = nn.Dropout2d(dropout_prob) if dropout_prob > 0.0 else identity
        self.conv_out = conv_op(wide, wide)
        self.norm_out = norm()

    def forward(self, og_x, a=None, mask=None):
        act = self.nonlinearity
        h = self.norm_input(self.conv_input(act(og_x), mask=mask), mask=mask)
        if a is not None:
            h = h + self.nin_skip(act(a))
        value, gate = self.conv_out(self.dropout(act(h)), mask=mask).chunk(2, dim=1)
        return og_x + self.norm_out(value, mask=mask) * torch.sigmoid(gate)


def pono(x, epsilon=1e-5):
    """Positional normalisation over the channel axis (unbiased variance): -> (normalised, mean, std)."""
    var, mean = torch.var_mean(x, dim=1, keepdim=True, unbiased=True)
    std = torch.sqrt(var + epsilon)
    return (x - mean) / std, mean, std


class PONO(nn.Module):
    def forward(self, x, mask=None):   # (the mask is accepted and ignored, like the reference's)
        return pono(x)[0]
