"""concat_elu of the reference (models/lmconv/utils.py:31-35); everything else in that file is
training-time code (mixture-of-logistics losses, EMA) and out of scope for the inference hot path."""
import torch
import torch.nn.functional as F


def concat_elu(x):
    """like concatenated ReLU (http://arxiv.org/abs/1603.05201), but then with ELU"""
    axis = len(x.size()) - 3
    return F.elu(torch.cat([x, -x], dim=axis), inplace=True)
