"""Deterministic INPUT generators shared by make_golden.py (which feeds them to the reference) and
the tests (which feed them to the oracle / HIP path).  Keeping inputs as code keeps the committed
fixtures small: the .npz files mostly hold the reference's OUTPUTS."""
import numpy as np

LAYER_CASES = [  # name, Ci, Co, dilation, H
    ("u_init", 513, 80, 1, 8), ("conv_input", 160, 80, 1, 8), ("conv_out", 160, 160, 1, 8),
    ("dilated", 80, 80, 2, 8), ("conv_input_full", 160, 80, 1, 32), ("dilated_full", 80, 80, 2, 32),
    ("tiny_ragged", 5, 3, 1, 4), ("tiny_dil", 4, 6, 2, 5)]


def rand_mask(rs, B, L):
    """Random 0/1 masks (B,9,L): exercises every tap independently of any generation order."""
    return (rs.rand(B, 9, L) > 0.4).astype(np.float32)


def layer_case(name):
    idx = [c[0] for c in LAYER_CASES].index(name)
    _, ci, co, dil, H = LAYER_CASES[idx]
    rs = np.random.RandomState(2100 + idx)
    B = 2 if H <= 8 else 1
    L = H * H
    x = rs.randn(B, ci, H, H).astype(np.float32)
    w = (rs.randn(co, ci, 3, 3) / np.sqrt(ci * 9)).astype(np.float32)
    b = (rs.randn(co) * 0.1).astype(np.float32)
    m = rand_mask(rs, B, L)
    return dict(x=x, w=w, b=b, m=m, dil=dil, B=B, ci=ci, co=co, H=H)


def block_state(shapes, seed):
    """Random parameters for a module given {name: shape}; weight_g kept positive."""
    rs = np.random.RandomState(seed)
    sd = {}
    for k in sorted(shapes):
        shp = tuple(shapes[k])
        v = (rs.randn(*shp) / np.sqrt(max(1, int(np.prod(shp[1:]))))).astype(np.float32)
        if k.endswith("weight_g"):
            v = (np.abs(v) + 0.5).astype(np.float32)
        sd[k] = v
    return sd


GATED_SHAPES = {
    0: {"conv_input.weight": (80, 160, 3, 3), "conv_input.bias": (80,),
        "conv_out.weight": (160, 160, 3, 3), "conv_out.bias": (160,)},
    1: {"conv_input.weight": (80, 160, 3, 3), "conv_input.bias": (80,),
        "nin_skip.lin_a.bias": (80,), "nin_skip.lin_a.weight_g": (80, 1), "nin_skip.lin_a.weight_v": (80, 160),
        "conv_out.weight": (160, 160, 3, 3), "conv_out.bias": (160,)},
}


def gated_case(skip):
    rs = np.random.RandomState(3100 + skip)
    H, B = 8, 2
    x = rs.randn(B, 80, H, H).astype(np.float32)
    a = rs.randn(B, 80, H, H).astype(np.float32) if skip else None
    m = rand_mask(rs, B, H * H)
    m[:, 4] = 1  # type-B centre
    return dict(sd=block_state(GATED_SHAPES[skip], 3200 + skip), x=x, a=a, m=m, H=H, B=B)


NIN_SHAPES = {"lin_a.bias": (512,), "lin_a.weight_g": (512, 1), "lin_a.weight_v": (512, 80)}


def small_case():
    rs = np.random.RandomState(3300)
    x = (rs.randn(2, 80, 4, 4) * 3 + 1).astype(np.float32)
    return dict(x=x, nin_sd=block_state(NIN_SHAPES, 3301))
