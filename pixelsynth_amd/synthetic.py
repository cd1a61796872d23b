"""Deterministic synthetic inputs for the novel-view hot path (numpy RandomState only, so the same
bytes are produced in the build container and on the GPU box, independent of torch's RNG).

Shapes/conventions follow the reference's drivers:
  demo cameras            demo.py:36-96            P = diag(2,-2,-1,1), K = I
  RealEstate10K cameras   data/realestate10k.py:59-74,120-140 (same offset matrix, K = I)
  Matterport cameras      data/create_rgb_dataset.py:204-216  K = diag(1/tan(hfov/2), .., 1, 1)
  depth range             options/options.py:69-70 (1..100 RealEstate), train_options.py:187-188 (0.5..10 MP3D)
"""
import math

import numpy as np


def image(seed, B=1, C=3, S=256):
    """Image / feature tensor U(-1,1), (B,C,S,S) f32 (the reference normalises RGB to [-1,1])."""
    rs = np.random.RandomState(seed)
    return (rs.rand(B, C, S, S).astype(np.float32) * 2.0 - 1.0).astype(np.float32)


def depth_uniform(seed, B=1, S=256, lo=1.0, hi=100.0):
    """Per-pixel depth U(lo,hi), (B,1,S,S) f32: the range sigmoid*(max_z-min_z)+min_z produces
    (models/z_buffermodel.py:304-308)."""
    rs = np.random.RandomState(seed)
    return (rs.rand(B, 1, S, S).astype(np.float32) * np.float32(hi - lo) + np.float32(lo)).astype(np.float32)


def depth_smooth(seed, B=1, S=256, lo=1.0, hi=100.0):
    """Low-frequency depth (realistic occlusion): lo + (hi-lo)*(0.5+0.5*sin(..)) with random phases."""
    rs = np.random.RandomState(seed)
    yy, xx = np.meshgrid(np.arange(S, dtype=np.float64), np.arange(S, dtype=np.float64), indexing="ij")
    out = np.empty((B, 1, S, S), np.float32)
    for b in range(B):
        fx, fy = rs.uniform(1.0, 3.0, 2)
        px, py = rs.uniform(0, 2 * math.pi, 2)
        w = 0.5 + 0.25 * np.sin(2 * math.pi * fx * xx / S + px) + 0.25 * np.sin(2 * math.pi * fy * yy / S + py)
        out[b, 0] = (lo + (hi - lo) * w).astype(np.float32)
    return out


def depth_from_image(img, lo=1.0, hi=100.0):
    """Deterministic stand-in for the depth Unet in chained (forward_scene) runs: a smooth depth in [lo, hi] computed
    from the frame itself (luminance, 16x16 box filter), torch in -> torch out, (B,3,S,S) -> (B,1,S,S)."""
    import torch.nn.functional as F
    lum = img.float().mean(1, keepdim=True)
    smooth = F.interpolate(F.avg_pool2d(lum, 16), size=img.shape[-2:], mode="bilinear", align_corners=False)
    return lo + (hi - lo) * (0.15 + 0.7 * (0.5 + 0.5 * smooth.clamp(-1, 1)))


def demo_cameras(B=1, ratio=1.0):
    """process_demo_data cameras (demo.py:36-96): dict of (B,4,4) f32 arrays P, Pinv, K, Kinv."""
    offset = np.array([[2, 0, -1], [0, -2, 1], [0, 0, -1]], dtype=np.float32)
    K = np.eye(4, dtype=np.float32)
    invK = np.linalg.inv(K)
    extrinsics = np.array([[1.0, 0, 0, 0], [0, 1.0, 0, 0], [0, 0, 1.0, 0]])
    intr = np.array([1.0, 1.0 * ratio, 0.5, 0.5])
    origK = np.array([[intr[0], 0, intr[2]], [0, intr[1], intr[3]], [0, 0, 1]], dtype=np.float32)
    Ktmp = np.matmul(offset, origK)
    P = np.matmul(Ktmp, extrinsics)
    P = np.vstack((P, np.zeros((1, 4), dtype=np.float32))).astype(np.float32)
    P[3, 3] = 1
    Pinv = np.linalg.inv(P)
    rep = lambda m: np.repeat(m[None].astype(np.float32), B, 0)
    return {"P": rep(P), "Pinv": rep(Pinv), "K": rep(K), "Kinv": rep(invK)}


def mp3d_cameras(B=1, hfov_deg=90.0):
    """Matterport/Habitat-shaped cameras (data/create_rgb_dataset.py:204-216), identity source pose."""
    hfov = hfov_deg * np.pi / 180.0
    K = np.array([[1.0 / np.tan(hfov / 2.0), 0, 0, 0], [0, 1.0 / np.tan(hfov / 2.0), 0, 0],
                  [0, 0, 1.0, 0], [0, 0, 0, 1.0]], dtype=np.float32)
    invK = np.linalg.inv(K)
    P = np.eye(4, dtype=np.float32)
    rep = lambda m: np.repeat(m[None].astype(np.float32), B, 0)
    return {"P": rep(P), "Pinv": rep(np.linalg.inv(P)), "K": rep(K), "Kinv": rep(invK)}


def euler_to_R(theta):
    """eulerAnglesToRotationMatrix (models/z_buffermodel.py:186-200): R = Rz @ Ry @ Rx, float64."""
    R_x = np.array([[1, 0, 0], [0, math.cos(theta[0]), -math.sin(theta[0])],
                    [0, math.sin(theta[0]), math.cos(theta[0])]])
    R_y = np.array([[math.cos(theta[1]), 0, math.sin(theta[1])], [0, 1, 0],
                    [-math.sin(theta[1]), 0, math.cos(theta[1])]])
    R_z = np.array([[math.cos(theta[2]), -math.sin(theta[2]), 0],
                    [math.sin(theta[2]), math.cos(theta[2]), 0], [0, 0, 1]])
    return np.dot(R_z, np.dot(R_y, R_x))


def yaw_pose(input_RT, yaw, pitch=0.0):
    """Target pose new_RT = M(euler (pitch,yaw,0)) @ input_RT and its inverse, as in
    get_rt_from_rot (models/z_buffermodel.py:229-242). input_RT (B,4,4) f32 -> (RTinv, RT)."""
    M = np.zeros((4, 4), np.float32)
    M[3, 3] = 1
    M[:3, :3] = euler_to_R(np.array([pitch, yaw, 0.0])).astype(np.float32)
    RT = np.matmul(M[None], input_RT).astype(np.float32)
    RTinv = np.linalg.inv(RT.astype(np.float64)).astype(np.float32)
    return RTinv, RT


def circle_pose(input_RT, num, denom):
    """Direction 'C' of get_rt_from_rot (models/z_buffermodel.py:217-225)."""
    rot = np.array([0.2 * np.cos(2 * np.pi * num / denom), 0.2 * np.sin(2 * np.pi * num / denom), 0])
    M = np.zeros((4, 4), np.float32)
    M[3, 3] = 1
    M[:3, :3] = euler_to_R(rot).astype(np.float32)
    RT = np.matmul(M[None], input_RT).astype(np.float32)
    RTinv = np.linalg.inv(RT.astype(np.float64)).astype(np.float32)
    return RTinv, RT


def codes(seed, B=1, H=32, W=32, num_classes=512):
    rs = np.random.RandomState(seed)
    return rs.randint(0, num_classes, size=(B, H, W)).astype(np.int64)


DOWN_NR = (2, 3, 3)


def pixelcnn_state_dict(seed=0):
    """Random weights with the reference OurPixelCNN's parameter names/shapes (PixelSynth config,
    models/z_buffermodel.py:62-74) and its default init *ranges* (kaiming_uniform(a=sqrt5) for
    lmconv: locally_masked_convolution.py:128-136; nn.Linear default + weight_norm g=||v||)."""
    rs = np.random.RandomState(seed)
    sd = {}

    def u(shape, bound):
        return ((rs.rand(*shape) * 2.0 - 1.0) * bound).astype(np.float32)

    def conv(name, co, ci):
        fan_in = ci * 9
        sd[name + ".weight"] = u((co, ci, 3, 3), math.sqrt(1.0 / fan_in))
        sd[name + ".bias"] = u((co,), 1.0 / math.sqrt(fan_in))

    def lin(name, ci, co):
        b = 1.0 / math.sqrt(ci)
        v = u((co, ci), b)
        sd[name + ".lin_a.bias"] = u((co,), b)
        sd[name + ".lin_a.weight_g"] = np.sqrt((v.astype(np.float64) ** 2).sum(1, keepdims=True)).astype(np.float32)
        sd[name + ".lin_a.weight_v"] = v

    for i in range(3):
        for j in range(DOWN_NR[i]):
            p = f"down_layers.{i}.u_stream.{j}."
            conv(p + "conv_input", 80, 160)
            lin(p + "nin_skip", 160, 80)
            conv(p + "conv_out", 160, 160)
    for i in range(3):
        for j in range(2):
            p = f"up_layers.{i}.u_stream.{j}."
            conv(p + "conv_input", 80, 160)
            conv(p + "conv_out", 160, 160)
    conv("u_init", 80, 513)
    for i in range(2):
        conv(f"downsize_u_stream.{i}", 80, 80)
    for i in range(2):
        conv(f"upsize_u_stream.{i}", 80, 80)
    lin("nin_out", 80, 512)
    return sd


def distance_maps():
    """>=20 integer (32,32) distance maps for custom_idx: half-planes, corners, islands, all-fg,
    all-bg, random with ties."""
    maps = []
    yy, xx = np.meshgrid(np.arange(32), np.arange(32), indexing="ij")
    maps.append(("halfplane_x", (16 - xx)))
    maps.append(("halfplane_y", (yy - 12)))
    maps.append(("halfplane_xr", (xx - 20)))
    maps.append(("corner", np.minimum(10 - xx, 10 - yy)))
    maps.append(("corner2", np.minimum(xx - 22, yy - 5)))
    maps.append(("island", 6 - np.maximum(np.abs(xx - 16), np.abs(yy - 16))))
    maps.append(("island_l1", 9 - (np.abs(xx - 8) + np.abs(yy - 20))))
    maps.append(("ring", np.abs(np.hypot(xx - 15.5, yy - 15.5).astype(int) - 9) - 3))
    maps.append(("all_fg", np.full((32, 32), 8191)))
    maps.append(("all_bg", np.full((32, 32), -8191)))
    maps.append(("zeros", np.zeros((32, 32), int)))
    maps.append(("diag", (xx - yy)))
    maps.append(("stripes", ((xx // 4) % 2) * 6 - 3))
    rs = np.random.RandomState(1234)
    for k in range(9):
        lo, hi = [(-3, 4), (-1, 2), (-20, 20), (0, 2), (-50, 50), (-2, 3), (-8, 1), (0, 5), (-5, 0)][k]
        maps.append((f"rand{k}", rs.randint(lo, hi, size=(32, 32))))
    return [(n, np.ascontiguousarray(m, dtype=np.int64)) for n, m in maps]


def background_masks(S=256):
    """A few (S,S) bool background masks: right half-plane, half-plane + foreground island,
    top band, none, all."""
    yy, xx = np.meshgrid(np.arange(S), np.arange(S), indexing="ij")
    out = {}
    out["right_half"] = xx >= S // 2
    isl = (xx >= (S * 5) // 16)
    isl &= ~((np.abs(xx - (S * 11) // 16) < S // 16) & (np.abs(yy - S // 2) < S // 16))
    out["half_plus_island"] = isl
    out["top_band"] = yy < S // 4
    out["none"] = np.zeros((S, S), bool)
    out["all"] = np.ones((S, S), bool)
    out["ragged"] = ((xx + (yy // 3) % 11) >= (S * 3) // 5)
    return out


# ------------------------------------------------------------------------------------------------
# VQ-VAE-2 top level (models/vqvae2/vqvae.py:VQVAETop with its default sizes): parameter / buffer names and shapes
# in the reference's state_dict order; conv weights (Co,Ci,k,k), transposed convs (Ci,Co,k,k).
# ------------------------------------------------------------------------------------------------
def _vq_res(prefix, ch=128, rc=32):
    return [(prefix + ".conv.1", (rc, ch, 3, 3), False), (prefix + ".conv.3", (ch, rc, 1, 1), False)]


VQVAE_LAYERS = (
    [("enc_b.blocks.0", (64, 3, 4, 4), False), ("enc_b.blocks.2", (128, 64, 4, 4), False), ("enc_b.blocks.4", (128, 128, 3, 3), False)]
    + _vq_res("enc_b.blocks.5") + _vq_res("enc_b.blocks.6")
    + [("enc_t.blocks.0", (64, 128, 4, 4), False), ("enc_t.blocks.2", (128, 64, 3, 3), False)]
    + _vq_res("enc_t.blocks.3") + _vq_res("enc_t.blocks.4")
    + [("quantize_conv_t", (64, 128, 1, 1), False), ("quantize_t", None, None), ("dec_t.blocks.0", (128, 64, 3, 3), False)]
    + _vq_res("dec_t.blocks.1") + _vq_res("dec_t.blocks.2")
    + [("dec_t.blocks.4", (128, 64, 4, 4), True), ("quantize_conv_b", (64, 192, 1, 1), False), ("quantize_b", None, None),
       ("upsample_t", (64, 64, 4, 4), True), ("dec.blocks.0", (128, 64, 3, 3), False)]
    + _vq_res("dec.blocks.1") + _vq_res("dec.blocks.2")
    + [("dec.blocks.4", (128, 64, 4, 4), True), ("dec.blocks.6", (64, 3, 4, 4), True)])


def vqvae_state_dict(seed=0, encoder_gain=2.0):
    """Random VQVAETop weights with the reference's names / shapes and torch's default init ranges
    (kaiming_uniform(a=sqrt5): bound 1/sqrt(fan_in)); codebooks ~ N(0,1) like the reference's buffers.
    The encoder weights are scaled by `encoder_gain` per layer: at the default init the latent of a random image
    is almost constant over the grid (spatial std 0.003 against a channel offset of 0.09) and every location would get
    the same code.  Use codebook_from_latents for a codebook that matches the latent distribution."""
    rs = np.random.RandomState(seed)
    sd = {}
    for name, shape, transposed in VQVAE_LAYERS:
        if shape is None:
            embed = rs.randn(64, 512).astype(np.float32)
            sd[name + ".embed"] = embed
            sd[name + ".cluster_size"] = np.zeros(512, np.float32)
            sd[name + ".embed_avg"] = embed.copy()
            continue
        fan_in = (shape[0] if transposed else shape[1]) * shape[2] * shape[3]
        bound = 1.0 / math.sqrt(fan_in)
        gain = encoder_gain if name.startswith(("enc_", "quantize_conv_t")) else 1.0
        sd[name + ".weight"] = ((rs.rand(*shape) * 2.0 - 1.0) * bound * gain).astype(np.float32)
        nb = shape[1] if transposed else shape[0]
        sd[name + ".bias"] = ((rs.rand(nb) * 2.0 - 1.0) * bound).astype(np.float32)
    return sd


def codebook_from_latents(lat, seed=0, n_embed=512, jitter=0.05):
    """A synthetic codebook that matches a latent distribution (what training's EMA update converges to): n_embed
    latent vectors of `lat` (B,D,H,W) picked at random, jittered by `jitter` x their spatial std -> (D, n_embed) f32."""
    rs = np.random.RandomState(seed)
    B, D, H, W = lat.shape
    flat = np.ascontiguousarray(lat.transpose(0, 2, 3, 1).reshape(-1, D))
    pick = flat[rs.randint(0, flat.shape[0], size=n_embed)]
    spread = float((flat - flat.mean(0, keepdims=True)).std())
    return np.ascontiguousarray((pick + rs.randn(n_embed, D) * jitter * spread).T.astype(np.float32))


# ------------------------------------------------------------------------------------------------
# Dense networks of SURVEY 8f row 2 (depth Unet, refinement decoder): synthetic weights for ANY module, by key name.
# ------------------------------------------------------------------------------------------------
def fill_state_dict(shapes, seed=0):
    """{key: shape} (a module's state_dict layout) -> {key: array}: deterministic per (seed, key), independent of key
    order, so the reference's module and its mirror get identical values from their own key lists.
    Weights ~ U(+-1/sqrt(fan_in)); spectral-norm vectors u, v = the leading singular pair of weight_orig (three power
    iterations), so the stored sigma is what training would have left there; normalisation statistics away from 0 / 1."""
    import zlib
    out = {}
    rs_of = lambda key: np.random.RandomState((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
    for key, shape in shapes.items():
        rs, leaf = rs_of(key), key.rsplit(".", 1)[-1]
        shape = tuple(shape)
        if leaf in ("weight_orig", "weight") and len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            out[key] = ((rs.rand(*shape) * 2 - 1) / math.sqrt(fan_in)).astype(np.float32)
        elif leaf in ("weight", "gain") and len(shape) == 1:
            out[key] = (0.8 + 0.4 * rs.rand(*shape)).astype(np.float32)
        elif leaf == "bias":
            out[key] = ((rs.rand(*shape) * 2 - 1) * 0.1).astype(np.float32)
        elif leaf in ("running_mean", "stored_mean"):
            out[key] = (rs.randn(*shape) * 0.1).astype(np.float32)
        elif leaf in ("running_var", "stored_var"):
            out[key] = (0.5 + rs.rand(*shape)).astype(np.float32)
        elif leaf == "num_batches_tracked":
            out[key] = np.zeros(shape, np.int64)
        elif leaf == "accumulation_counter":
            out[key] = np.zeros(shape, np.float32)
        elif leaf in ("weight_u", "weight_v"):
            continue
        else:
            raise KeyError("fill_state_dict: no rule for %r" % key)
    for key in shapes:
        if key.endswith(".weight_u"):
            base = key[:-len("weight_u")]
            W = out[base + "weight_orig"].reshape(shapes[base + "weight_orig"][0], -1).astype(np.float64)
            u = rs_of(key).randn(W.shape[0])
            for _ in range(3):
                v = W.T @ u
                v /= np.linalg.norm(v) + 1e-12
                u = W @ v
                u /= np.linalg.norm(u) + 1e-12
            out[key], out[base + "weight_v"] = u.astype(np.float32), v.astype(np.float32)
    return out


def resnet_state_dict(shapes, seed=0):
    """fill_state_dict for the ResNet-18 scene classifier with wider convolution / head weights (U(+-sqrt(3 / fan_in))): the
    activations keep their scale through the 18 layers (He-uniform's sqrt(6) saturates the softmax of this residual stack: the
    reference's own entropy formula then yields 0 * log 0), so that the logits -- and the entropy score built on them -- are not
    degenerate numbers a wrong layer could hide behind (logit std about 1.4, entropy about 4.3 of log 365 = 5.9)."""
    out = fill_state_dict(shapes, seed)
    for k, v in out.items():
        if k.endswith(".weight") and v.ndim >= 2:
            out[k] = (v * math.sqrt(3.0)).astype(np.float32)
    return out


def network_opts(**kw):
    """The generator options PixelSynth trains with (scripts/train_dpr_realestate.sh); an argparse.Namespace, which
    the reference tests with `"name" in opt`."""
    import argparse
    o = dict(norm_G="sync:spectral_batch", refine_model_type="resnet_256W8UpDown3", ngf=64, predict_residual=True,
             normalize_before_residual=False)
    o.update(kw)
    return argparse.Namespace(**o)
