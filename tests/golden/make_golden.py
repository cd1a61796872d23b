"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference (read-only) in the
build container.  Nothing from /root/reference is copied: the fixtures hold inputs (or the seeds of
pixelsynth_amd/synthetic.py generators) and the reference's OUTPUTS only.

Shims needed to import the reference here (none of them is copied into the repo):
  * pytorch3d is absent -> empty stub modules (project_pts* are pure torch and never touch it);
  * models/lmconv/get_custom_order.so is cpython-37m -> the reference's own Cython C is compiled by
    oracle/build_oracle.py into oracle/_ref/ and registered under the expected module name;
  * no GPU here -> torch.Tensor.cuda is patched to identity and np.int (removed in numpy>=1.24)
    is aliased to int, so that models/lmconv/sample.py:sample() itself can run on CPU.

Run:  python tests/golden/make_golden.py        (about 2 minutes)
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from oracle import build_oracle  # noqa: E402
from pixelsynth_amd import synthetic as syn  # noqa: E402
sys.path.insert(0, HERE)
import cases  # noqa: E402


def _import_reference():
    ref_so = build_oracle.build_ref()
    spec = importlib.util.spec_from_file_location("get_custom_order", ref_so)
    gco = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gco)
    sys.modules["models.lmconv.get_custom_order"] = gco
    for name in ["pytorch3d", "pytorch3d.structures", "pytorch3d.renderer", "pytorch3d.renderer.points"]:
        sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision"] = types.ModuleType("torchvision")     # architectures.py only touches it inside VGG19
    sys.modules["pytorch3d.structures"].Pointclouds = object
    sys.modules["pytorch3d.renderer"].compositing = object
    sys.modules["pytorch3d.renderer.points"].rasterize_points = object
    torch.Tensor.cuda = lambda self, *a, **k: self
    np.int = int
    import models.lmconv
    models.lmconv.get_custom_order = gco
    import models.lmconv.masking as masking
    from models.lmconv.model import OurPixelCNN
    from models.lmconv.layers import PONO, gated_resnet, nin
    from models.lmconv.locally_masked_convolution import locally_masked_conv2d
    from models.lmconv.sample import sample
    from models.lmconv.utils import concat_elu
    from models.projection.z_buffer_manipulator import PtsManipulator
    return dict(gco=gco, masking=masking, OurPixelCNN=OurPixelCNN, PONO=PONO, gated_resnet=gated_resnet,
                nin=nin, lmconv=locally_masked_conv2d, sample=sample, concat_elu=concat_elu,
                PtsManipulator=PtsManipulator)


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def make_net(R, seed):
    net = R["OurPixelCNN"](nr_resnet=2, nr_filters=80, input_channels=512, nr_logistic_mix=10,
                           kernel_size=(3, 3), max_dilation=2, weight_norm=False,
                           feature_norm_op=lambda c: R["PONO"](), dropout_prob=0, conv_bias=True,
                           conv_mask_weight=False, rematerialize=False, binarize=False).eval()
    sd = {k: t(v) for k, v in syn.pixelcnn_state_dict(seed).items()}
    missing = net.load_state_dict(sd, strict=True)
    return net


def poses():
    """(name, cams, RT2, RT2inv, depth range) used for the projection fixtures."""
    out = []
    cam = syn.demo_cameras(1)
    for name, yaw in [("demo_L", -0.6), ("demo_R", 0.6), ("demo_small", 0.05), ("demo_identity", 0.0)]:
        RTinv, RT = syn.yaw_pose(cam["P"], yaw)
        out.append((name, cam, RT, RTinv, (1.0, 100.0)))
    RTinv, RT = syn.circle_pose(cam["P"], 5, 64)
    out.append(("demo_circle5", cam, RT, RTinv, (1.0, 100.0)))
    cam = syn.mp3d_cameras(1)
    for name, yaw in [("mp3d_yaw", 0.4), ("mp3d_back", 2.8)]:
        RTinv, RT = syn.yaw_pose(cam["P"], yaw, pitch=0.1)
        out.append((name, cam, RT, RTinv, (0.5, 10.0)))
    return out


def gen_projection(R):
    ns = types.SimpleNamespace(splatter="xyblending", learn_default_feature=True, radius=4, pp_pixel=128)
    fx = {}
    for W in (8, 16):
        pm = R["PtsManipulator"](W, C=3, opt=ns)
        fx[f"xyzs_W{W}"] = pm.xyzs.numpy().copy()
    pm256 = R["PtsManipulator"](256, C=3, opt=ns)
    fx["xyzs_W256_stride97"] = pm256.xyzs.numpy()[:, :, ::97].copy()
    names = []
    for name, cam, RT2, RT2inv, (lo, hi) in poses():
        names.append(name)
        for W, stride in ((16, 1), (256, 61)):
            pm = pm256 if W == 256 else R["PtsManipulator"](W, C=3, opt=ns)
            d = syn.depth_uniform(7, 2, W, lo, hi)
            B = d.shape[0]
            rep = lambda m: t(np.repeat(m, B, 0))
            with torch.no_grad():
                s = pm.project_pts(t(d).view(B, 1, -1), rep(cam["K"]), rep(cam["Kinv"]), rep(cam["P"]),
                                   rep(cam["Pinv"]), rep(RT2), rep(RT2inv))
            fx[f"proj_{name}_W{W}"] = s.numpy()[:, :, ::stride].copy()
        fx[f"pose_{name}_RT2"] = RT2
        fx[f"pose_{name}_RT2inv"] = RT2inv
    fx["pose_names"] = np.array(names)
    # near-zero z: points whose projected |z| < EPS must become (-10, +10, +10)
    cam = syn.demo_cameras(1)
    RTinv, RT = syn.yaw_pose(cam["P"], 1.5)
    pm = R["PtsManipulator"](16, C=3, opt=ns)
    d = syn.depth_uniform(3, 1, 16, 0.001, 0.02)
    s = pm.project_pts(t(d).view(1, 1, -1), t(cam["K"]), t(cam["Kinv"]), t(cam["P"]), t(cam["Pinv"]),
                       t(RT), t(RTinv))
    fx["proj_epscase_depth"] = d
    fx["proj_epscase_RT2"] = RT
    fx["proj_epscase_out"] = s.numpy()
    # cumulative (scene mode): W=16, half of the pixels "new", a prior cloud of 100 points
    W = 16
    pm = R["PtsManipulator"](W, C=3, opt=ns)
    rs = np.random.RandomState(11)
    last_bg = (rs.rand(1, W * W) > 0.5)
    n_new = int(last_bg.sum())
    d_new = (rs.rand(1, 1, n_new).astype(np.float32) * 9 + 1)
    prior = rs.randn(1, 4, 100).astype(np.float32) * 2
    prior[:, 2] = -np.abs(prior[:, 2]) - 0.5
    prior[:, 3] = 1
    RT3inv, _ = syn.yaw_pose(cam["P"], 0.1)
    RTinv2, RT2 = syn.yaw_pose(cam["P"], 0.3)
    # force the EPS branch on 3 prior points: choose p with (RT2 @ RT3inv @ p).z ~ 1e-3
    M = RT2[0].astype(np.float64) @ RT3inv[0].astype(np.float64)
    for k in range(3):
        prior[0, :, k] = (np.linalg.inv(M) @ np.array([0.3 * k, -0.2, 0.001 * (k - 1), 1.0])).astype(np.float32)
    with torch.no_grad():
        s, cloud = pm.project_pts_cumulative(t(d_new), t(cam["K"]), t(cam["Kinv"]), t(cam["P"]), t(cam["Pinv"]),
                                             t(RT2), t(RTinv2), t(prior), t(last_bg).view(1, 1, -1), t(RT3inv))
    fx.update(cum_last_bg=last_bg, cum_depth_new=d_new, cum_prior=prior, cum_RT2=RT2, cum_RT3inv=RT3inv,
              cum_sampler=s.numpy(), cum_cloud=cloud.numpy())
    # cumulative without prior / mask (first frame of a scene)
    d0 = syn.depth_uniform(5, 1, W, 1, 10)
    with torch.no_grad():
        s0, cloud0 = pm.project_pts_cumulative(t(d0).view(1, 1, -1), t(cam["K"]), t(cam["Kinv"]), t(cam["P"]),
                                               t(cam["Pinv"]), t(RT2), t(RTinv2), None, None, None)
    fx.update(cum0_depth=d0, cum0_sampler=s0.numpy(), cum0_cloud=cloud0.numpy())
    np.savez_compressed(os.path.join(HERE, "projection.npz"), **fx)
    print("projection.npz", len(fx))


def gen_orders_masks(R):
    fx = {}
    names = []
    for name, D in syn.distance_maps():
        names.append(name)
        d = D.copy()
        order = R["masking"].get_generation_order_idx("custom", 32, 32, d, (16, 16))
        fx[f"order_{name}"] = np.asarray(order).astype(np.int16)
        assert (d == D * 10000).all()  # the reference mutates its argument
    fx["names"] = np.array(names)
    for name in ("halfplane_x", "island", "rand0", "all_fg", "ring", "rand4"):
        order = fx[f"order_{name}"].astype(np.int64)
        for tag, dil, typ in (("A1", 1, "A"), ("B1", 1, "B"), ("B2", 2, "B")):
            m = R["masking"].get_unfolded_masks(order, 32, 32, k=3, dilation=dil, mask_type=typ)
            fx[f"mask_{name}_{tag}"] = np.packbits(m.numpy()[0].astype(np.uint8), axis=1)
    # small non-32 grid (8x8) as an edge case of the mask builder
    rs = np.random.RandomState(5)
    D8 = rs.randint(-3, 4, size=(8, 8)).astype(np.int64)
    o8 = R["masking"].get_generation_order_idx("custom", 8, 8, D8.copy(), (4, 4))
    fx["order8_D"] = D8
    fx["order8"] = np.asarray(o8).astype(np.int16)
    for tag, dil, typ in (("A1", 1, "A"), ("B1", 1, "B"), ("B2", 2, "B")):
        fx[f"mask8_{tag}"] = R["masking"].get_unfolded_masks(o8, 8, 8, k=3, dilation=dil, mask_type=typ).numpy()
    np.savez_compressed(os.path.join(HERE, "orders_masks.npz"), **fx)
    print("orders_masks.npz", len(fx))


def gen_lmconv_layers(R):
    fx = {}
    for name, ci, co, dil, H in cases.LAYER_CASES:
        c = cases.layer_case(name)
        layer = R["lmconv"](ci, co, kernel_size=(3, 3), dilation=dil, bias=True)
        with torch.no_grad():
            layer.weight.copy_(t(c["w"]))
            layer.bias.copy_(t(c["b"]))
            mrep = t(c["m"]).unsqueeze(1).repeat(1, ci, 1, 1).view(c["B"] * ci, 9, H * H)
            y = layer(t(c["x"]), mrep)
        fx[f"{name}_y"] = y.numpy()
    np.savez_compressed(os.path.join(HERE, "lmconv_layers.npz"), **fx)
    print("lmconv_layers.npz", len(fx))


def gen_blocks(R):
    """gated_resnet (with and without skip), PONO, nin, concat_elu."""
    fx = {}
    conv_op = lambda cin, cout: R["lmconv"](cin, cout, kernel_size=(3, 3), bias=True, mask_weight=False)
    for skip in (0, 1):
        c = cases.gated_case(skip)
        blk = R["gated_resnet"](80, conv_op, lambda ch: R["PONO"](), R["concat_elu"], skip_connection=skip,
                                dropout_prob=0).eval()
        blk.load_state_dict({k: t(v) for k, v in c["sd"].items()}, strict=True)
        L = c["H"] * c["H"]
        with torch.no_grad():
            mrep = t(c["m"]).unsqueeze(1).repeat(1, 160, 1, 1).view(c["B"] * 160, 9, L)
            y = blk(t(c["x"]), a=None if c["a"] is None else t(c["a"]), mask=mrep)
        fx[f"gr{skip}_y"] = y.numpy()
    c = cases.small_case()
    fx["pono_y"] = R["PONO"]()(t(c["x"])).numpy()
    fx["celu_y"] = R["concat_elu"](t(c["x"])).numpy()
    lin = R["nin"](80, 512).eval()
    lin.load_state_dict({k: t(v) for k, v in c["nin_sd"].items()}, strict=True)
    with torch.no_grad():
        fx["nin_y"] = lin(t(c["x"])).numpy()
    np.savez_compressed(os.path.join(HERE, "blocks.npz"), **fx)
    print("blocks.npz", len(fx))


def masks_from_order(R, order):
    mi = R["masking"].get_unfolded_masks(order, 32, 32, k=3, dilation=1, mask_type="A")
    mu = R["masking"].get_unfolded_masks(order, 32, 32, k=3, dilation=1, mask_type="B")
    md = R["masking"].get_unfolded_masks(order, 32, 32, k=3, dilation=2, mask_type="B")
    rep = lambda m, c: m[0:1].repeat(c, 1, 1).view(-1, 9, 1024)
    return rep(mi, 513), rep(mu, 160), rep(md, 80)


def gen_network(R):
    """Full OurPixelCNN logits for 2 weight seeds x orders; subset of positions stored."""
    fx = {}
    dmaps = dict(syn.distance_maps())
    pos = np.random.RandomState(3).permutation(1024)[:48]
    fx["positions"] = pos
    for wi, (wseed, oname) in enumerate([(0, "halfplane_x"), (1, "rand0")]):
        net = make_net(R, wseed)
        order = R["masking"].get_generation_order_idx("custom", 32, 32, dmaps[oname].copy(), (16, 16))
        masks = masks_from_order(R, order)
        codes = syn.codes(100 + wi, 1)
        x = torch.nn.functional.one_hot(t(codes), 512).permute(0, 3, 1, 2).float()
        with torch.no_grad():
            logits = net([x, *masks], sample=True)
        lg = logits[0].reshape(512, 1024).numpy()
        fx[f"net{wi}_wseed"] = np.array(wseed)
        fx[f"net{wi}_order_name"] = np.array(oname)
        fx[f"net{wi}_codes_seed"] = np.array(100 + wi)
        fx[f"net{wi}_logits_sub"] = lg[:, pos].copy()
        fx[f"net{wi}_logits_sum"] = lg.astype(np.float64).sum(0)  # per-position checksum
    np.savez_compressed(os.path.join(HERE, "network.npz"), **fx)
    print("network.npz", len(fx))


def gen_ar_trace(R):
    """The reference's own sample() (models/lmconv/sample.py:8-73), run on CPU through the shims.

    Background: right part of the grid + a foreground island whose order indices fall after the first
    sampled index.  seed=1 -> exercises the 'seed extra draws' loop.  Stores the final one-hot argmax,
    and (teacher-forcing check) the logits of ONE full forward on the completed grid at the sampled
    positions -- asserted here to be bit-equal to the per-step logits the loop saw."""
    fx = {}
    net = make_net(R, 0)
    bg32 = np.zeros((32, 32), np.float32)
    bg32[:, 22:] = 1
    bg32[12:15, 26:29] = 0  # foreground island inside the background
    fgb = (bg32 == 0).astype(np.uint8)
    sys.path.insert(0, ROOT)
    from oracle import c_oracle
    D = c_oracle.signed_distance(fgb, (bg32 == 1).astype(np.uint8))
    order = R["masking"].get_generation_order_idx("custom", 32, 32, D.copy(), (16, 16))
    masks = masks_from_order(R, order)
    codes = syn.codes(200, 1)
    args = types.SimpleNamespace(num_classes=512, dataloader_seed=0)
    seen = []
    orig_forward = net.forward

    def spy(x, sample=False, **kw):
        out = orig_forward(x, sample=sample, **kw)
        seen.append(out.detach().clone())
        return out
    net.forward = spy
    with torch.no_grad():
        data, loss = R["sample"](net, [order], *masks, t(codes), [3, 32, 32], args, seed=1, temperature=0.7,
                                 background_mask=t(bg32)[None])
    net.forward = orig_forward
    final = data.argmax(1)[0].numpy()
    region = [(int(i), int(j)) for i, j in order if bg32[i, j] == 1]
    assert len(seen) == len(region)
    with torch.no_grad():
        full = net([data, *masks], sample=True)
    step_logits = np.stack([seen[n][0, :, i, j].numpy() for n, (i, j) in enumerate(region)])
    full_logits = np.stack([full[0, :, i, j].numpy() for (i, j) in region])
    causal_maxdiff = float(np.abs(step_logits - full_logits).max())
    print("AR trace: steps", len(region), "max|step - full| =", causal_maxdiff)
    fx.update(D=D, order=np.asarray(order).astype(np.int16), bg32=bg32, codes_seed=np.array(200),
              wseed=np.array(0), seed=np.array(1), temperature=np.array(0.7), final_codes=final.astype(np.int16),
              step_logits=step_logits[::4].astype(np.float32), causal_maxdiff=np.array(causal_maxdiff),
              n_steps=np.array(len(region)), loss=np.array(float(loss)))
    np.savez_compressed(os.path.join(HERE, "ar_trace.npz"), **fx)
    print("ar_trace.npz", len(fx))


def gen_vqvae(R):
    """VQ-VAE-2 top level: the reference's own VQVAETop (models/vqvae2/vqvae.py:229-311) with the
    synthetic weights of pixelsynth_amd.synthetic.vqvae_state_dict(0) and a codebook calibrated on image A
    (codebook_from_latents); encode image B -> top codes, decode_code(codes) -> image.  Stored: the codebook, the
    pre-quantisation latent of B, the codes, the two smallest distances per location (tie margin), the decoded
    image on a 4x4-subsampled grid, and the state_dict key / shape list (strict load = name compatibility)."""
    from models.vqvae2.vqvae import VQVAETop
    sd = {k: t(v) for k, v in syn.vqvae_state_dict(0).items()}
    ref = VQVAETop().eval()
    assert list(ref.state_dict().keys()) == list(sd.keys())
    ref.load_state_dict(sd, strict=True)
    imgA, imgB = t(syn.image(11, 1, 3, 256)), t(syn.image(7, 1, 3, 256))
    with torch.no_grad():
        latA = ref.quantize_conv_t(ref.enc_t(ref.enc_b(imgA.clone())))
        embed = syn.codebook_from_latents(latA.numpy(), 0)
        ref.quantize_t.embed.copy_(t(embed))
        latB = ref.quantize_conv_t(ref.enc_t(ref.enc_b(imgB.clone())))
        _, _, _, id_t, _ = ref.encode(imgB.clone())
        flat = latB.permute(0, 2, 3, 1).reshape(-1, 64)
        dist = flat.pow(2).sum(1, keepdim=True) - 2 * flat @ ref.quantize_t.embed + ref.quantize_t.embed.pow(2).sum(0, keepdim=True)
        two = dist.sort(1)[0][:, :2]
        dec = ref.decode_code(id_t)
    keys = np.array([f"{k}:{','.join(map(str, v.shape))}" for k, v in ref.state_dict().items()])
    np.savez_compressed(os.path.join(HERE, "vqvae.npz"), embed=embed, latB=latB.numpy(), codes=id_t.numpy().astype(np.int32),
                        two_smallest=two.numpy(), dec_sub=dec.numpy()[:, :, ::4, ::4], keys=keys,
                        image_seeds=np.array([11, 7]))
    print("vqvae: unique codes", int(id_t.unique().numel()), "min gap", float((two[:, 1] - two[:, 0]).min()))


def gen_networks(R):
    """Depth Unet and refinement ResNetDecoder: the reference's own modules
    (models/networks/architectures.py:126-279, built by models/networks/utilities.py:get_decoder) with the options
    PixelSynth trains with, in eval mode, filled by pixelsynth_amd.synthetic.fill_state_dict from their own key lists.
    The decoder's noise draws (torch.randn inside LinearNoiseLayer, normalization.py:39) are replaced for the run by a
    recorded sequence.  Stored: outputs on a 2x2 / 4x4 subsampled grid, the noise, key/shape lists."""
    from models.networks.architectures import Unet
    from models.networks.utilities import get_decoder
    opt = syn.network_opts()
    unet = Unet(channels_in=3, channels_out=1, opt=opt).eval()
    dec = get_decoder(opt).eval()
    out = {}
    for name, mod in (("unet", unet), ("decoder", dec)):
        shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
        mod.load_state_dict({k: t(v) for k, v in syn.fill_state_dict(shapes, 5).items()}, strict=True)
        out[name + "_keys"] = np.array([f"{k}:{','.join(map(str, v))}" for k, v in shapes.items()])
    img = t(syn.image(41, 1, 3, 256))
    x = t(syn.image(42, 1, 3, 256))
    bgm = torch.from_numpy(syn.background_masks(256)["ragged"])[None]
    noise = np.random.RandomState(43).randn(16, 1, 20).astype(np.float32)
    draws = iter(noise)
    real_randn = torch.randn
    torch.randn = lambda *a, **k: t(next(draws))
    try:
        with torch.no_grad():
            depth = unet(img)
            refined = dec(x, bgm)
            opt.predict_residual = False
            draws = iter(noise)
            plain = dec(x, bgm)
    finally:
        torch.randn = real_randn
    np.savez_compressed(os.path.join(HERE, "networks.npz"), unet_out_sub=depth.numpy()[:, :, ::2, ::2],
                        decoder_out_sub=refined.numpy()[:, :, ::4, ::4], decoder_plain_sub=plain.numpy()[:, :, ::4, ::4],
                        noise=noise, image_seeds=np.array([41, 42]), weight_seed=np.array(5), **out)
    print("networks: unet out", float(depth.mean()), float(depth.std()), "decoder out", float(refined.mean()), float(refined.std()))


def _import_reference_model():
    """models/z_buffermodel.py itself (SURVEY 8c lists it as not importable: torchvision, cv2, mock are absent).  With
    attribute-only stand-ins for those three -- none of their code is ever reached by what is called below -- the
    reference's ZbufferModelPts constructs on CPU, so get_rt_from_rot, get_combined and the frame loop of forward_scene
    run as written."""
    tv = sys.modules["torchvision"]
    tr = types.ModuleType("torchvision.transforms")

    class _T:
        def __init__(self, *a, **k):
            pass
    for n in ("Compose", "Resize", "CenterCrop", "ToTensor", "Normalize"):
        setattr(tr, n, _T)
    tv.transforms = tr
    sys.modules["torchvision.transforms"] = tr
    tm = types.ModuleType("torchvision.models")
    tm.__dict__["resnet18"] = lambda num_classes: torch.nn.Identity()
    tv.models = tm
    sys.modules["torchvision.models"] = tm
    mock = types.ModuleType("mock")
    mock.Mock = type("Mock", (), {})
    sys.modules["mock"] = mock
    sys.modules["cv2"] = types.ModuleType("cv2")
    import models.z_buffermodel as zb
    return zb


def _reference_model(zb, **kw):
    opt = syn.network_opts()
    o = dict(W=256, use_rgb_features=True, depth_predictor_type="unet", seed=0, max_z=100.0, min_z=1.0, voxel_size=64,
             model_setting="gen_img", rotation=0.6, homography=False, losses=["1.0_l1"], discriminator_losses=None,
             splatter="xyblending", learn_default_feature=True, radius=4, pp_pixel=128, tau=1.0, rad_pow=2,
             accumulation="alphacomposite")
    o.update(kw)
    for k, v in o.items():
        setattr(opt, k, v)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        return zb.ZbufferModelPts(opt).eval()


def pose_inputs():
    """Source poses the a15 fixture is generated for: the demo camera, a Matterport-style camera, a generic rigid pose."""
    rs = np.random.RandomState(77)
    A = rs.randn(3, 3)
    Q, _ = np.linalg.qr(A)
    G = np.eye(4, dtype=np.float32)
    G[:3, :3] = Q.astype(np.float32)
    G[:3, 3] = rs.randn(3).astype(np.float32)
    return [("demo", syn.demo_cameras(1)["P"]), ("mp3d", syn.mp3d_cameras(1)["P"]), ("rigid", G[None])]


POSE_CASES = (  # (model_setting, homography, rotation, direction, num, denom)
    [("gen_img", h, rot, d, None, None) for h in (False, True) for rot in (0.6, 0.25)
     for d in ("R", "L", "U", "D", "UR", "UL", "DR", "DL")]
    + [("gen_scene", h, 0.6, d, n, dn) for h in (False, True) for d in ("R", "L", "U", "D", "UL", "DR")
       for (n, dn) in ((0, 4), (1, 4), (3, 4), (4, 4), (2, 2))]
    + [("gen_scene", False, 0.6, d, n, dn) for d in ("C", "S") for (n, dn) in ((0, 64), (5, 64), (16, 64), (37, 64), (3, 8))]
    + [("gen_two_imgs", False, 0.6, d, n, 2) for d in ("R", "DL", "C", "S") for n in (0, 1, 2)])


def gen_poses(R):
    """a15: the reference's own get_rt_from_rot / eulerAnglesToRotationMatrix (models/z_buffermodel.py:186-242) for every
    branch -- gen_img directions at opt.rotation, num/denom sweeps, the 'C' rotation circle, the 'S' translation circle,
    homography -- plus get_combined (:703-708) and the (source pose, target pose, output key) schedule of forward_scene
    (:420-584) in both outpainting orders, recorded from the reference's own loop with the point-cloud renderer, the depth
    net and the decoder replaced by recorders (no_outpainting: the AR part is not what this fixture pins)."""
    zb = _import_reference_model()
    fx = {}
    m = _reference_model(zb)
    for pname, P in pose_inputs():
        fx[f"P_{pname}"] = P
    rows = []
    for ci, (setting, hom, rot, d, n, dn) in enumerate(POSE_CASES):
        m.opt.model_setting, m.opt.homography, m.opt.rotation = setting, hom, rot
        for pname, P in pose_inputs():
            with torch.no_grad():
                RTinv, RT = m.get_rt_from_rot(d, t(P), n, dn)
            fx[f"pose{ci}_{pname}_RT"] = RT.numpy()
            fx[f"pose{ci}_{pname}_RTinv"] = RTinv.numpy()
        rows.append(f"{setting}|{int(hom)}|{rot}|{d}|{'' if n is None else n}|{'' if dn is None else dn}")
    fx["pose_cases"] = np.array(rows)
    fx["euler_theta"] = np.array([[0.1, -0.7, 0.3], [0.0, 0.6, 0.0], [-1.2, 0.4, 2.0]])
    fx["euler_R"] = np.stack([m.eulerAnglesToRotationMatrix(th) for th in fx["euler_theta"]])
    # get_combined
    rs = np.random.RandomState(5)
    g, a = rs.randn(2, 3, 16, 16).astype(np.float32), rs.randn(2, 3, 16, 16).astype(np.float32)
    bgm = rs.rand(2, 16, 16) > 0.5
    fx.update(comb_gen=g, comb_ar=a, comb_bg=bgm, comb_out=m.get_combined(t(g), t(a), t(bgm)).numpy())
    # forward_scene schedule
    for tag, seq, dirs, split, setting in (("far_first", False, ["R", "L"], 2, "gen_scene"), ("sequential", True, ["R", "L"], 2, "gen_scene"),
                                           ("far_first_UC", False, ["U", "C", "DL"], 3, "gen_scene"), ("sequential_UC", True, ["U", "C", "DL"], 3, "gen_scene"),
                                           ("two_imgs", False, None, 4, "gen_two_imgs")):
        mm = _reference_model(zb, model_setting=setting, directions=dirs, num_split=split, sequential_outpainting=seq,
                              no_outpainting=True, num_samples=1)
        calls = []

        def fake_cumulative(fs, pts, K, K_inv, RT1, RT1inv, RT2, RT2inv, prior, fs_old, last_bg, RT3inv):
            k = len(calls)
            calls.append(dict(RT1=RT1.numpy().copy(), RT1inv=RT1inv.numpy().copy(), RT2=RT2.numpy().copy(),
                              RT2inv=RT2inv.numpy().copy(), RT3inv=None if RT3inv is None else RT3inv.numpy().copy(),
                              src_tag=float(fs.flatten()[0]), prior_tag=None if prior is None else float(prior.flatten()[0])))
            gen = torch.full((1, 3, 8, 8), float(k + 1))           # frame k is recognisable by its value
            return gen, torch.zeros(1, 8, 8, dtype=torch.bool), torch.full((1, 4, 5), float(k + 1)), torch.full((1, 3, 5), float(k + 1))
        mm.pts_transformer.forward_justpts_cumulative = fake_cumulative
        class _Fn(torch.nn.Module):
            def __init__(self, fn):
                super().__init__()
                self.fn = fn

            def forward(self, x):
                return self.fn(x)
        mm.pts_regressor = _Fn(lambda img: torch.zeros(1, 1, 8, 8))
        mm.projector = _Fn(lambda x: x)
        del mm._modules['loss_function']
        mm.__dict__['loss_function'] = lambda a, b: {}     # 'loss not used' (:591); its modules only exist on a GPU
        cam = {k: t(v) for k, v in syn.demo_cameras(1).items()}
        batch = {"images": [torch.zeros(1, 3, 8, 8)], "cameras": [cam]}
        if setting == "gen_two_imgs":
            batch["direction"] = torch.tensor(5)     # mapping[5] = 'UR'
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()), torch.no_grad():
            out = mm.forward_scene(batch)
        if isinstance(out, tuple):
            out = out[-1]
        fx[f"scene_{tag}_n"] = np.array(len(calls))
        for k, c in enumerate(calls):
            for key in ("RT1", "RT1inv", "RT2", "RT2inv"):
                fx[f"scene_{tag}_{k}_{key}"] = c[key]
            fx[f"scene_{tag}_{k}_RT3inv"] = np.zeros((0,)) if c["RT3inv"] is None else c["RT3inv"]
            fx[f"scene_{tag}_{k}_src"] = np.array(c["src_tag"])       # 0 = the input image, k = output of call k-1
        keys = sorted(k for k in out if k.startswith("PredImg_"))
        fx[f"scene_{tag}_pred_keys"] = np.array(keys)
        fx[f"scene_{tag}_pred_frame"] = np.array([float(out[k].flatten()[0]) for k in keys])   # which call produced the key
        fx[f"scene_{tag}_all_keys"] = np.array(sorted(out.keys()))
        fx[f"scene_{tag}_opts"] = np.array([str(seq), ",".join(dirs or ["UR(two)"]), str(split), setting])
    np.savez_compressed(os.path.join(HERE, "poses.npz"), **fx)
    print("poses.npz", len(fx))


from make_golden_inputs import zbuffer_cases  # noqa: E402  (shared with the tests)


def gen_zbuffer(R):
    """8f row 4: the reference's own DepthManipulator.project_zbuffer (models/projection/depth_manipulator.py:37-104; pure torch),
    run on ONE CPU thread so that its indexed assignment is sequential (last write per pixel stays).  Stored: the sampler on a
    strided grid plus checksums of the full tensors, and the projected depth on the strided grid."""
    import contextlib
    import io
    from models.projection.depth_manipulator import DepthManipulator
    fx = {}
    nt = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        seeds = {}
        # torch's sort is not stable, and among 65536 float z values a few are always equal: the order of such a pair changes
        # the result (the out-of-range flag goes by sorted POSITION).  The fixture defines it: ties in original point order.
        plain_sort = torch.Tensor.sort
        torch.Tensor.sort = lambda self, *a, **k: plain_sort(self, *a, **dict(k, stable=True))
        for name, W, d, cams, RT2 in zbuffer_cases(seeds):
            dm = DepthManipulator(W)
            with contextlib.redirect_stdout(io.StringIO()), torch.no_grad():
                smp, dep = dm.project_zbuffer(t(d), t(cams["K"]), t(cams["Kinv"]), t(cams["Pinv"]), t(RT2))
            smp, dep = smp.numpy(), dep.numpy()
            fx[f"{name}_sampler_sub"] = smp[:, :, ::3, ::3].copy()
            fx[f"{name}_sampler_rowsum"] = smp.astype(np.float64).sum(3)
            fx[f"{name}_filled"] = np.array([(smp[b, 0] != -2).sum() for b in range(smp.shape[0])])
            fx[f"{name}_depth_sub"] = dep[:, :, ::5, ::5].copy()
            fx[f"{name}_RT2"] = RT2
            fx[f"{name}_seed"] = np.array(0)
    finally:
        torch.set_num_threads(nt)
        torch.Tensor.sort = plain_sort
    fx["names"] = np.array([c[0] for c in zbuffer_cases()])
    np.savez_compressed(os.path.join(HERE, "zbuffer.npz"), **fx)
    print("zbuffer.npz", len(fx), {n: fx[f"{n}_filled"].tolist() for n in fx["names"]})


def scorer_opts():
    import argparse
    return argparse.Namespace(discriminator_losses="pix2pixHD", gan_mode="hinge", norm_D="spectralinstance", ndf=64, output_nc=3,
                              no_ganFeat_loss=False, isTrain=False, lambda_feat=10.0)


def gen_scorers(R):
    """8f row 3: the discriminator score of get_best_sample (models/z_buffermodel.py:254) from the reference's own
    DiscriminatorLoss / MultiscaleDiscriminator (models/losses/gan_loss.py:116-288, models/networks/discriminators.py:78-216) in
    eval mode on the CPU, weights from pixelsynth_amd.synthetic.fill_state_dict by key name (spectral-norm vectors included).
    Stored: D_Fake / D_real per candidate, the last feature map of both scales on a strided grid, the key / shape list.
    (The Places365 ResNet-18 is torchvision's -- absent here, no fixture.)"""
    import contextlib
    import io
    from models.losses.gan_loss import DiscriminatorLoss
    with contextlib.redirect_stdout(io.StringIO()):
        ref = DiscriminatorLoss(scorer_opts()).eval()
    shapes = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    ref.load_state_dict({k: t(v) for k, v in syn.fill_state_dict(shapes, 9).items()}, strict=True)
    cand, real = t(syn.image(51, 3, 3, 256)), t(np.repeat(syn.image(52, 1, 3, 256), 3, 0))
    with torch.no_grad():   # one candidate at a time, as get_best_sample scores them (:254)
        outs = [ref.run_discriminator_one_step(cand[i:i + 1], real[i:i + 1]) for i in range(3)]
        out = {k: torch.stack([o[k].reshape(()) for o in outs]) for k in outs[0]}
        feats = ref.netD.netD(torch.cat([cand, real], 0))
    np.savez_compressed(os.path.join(HERE, "scorers.npz"), D_Fake=out["D_Fake"].numpy(), D_real=out["D_real"].numpy(),
                        total=out["Total Loss"].numpy(), last0=feats[0][-1].numpy()[:, :, ::4, ::4], last1=feats[1][-1].numpy()[:, :, ::2, ::2],
                        keys=np.array([f"{k}:{','.join(map(str, v))}" for k, v in shapes.items()]), image_seeds=np.array([51, 52]),
                        weight_seed=np.array(9))
    print("scorers: D_Fake", out["D_Fake"].numpy(), "D_real", out["D_real"].numpy())


if __name__ == "__main__":
    torch.set_num_threads(8)
    R = _import_reference()
    which = sys.argv[1:] or ["projection", "orders", "layers", "blocks", "network", "ar", "vqvae", "networks", "poses", "zbuffer", "scorers"]
    if "projection" in which:
        gen_projection(R)
    if "orders" in which:
        gen_orders_masks(R)
    if "layers" in which:
        gen_lmconv_layers(R)
    if "blocks" in which:
        gen_blocks(R)
    if "network" in which:
        gen_network(R)
    if "ar" in which:
        gen_ar_trace(R)
    if "vqvae" in which:
        gen_vqvae(R)
    if "networks" in which:
        gen_networks(R)
    if "poses" in which:
        gen_poses(R)
    if "zbuffer" in which:
        gen_zbuffer(R)
    if "scorers" in which:
        gen_scorers(R)
