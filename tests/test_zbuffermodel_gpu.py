"""GPU tests of the orchestration mirror (pixelsynth_amd/z_buffermodel.py): poses, masks for batch in the
reference's layout, the batched view path against its pieces, and the reference-shaped forward_image."""
import os
import types

import numpy as np
import pytest
import torch

from oracle import c_oracle
from pixelsynth_amd import synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def tt(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def make_model(S=256, K=128, **kw):
    from pixelsynth_amd.z_buffermodel import ZbufferModelPts
    o = dict(W=S, use_rgb_features=True, splatter="xyblending", learn_default_feature=True, radius=4, pp_pixel=K, tau=1.0,
             rad_pow=2, accumulation="alphacomposite", background_smoothing_kernel_size=13, min_z=1.0, max_z=100.0,
             rotation=0.6, direction="R", temperature=0.7, model_setting="gen_img", seed=0, homography=False)
    o.update(kw)
    m = ZbufferModelPts(types.SimpleNamespace(**o)).eval()
    m.outpaint2.load_state_dict({k: torch.from_numpy(v) for k, v in syn.pixelcnn_state_dict(0).items()})
    return m.to(DEV)


def test_get_rt_from_rot_matches_restated_poses():
    m = make_model()
    P = tt(syn.demo_cameras(1)["P"])
    for direction, yaw in (("R", 0.6), ("L", -0.6)):
        m.opt.direction = direction
        RTinv, RT = m.get_rt_from_rot(direction, P)
        inv_ref, rt_ref = syn.yaw_pose(syn.demo_cameras(1)["P"], yaw)
        np.testing.assert_allclose(RT.cpu().numpy(), rt_ref, rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(RTinv.cpu().numpy(), inv_ref, rtol=1e-5, atol=1e-5)
    m.opt.model_setting = "gen_scene"
    RTinv, RT = m.get_rt_from_rot("C", P, 5, 64)      # circle trajectory, z_buffermodel.py:217-225
    inv_ref, rt_ref = syn.circle_pose(syn.demo_cameras(1)["P"], 5, 64)
    np.testing.assert_allclose(RT.cpu().numpy(), rt_ref, rtol=1e-6, atol=1e-6)
    RTinv, RT = m.get_rt_from_rot("U", P, 3, 8)       # rotvec * num / denom
    rt_ref = syn.yaw_pose(syn.demo_cameras(1)["P"], 0.0, pitch=-0.3 * 3 / 8)[1]
    np.testing.assert_allclose(RT.cpu().numpy(), rt_ref, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose((RT @ RTinv).cpu().numpy()[0], np.eye(4), atol=1e-5)


def test_get_masks_for_batch_reference_layout_and_compact():
    m = make_model()
    bgs = syn.background_masks(256)
    names = ["right_half", "half_plus_island", "none", "all"]
    bg = tt(np.stack([bgs[n] for n in names]))
    mi, mu, md, gen_order = m.get_masks_for_batch(None, None, bg)
    assert mi.shape == (4 * 513, 9, 1024) and mu.shape == (4 * 160, 9, 1024) and md.shape == (4 * 80, 9, 1024)
    plan = m.get_masks_for_batch(None, None, bg, compact=True)
    for b, n in enumerate(names):
        ref = c_oracle.masks_for_background(bgs[n], 32)
        assert np.array_equal(np.asarray(gen_order[b]), ref["order"]), n
        assert np.array_equal(mi.view(4, 513, 9, 1024)[b, 7].cpu().numpy(), ref["mask_init"][0])
        assert np.array_equal(mu.view(4, 160, 9, 1024)[b, 159].cpu().numpy(), ref["mask_undilated"][0])
        assert np.array_equal(md.view(4, 80, 9, 1024)[b, 0].cpu().numpy(), ref["mask_dilated"][0])
        assert np.array_equal(plan.region[b].cpu().numpy(), ref["bg32"].reshape(-1))
        assert np.array_equal(plan.order_loc[b].cpu().numpy(), ref["order"][:, 0] * 32 + ref["order"][:, 1])
    # first_step = earliest sampled order position over the batch ("all" background starts at 0)
    assert plan.first_step == 0
    plan2 = m.get_masks_for_batch(None, None, bg[:1], compact=True)
    reg = plan2.region[0].cpu().numpy()[plan2.order_loc[0].cpu().numpy()]
    assert plan2.first_step == int(np.nonzero(reg)[0][0]) > 0


def test_outpaint_views_equals_its_pieces():
    """The batched path (fused project+splat -> plan -> fused AR) against the piecewise mirrors."""
    V = 3
    m = make_model()
    cam = syn.demo_cameras(V)
    img, depth = syn.image(1, V, 3, 256), syn.depth_smooth(2, V, 256, 1.0, 100.0)
    RT2 = np.empty((V, 4, 4), np.float32)
    RT2inv = np.empty((V, 4, 4), np.float32)
    for v, yaw in enumerate((-0.6, 0.2, 0.6)):
        inv, rt = syn.yaw_pose(cam["P"][v:v + 1], yaw)
        RT2[v], RT2inv[v] = rt[0], inv[0]
    codes = syn.codes(3, V)
    u = np.random.RandomState(5).rand(V, 1024).astype(np.float32)
    args = [tt(a) for a in (img, depth, cam["K"], cam["Kinv"], cam["P"], cam["Pinv"], RT2, RT2inv)]
    out = m.outpaint_views(*args, tt(codes), temperature=0.7, uniforms=tt(u))
    torch.cuda.synchronize()
    # splat piece vs the oracle
    sampler = c_oracle.project_pts(depth, cam["K"], cam["Kinv"], cam["Pinv"], RT2, 256)
    ref = c_oracle.splat_forward(np.ascontiguousarray(sampler.transpose(0, 2, 1)), img.reshape(V, 3, -1), 256)
    assert np.array_equal(out["background_mask"].cpu().numpy(), ref["bg"])
    np.testing.assert_allclose(out["gen_fs"].cpu().numpy(), ref["feat"], rtol=0, atol=1e-6)
    # AR piece: per view alone, same uniforms -> same codes (batch independence through the whole path)
    got = out["codes"].cpu().numpy()
    for v in range(V):
        single = m.outpaint_views(*[a[v:v + 1] for a in args], tt(codes[v:v + 1]), temperature=0.7, uniforms=tt(u[v:v + 1]))
        assert np.array_equal(single["codes"].cpu().numpy()[0], got[v])
        keep = ~c_oracle.masks_for_background(ref["bg"][v], 32)["bg32"].astype(bool)
        assert np.array_equal(got[v][keep], codes[v][keep])
    comb = m.get_combined(out["gen_fs"], torch.ones_like(out["gen_fs"]) * 7, out["background_mask"])
    bgm = out["background_mask"][:, None].expand_as(comb)
    assert torch.all(comb[bgm] == 7) and torch.equal(comb[~bgm], out["gen_fs"][~bgm])


def test_forward_image_reference_shaped_outputs():
    m = make_model()
    cam = {k: torch.from_numpy(v) for k, v in syn.demo_cameras(1).items()}
    batch = {"images": [torch.from_numpy(syn.image(4, 1, 3, 256))], "cameras": [cam],
             "depths": [torch.from_numpy(syn.depth_smooth(5, 1, 256, 1.0, 100.0))],
             "codes": torch.from_numpy(syn.codes(6, 1))}
    _, outputs = m.forward_image(batch)
    for k in ("InputImg", "PredDepthImg", "ForegroundImg", "FeaturesImg", "PredCodes"):
        assert k in outputs
    assert outputs["FeaturesImg"].shape == (1, 3, 256, 256) and outputs["ForegroundImg"].shape == (1, 1, 256, 256)
    fg32 = torch.nn.functional.avg_pool2d(outputs["ForegroundImg"], 8)[0, 0] > 0     # blocks that are not all background
    keep = fg32.cpu().numpy()
    assert np.array_equal(outputs["PredCodes"][0].cpu().numpy()[keep], syn.codes(6, 1)[0][keep])
    assert 0.2 < 1 - keep.mean() < 0.9   # a 0.6 rad yaw leaves a large region to outpaint


def test_views_with_the_vqvae_in_the_loop():
    """reprojected view -> VQ-VAE top codes (device int32) -> AR outpainting -> decode_code.
    The observed part of the code grid must come through untouched, the sampled part must be valid codes, and the
    decoded image must equal the oracle's decode of the same codes."""
    from oracle import vqvae_oracle as vo
    m = make_model(vqvae=True)
    sd = {k: torch.from_numpy(v) for k, v in syn.vqvae_state_dict(0).items()}
    m.vqvae.load_state_dict(sd, strict=True)
    m = m.to(DEV).eval()
    V = 2
    cam = syn.demo_cameras(V)
    img = tt(syn.image(21, V, 3, 256))
    depth = tt(syn.depth_smooth(22, V, 256, 1.0, 100.0))
    rts = [syn.yaw_pose(cam["P"][v:v + 1], y) for v, y in enumerate((0.3, -0.45))]
    RT2 = tt(np.concatenate([r[1] for r in rts]))
    RT2inv = tt(np.concatenate([r[0] for r in rts]))
    uni = tt(np.random.RandomState(5).rand(V, 1024).astype(np.float32))
    out = m.outpaint_views(img, depth, tt(cam["K"]), tt(cam["Kinv"]), tt(cam["P"]), tt(cam["Pinv"]), RT2, RT2inv, None,
                           temperature=0.7, uniforms=uni)
    m.outpaint2.engine(32, 32, V).check()
    codes = out["codes"].cpu().numpy()
    enc = m.vqvae.encode_codes(out["gen_fs"]).cpu().numpy()
    region = out["plan"].region.cpu().numpy().astype(bool).reshape(V, 32, 32)
    assert region.any() and (~region).any()
    assert np.array_equal(codes[~region], enc[~region])
    assert codes.min() >= 0 and codes.max() < 512
    dec = m.vqvae.decode_code(out["codes"])
    assert tuple(dec.shape) == (V, 3, 256, 256)
    with torch.no_grad():
        want = vo.decode_code(sd, torch.from_numpy(codes).long())
    np.testing.assert_allclose(dec.cpu().numpy(), want.numpy(), rtol=1e-4, atol=1e-5)


def test_driver_renders_a_trajectory_and_writes_the_video_layout(tmp_path):
    """the driver (counterpart of demo.py / create_vid.py): poses of the 'C' circle and of a direction sweep,
    views rendered from the source, PNGs under <out>/video/%d.png starting with the source frame."""
    from PIL import Image
    from pixelsynth_amd import driver
    driver.main(["--trajectory", "circle", "--frames", "5", "--batch", "3", "--out", str(tmp_path)])
    names = sorted(os.listdir(tmp_path / "video"), key=lambda n: int(n.split(".")[0]))
    assert names == [f"{i}.png" for i in range(6)]
    frames = [np.asarray(Image.open(tmp_path / "video" / n)) for n in names]
    assert all(f.shape == (256, 256, 3) and f.dtype == np.uint8 for f in frames)
    src = ((np.clip(syn.image(1000, 1, 3, 256)[0], -1, 1) * 0.5 + 0.5) * 255.0 + 0.5).astype(np.uint8).transpose(1, 2, 0)
    assert np.array_equal(frames[0], src)
    assert not np.array_equal(frames[1], frames[3])                       # different poses give different views
    m = driver.build_model(torch.device(DEV))
    P = tt(syn.demo_cameras(1)["P"])
    poses = driver.trajectory(m, P, "circle", 8)
    assert [p[0] for p in poses] == [f"C_{i}" for i in range(8)]
    inv_ref, rt_ref = syn.circle_pose(syn.demo_cameras(1)["P"], 3, 8)
    np.testing.assert_allclose(poses[3][2].cpu().numpy(), rt_ref, rtol=1e-6, atol=1e-6)
    sweep = driver.trajectory(m, P, "R", 4)
    np.testing.assert_allclose(sweep[-1][2].cpu().numpy(), syn.yaw_pose(syn.demo_cameras(1)["P"], 0.6)[1], rtol=1e-6, atol=1e-6)
