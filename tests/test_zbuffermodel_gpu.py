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


def test_get_rt_from_rot_on_the_device_matches_the_reference_fixture():
    """a15 on device tensors against poses.npz, recorded from the reference's own get_rt_from_rot
    (tests/golden/make_golden.py:gen_poses); every branch: directions at opt.rotation, num/denom sweeps, the 'C' and 'S'
    circles, homography.  (tests/test_poses_cpu.py runs the same fixture on the host.)  Tolerance: torch.inverse on the
    GPU is another LU than on the CPU, 1e-5."""
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "poses.npz"))
    m = make_model()
    for ci, row in enumerate(fx["pose_cases"]):
        setting, hom, rot, d, n, dn = str(row).split("|")
        m.opt.model_setting, m.opt.homography, m.opt.rotation = setting, bool(int(hom)), float(rot)
        for pname in ("demo", "mp3d", "rigid"):
            RTinv, RT = m.get_rt_from_rot(d, tt(fx[f"P_{pname}"]), int(n) if n else None, int(dn) if dn else None)
            assert RT.is_cuda and RTinv.is_cuda
            np.testing.assert_allclose(RT.cpu().numpy(), fx[f"pose{ci}_{pname}_RT"], rtol=0, atol=1e-6)
            np.testing.assert_allclose(RTinv.cpu().numpy(), fx[f"pose{ci}_{pname}_RTinv"], rtol=1e-5, atol=1e-5)


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


def _scene_model(**kw):
    o = dict(vqvae=True, model_setting="gen_scene", num_split=2, directions=["R", "L"], num_samples=1,
             sequential_outpainting=False)
    o.update(kw)
    m = make_model(**o)
    m.vqvae.load_state_dict({k: torch.from_numpy(v) for k, v in syn.vqvae_state_dict(0).items()}, strict=True)
    return m.to(DEV).eval()


def _scene_batch():
    cam = {k: torch.from_numpy(v) for k, v in syn.demo_cameras(1).items()}
    return {"images": [torch.from_numpy(syn.image(31, 1, 3, 256))], "cameras": [cam], "depth_fn": syn.depth_from_image}


def _record_scene(m, batch):
    """Run forward_scene with every cumulative reprojection call recorded (arguments and results)."""
    calls, inner = [], m.pts_transformer.forward_justpts_cumulative

    def spy(*a):
        r = inner(*a)
        calls.append((a, r))
        return r
    m.pts_transformer.forward_justpts_cumulative = spy
    try:
        _, out = m(batch)
    finally:
        m.pts_transformer.forward_justpts_cumulative = inner
    return calls, out


@pytest.mark.parametrize("sequential", [False, True])
def test_forward_scene_pose_schedule_and_state_chain(sequential):
    """SURVEY 8f row 4, z_buffermodel.py:420-584: the chained trajectory.  The schedule of (source pose, target pose)
    pairs is restated here from the reference's loop and compared with what the model hands to the cumulative
    reprojection; every frame must be rendered from the previous generated frame, on top of the previous frame's
    cloud, features and background mask; the output keys are the reference's."""
    m = _scene_model(sequential_outpainting=sequential)
    batch = _scene_batch()
    calls, out = _record_scene(m, batch)
    P = tt(syn.demo_cameras(1)["P"])
    pose = lambda d, n: m.get_rt_from_rot(d, P, n, 2)[1].cpu().numpy()
    src = P.cpu().numpy()
    if not sequential:   # far end first, then back towards the source view; the next direction starts from R_0
        want = [(src, pose("R", 2), "R_2"), (pose("R", 2), pose("R", 1), "R_1"), (pose("R", 1), pose("R", 0), "R_0"),
                (pose("R", 0), pose("L", 2), "L_2"), (pose("L", 2), pose("L", 1), "L_1"), (pose("L", 1), pose("L", 0), "L_0")]
    else:                # 0, 1, 2 in order; the next direction starts from the last rendered view R_2
        want = [(src, pose("R", 0), "R_0"), (pose("R", 0), pose("R", 1), "R_1"), (pose("R", 1), pose("R", 2), "R_2"),
                (pose("R", 2), pose("L", 0), "L_0"), (pose("L", 0), pose("L", 1), "L_1"), (pose("L", 1), pose("L", 2), "L_2")]
    assert len(calls) == len(want)
    prev_img = batch["images"][0].to(DEV)
    prev = None
    for (a, r), (rt_in, rt_out, tag) in zip(calls, want):
        src1, depth, K, Kinv, RT1, RT1inv, RT2, RT2inv, prior, src2, last_bg, RT3inv = a
        np.testing.assert_allclose(RT1.cpu().numpy(), rt_in, rtol=0, atol=1e-6)
        np.testing.assert_allclose(RT2.cpu().numpy(), rt_out, rtol=0, atol=1e-6)
        np.testing.assert_allclose(RT1inv.cpu().numpy(), np.linalg.inv(rt_in), rtol=0, atol=1e-5)
        assert torch.equal(src1, prev_img)                                   # rendered from the previous generated frame
        assert torch.equal(depth, syn.depth_from_image(prev_img))
        if prev is None:
            assert prior is None and src2 is None and last_bg is None and RT3inv is None
        else:
            (pa, pr) = prev
            assert prior is pr[2] and src2 is pr[3] and last_bg is pr[1] and torch.equal(RT3inv, pa[7])
            n_new = int(pr[1].sum())
            assert r[2].shape[2] == n_new + pr[2].shape[2]                     # only last frame's background is new
        assert torch.equal(out[f"FeaturesImg_{tag}"], r[0])
        prev_img, prev = out[f"PredImg_{tag}"], (a, r)
        assert tuple(prev_img.shape) == (1, 3, 256, 256) and torch.isfinite(prev_img).all()
    for d in ("R", "L"):
        assert tuple(out[f"PredDepthImg_{d}_2"].shape) == (1, 1, 256, 256)
        assert tuple(out[f"ForegroundImg_{d}_2"].shape) == (1, 1, 256, 256)
    m.outpaint2.engine(32, 32, 1).check()
    # an outpainted frame = reprojected features where visible, decoded sample elsewhere (no refinement net here)
    a, r = calls[0]
    fg = ~r[1][:, None].expand(1, 3, 256, 256)
    assert torch.equal(out["PredImg_" + want[0][2]][fg], r[0][fg])


def test_forward_scene_first_frame_equals_the_single_view_path():
    """No prior cloud yet: the first chained frame is forward_justpts of the source, and its outpainting is what
    outpaint_views produces for the same draws."""
    m = _scene_model()
    batch = _scene_batch()
    calls, out = _record_scene(m, batch)
    a, r = calls[0]
    img, cam = batch["images"][0].to(DEV), {k: v.to(DEV) for k, v in batch["cameras"][0].items()}
    RTinv, RT = m.get_rt_from_rot("R", cam["P"], 2, 2)
    u = torch.rand(1, 1024, generator=torch.Generator(device="cpu").manual_seed(0)).to(DEV)
    ref = m.outpaint_views(img, syn.depth_from_image(img), cam["K"], cam["Kinv"], cam["P"], cam["Pinv"], RT, RTinv, None,
                           temperature=0.7, uniforms=u)
    assert torch.equal(r[0], ref["gen_fs"]) and torch.equal(r[1], ref["background_mask"])
    want = m.get_combined(ref["gen_fs"], m.vqvae.decode_code(ref["codes"]), ref["background_mask"])
    assert torch.equal(out["PredImg_R_2"], want)


def test_get_best_sample_ranks_candidates():
    """num_samples > 1: every candidate is scored by the (injected) discriminator and scene classifier and the
    reference's rank rule picks one (z_buffermodel.py:244-276); without scorers the call refuses."""
    from pixelsynth_amd.z_buffermodel import build_ar_plan, rank_samples
    m = _scene_model(num_samples=3)
    img = tt(syn.image(31, 1, 3, 256))
    cam = {k: tt(v) for k, v in syn.demo_cameras(1).items()}
    RTinv, RT = m.get_rt_from_rot("R", cam["P"], 2, 2)
    gen_fs, bg = m.pts_transformer.forward_justpts(img, syn.depth_from_image(img), cam["K"], cam["Kinv"], cam["P"],
                                                   cam["Pinv"], RT, RTinv)
    plan = build_ar_plan(bg, 32)
    codes = m.vqvae.encode_codes(gen_fs)
    with pytest.raises(RuntimeError, match="num_samples"):
        m.get_best_sample(plan, codes, bg, gen_fs, None, img)
    seen = []

    class D:   # scores a candidate by its mean (any deterministic function will do)
        def run_discriminator_one_step(self, fake, real):
            seen.append(fake)
            return {"D_Fake": fake.mean().reshape(1)}
    class C(torch.nn.Module):   # a stand-in classifier: ten "logits" from the image mean
        def forward(self, x):
            return torch.cat([x.mean().reshape(1, 1) * k for k in range(1, 11)], 1)
    m.classifier = C()
    best = m.get_best_sample(plan, codes, bg, gen_fs, D(), img)
    assert len(seen) == 3 and not torch.equal(seen[0], seen[1])
    disc = [float(s.mean()) for s in seen]
    entr = [m._entropy_score(s) for s in seen]
    assert torch.equal(best, seen[rank_samples(disc, entr)])


def test_get_best_sample_runs_end_to_end_with_the_real_scorers():
    """SURVEY 8f row 3, the quality mode (num_samples > 1) with the scorers of this repository: the multiscale discriminator
    mirror (weights as in tests/golden/scorers.npz, scores pinned there against the reference's own class) and the ResNet-18
    the model builds itself, as the reference does (z_buffermodel.py:88).  The kept candidate is the one the reference's rank
    rule picks from the scores of all candidates; forward_image routes through it."""
    import argparse
    from pixelsynth_amd.losses import DiscriminatorLoss
    from pixelsynth_amd.networks import ResNet18
    from pixelsynth_amd.z_buffermodel import build_ar_plan, rank_samples
    torch.manual_seed(0)
    m = _scene_model(num_samples=4, model_setting="gen_img")
    assert isinstance(m.classifier, ResNet18)
    opt = argparse.Namespace(discriminator_losses="pix2pixHD", gan_mode="hinge", norm_D="spectralinstance", ndf=64, output_nc=3,
                             no_ganFeat_loss=False, isTrain=False, lambda_feat=10.0)
    netD = DiscriminatorLoss(opt).eval()
    shapes = {k: tuple(v.shape) for k, v in netD.state_dict().items()}
    netD.load_state_dict({k: torch.from_numpy(v) for k, v in syn.fill_state_dict(shapes, 9).items()}, strict=True)
    netD = netD.to(DEV)
    img = tt(syn.image(31, 1, 3, 256))
    cam = {k: tt(v) for k, v in syn.demo_cameras(1).items()}
    RTinv, RT = m.get_rt_from_rot("R", cam["P"])
    gen_fs, bg = m.pts_transformer.forward_justpts(img, syn.depth_from_image(img), cam["K"], cam["Kinv"], cam["P"],
                                                   cam["Pinv"], RT, RTinv)
    plan = build_ar_plan(bg, 32)
    codes = m.vqvae.encode_codes(gen_fs)
    seen, disc, entr = [], [], []
    inner, inner_e = netD.run_discriminator_one_step, m._entropy_score

    def spy(fake, real):            # the scores the model ranks with, as it computed them
        seen.append(fake.clone())
        out = inner(fake, real)
        disc.append(float(out["D_Fake"].mean().cpu()))
        return out

    def spy_e(g):
        entr.append(inner_e(g))
        return entr[-1]
    netD.run_discriminator_one_step, m._entropy_score = spy, spy_e
    best = m.get_best_sample(plan, codes, bg, gen_fs, netD, img)
    assert len(seen) == 4 and all(tuple(s.shape) == (1, 3, 256, 256) for s in seen) and len(entr) == 4
    assert len(set(np.round(disc, 7))) > 1 and all(np.isfinite(entr))
    assert torch.equal(best, seen[rank_samples(disc, entr)])
    for s_, d_ in zip(seen, disc):  # (the convolution library may pick another algorithm on a second call: 1e-4)
        assert abs(float(inner(s_, img)["D_Fake"].mean()) - d_) < 1e-4
    seen.clear(); disc.clear(); entr.clear()
    # the reference-shaped entry point takes the same road (z_buffermodel.py:349)
    batch = {"images": [img.cpu()], "cameras": [{k: v.cpu() for k, v in cam.items()}], "depths": [syn.depth_from_image(img).cpu()]}
    _, out = m.forward_image(batch, netD=netD)
    assert len(seen) == 4 and torch.equal(out["PredImg"], seen[rank_samples(disc, entr)])


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
    # equal chunks: their AR runs overlap (outpaint_pipelined) -- the pictures are those of one batch of all six views
    for batch, sub in ((3, "a"), (6, "b")):
        driver.main(["--trajectory", "circle", "--frames", "6", "--batch", str(batch), "--out", str(tmp_path / sub)])
    pics = [[np.asarray(Image.open(tmp_path / sub / "video" / f"{i}.png")).astype(np.int32) for i in range(7)] for sub in ("a", "b")]
    assert all(np.abs(x - y).max() <= 2 for x, y in zip(*pics)) and not np.array_equal(pics[0][2], pics[0][5])
    m = driver.build_model(torch.device(DEV))
    P = tt(syn.demo_cameras(1)["P"])
    poses = driver.trajectory(m, P, "circle", 8)
    assert [p[0] for p in poses] == [f"C_{i}" for i in range(8)]
    inv_ref, rt_ref = syn.circle_pose(syn.demo_cameras(1)["P"], 3, 8)
    np.testing.assert_allclose(poses[3][2].cpu().numpy(), rt_ref, rtol=1e-6, atol=1e-6)
    sweep = driver.trajectory(m, P, "R", 4)
    np.testing.assert_allclose(sweep[-1][2].cpu().numpy(), syn.yaw_pose(syn.demo_cameras(1)["P"], 0.6)[1], rtol=1e-6, atol=1e-6)


def test_driver_chained_scene_writes_scene_and_video_layout(tmp_path):
    """--scene: forward_scene on one GPU, files in the reference's save_scene / save_video layout (demo.py:100-164)."""
    from pixelsynth_amd import driver
    driver.main(["--scene", "R", "U", "--num-split", "2", "--out", str(tmp_path)])
    assert sorted(os.listdir(tmp_path / "scene")) == ["output_image_R_0001.png", "output_image_R_0002.png", "output_image_U_0001.png"]
    # R: 1, then back 1, 0; U (num_split 1): back 0  -> 1 + 3 + 1 frames
    assert sorted(os.listdir(tmp_path / "video"), key=lambda n: int(n.split(".")[0])) == [f"{i}.png" for i in range(5)]


def test_forward_image_with_every_network_in_the_loop():
    """SURVEY 8f rows 1-2 around the path: depth Unet -> reproject/splat -> VQ-VAE codes -> AR -> decode -> blend ->
    refinement decoder, all on the device; the output dict is the reference's (z_buffermodel.py:385-417)."""
    o = vars(syn.network_opts())
    m = make_model(vqvae=True, **o)
    for mod, seed in ((m.pts_regressor, 5), (m.projector, 5)):
        shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
        mod.load_state_dict({k: torch.from_numpy(v) for k, v in syn.fill_state_dict(shapes, seed).items()}, strict=True)
    m.vqvae.load_state_dict({k: torch.from_numpy(v) for k, v in syn.vqvae_state_dict(0).items()}, strict=True)
    m = m.to(DEV).eval()
    cam = {k: torch.from_numpy(v) for k, v in syn.demo_cameras(1).items()}
    img = torch.from_numpy(syn.image(4, 1, 3, 256))
    torch.manual_seed(3)
    _, out = m.forward_image({"images": [img], "cameras": [cam]})
    m.outpaint2.engine(32, 32, 1).check()
    depth = torch.sigmoid(m.pts_regressor(img.to(DEV))) * 99.0 + 1.0
    assert torch.allclose(out["PredDepthImg"], depth / 5 - 1, atol=1e-5)
    assert tuple(out["PredImg"].shape) == (1, 3, 256, 256) and torch.isfinite(out["PredImg"]).all()
    assert float(out["PredImg"].abs().max()) <= 1.0          # tanh
    bg = out["ForegroundImg"][0, 0] == 0
    assert 0.2 < float(bg.float().mean()) < 0.9


def test_a_decoder_pass_that_overflowed_fp16_is_run_again_in_fp32():
    """_decode_checked: when a split-fp16 convolution of the refinement decoder raised its overflow flag (an activation beyond
    65000: csrc/conv_f16x3.hip), the model warns and runs the pass again with every convolution through torch; without the flag the
    split-fp16 kernels are what ran."""
    from pixelsynth_amd.networks import architectures as A
    o = vars(syn.network_opts())
    m = make_model(vqvae=True, **o)
    for mod, seed in ((m.pts_regressor, 5), (m.projector, 5)):
        shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
        mod.load_state_dict({k: torch.from_numpy(v) for k, v in syn.fill_state_dict(shapes, seed).items()}, strict=True)
    m.vqvae.load_state_dict({k: torch.from_numpy(v) for k, v in syn.vqvae_state_dict(0).items()}, strict=True)
    m = m.to(DEV).eval()
    gen_fs = tt(syn.image(8, 2, 3, 256))
    bgm = tt(syn.background_masks(256)["ragged"])[None].expand(2, -1, -1).contiguous()
    codes = tt(syn.codes(9, 2)).to(torch.int64)
    calls, raise_at = [], [None]
    real = A._f16x3_conv

    def spy(*a, **k):
        calls.append(A._conv_mode(None))
        y = real(*a, **k)
        if raise_at[0] is not None and len(calls) == raise_at[0]:
            A._overflow_flag(gen_fs.device).fill_(1)          # as the kernel would, in the middle of the pass
        return y
    A._f16x3_conv = spy
    try:
        with torch.no_grad():
            img = m._decode_checked(gen_fs, bgm, codes)
            assert len(calls) == 13 and torch.isfinite(img).all()
            del calls[:]
            raise_at[0] = 5
            with pytest.warns(UserWarning, match="run again in fp32"):
                img2 = m._decode_checked(gen_fs, bgm, codes)
            assert len(calls) == 13                           # the first attempt only: the rerun went through torch
            assert torch.isfinite(img2).all() and img2.shape == img.shape
            A.check_f16x3_overflow(gen_fs.device)              # (cleared by the report)
            # a flag an earlier, UNCHECKED pass left behind is not this pass's: cleared before the pass, no rerun, no warning
            del calls[:]
            raise_at[0] = None
            A._overflow_flag(gen_fs.device).fill_(1)
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("error")
                img3 = m._decode_checked(gen_fs, bgm, codes)
            assert len(calls) == 13 and torch.isfinite(img3).all()
            # get_best_sample's candidates (forward_image with num_samples, every frame of forward_scene) and the no_outpainting
            # projector call go through the same check: an overflow in the middle of a candidate's pass reruns THAT candidate
            plan = m.get_masks_for_batch(None, None, bgm[:1], compact=True)
            del calls[:]
            raise_at[0] = 7
            with pytest.warns(UserWarning, match="run again in fp32"):
                best = m.get_best_sample(plan, codes[:1], bgm[:1], gen_fs[:1], None, gen_fs[:1])
            assert len(calls) == 13 and tuple(best.shape) == (1, 3, 256, 256) and torch.isfinite(best).all()
            A.check_f16x3_overflow(gen_fs.device)
    finally:
        A._f16x3_conv = real


@pytest.mark.parametrize("depth", [4, 3])
def test_small_batches_pipelined_four_and_three_deep(monkeypatch, depth):
    """outpaint_pipelined keeps up to pipe_depth = 4 batches in flight (3 with PS_PIPE_DEPTH), here of 8 views: every launch takes what is
    left of each batch's current wavefront, oldest first (lmconv.model.pack_launches), and a batch comes back at most depth - 1 calls
    late.  Six different batches: every batch's codes are outpaint_planned's bit for bit, in order, and the shared launches are fewer
    than the batches' own."""
    if depth != 4:
        monkeypatch.setenv("PS_PIPE_DEPTH", str(depth))
    m = make_model()
    V = 8
    assert m.pipe_depth(V) == depth and m.pipe_frames(V) == depth * V and m.pipe_depth(128) == depth
    cam = syn.demo_cameras(V)
    K, Kinv, P, Pinv = (tt(cam[k]) for k in ("K", "Kinv", "P", "Pinv"))
    batches = []
    for b in range(6):
        img, depth_ = tt(syn.image(281 + b, V, 3, 256)), tt(syn.depth_smooth(291 + b, V, 256, 1.0, 100.0))
        yaws = np.linspace(-0.7 + 0.05 * b, 0.5 + 0.05 * b, V)
        rts = [syn.yaw_pose(cam["P"][v:v + 1], float(y)) for v, y in enumerate(yaws)]
        RT2, RT2inv = tt(np.concatenate([r[1] for r in rts])), tt(np.concatenate([r[0] for r in rts]))
        batches.append(((img, depth_, K, Kinv, P, Pinv, RT2, RT2inv), tt(syn.codes(301 + b, V)), tt(np.random.RandomState(311 + b).rand(V, 1024).astype(np.float32))))
    eng1 = m.outpaint2.engine(32, 32, V)
    n0 = sum(eng1.launch_counts().values())
    ref = [m.outpaint_planned(m.plan_views(*a), c, temperature=0.7, uniforms=u)["codes"].clone() for a, c, u in batches]
    own = sum(eng1.launch_counts().values()) - n0
    eng = m.outpaint2.engine(32, 32, depth * V)
    n0 = sum(eng.launch_counts().values())
    got, late = [], []
    for k, (a, c, u) in enumerate(batches):
        done = m.outpaint_pipelined(m.plan_views(*a), c, temperature=0.7, uniforms=u)
        late.append(done is None)
        if done is not None:
            got.append(done["codes"].clone())
    got += [o["codes"].clone() for o in m.outpaint_flush()]
    torch.cuda.synchronize()
    eng.check()
    assert late[0] and not late[-1] and sum(late) <= depth - 1 and len(got) == 6 and m.outpaint_flush() == []     # (at most depth - 1 calls late, in order)
    for b in range(6):
        assert torch.equal(got[b], ref[b]), (b, int((got[b] != ref[b]).sum()))
    assert sum(eng.launch_counts().values()) - n0 < 0.8 * own


def test_middle_sized_batches_keep_more_batches_in_flight():
    """pipe_depth(V): as many batches as it takes to have ~384 frames in the handle, between four and eight -- the throughput form's launches
    take 1 024 columns, which batches of 24 to 96 views only fill with more of them in flight (lmconv.model.TP_MIN_FRAMES = 24 since the
    launches are packed).  Ten different batches of 32 views, eight in flight, throughput-form launches: every batch's codes are
    outpaint_planned's bit for bit, in order, at most seven calls late."""
    from pixelsynth_amd.lmconv import model as lm
    m = make_model()
    assert lm.TP_MIN_FRAMES == 24
    assert [m.pipe_depth(v) for v in (8, 16, 24, 32, 48, 64, 96, 128, 256)] == [4, 4, 8, 8, 8, 6, 4, 4, 4]
    V, depth, nb = 32, 8, 10
    cam = syn.demo_cameras(V)
    K, Kinv, P, Pinv = (tt(cam[k]) for k in ("K", "Kinv", "P", "Pinv"))
    batches = []
    for b in range(nb):
        img, depth_ = tt(syn.image(481 + b, V, 3, 256)), tt(syn.depth_smooth(491 + b, V, 256, 1.0, 100.0))
        yaws = np.linspace(-0.6 + 0.03 * b, 0.6 - 0.03 * b, V)
        rts = [syn.yaw_pose(cam["P"][v:v + 1], float(y)) for v, y in enumerate(yaws)]
        RT2, RT2inv = tt(np.concatenate([r[1] for r in rts])), tt(np.concatenate([r[0] for r in rts]))
        batches.append(((img, depth_, K, Kinv, P, Pinv, RT2, RT2inv), tt(syn.codes(501 + b, V)), tt(np.random.RandomState(511 + b).rand(V, 1024).astype(np.float32))))
    ref = [m.outpaint_planned(m.plan_views(*a), c, temperature=0.7, uniforms=u)["codes"].clone() for a, c, u in batches]
    eng = m.outpaint2.engine(32, 32, m.pipe_frames(V))
    assert m.pipe_frames(V) == depth * V
    n0 = dict(eng.launch_counts())
    got, late = [], []
    for a, c, u in batches:
        done = m.outpaint_pipelined(m.plan_views(*a), c, temperature=0.7, uniforms=u)
        late.append(done is None)
        if done is not None:
            got.append(done["codes"].clone())
    got += [o["codes"].clone() for o in m.outpaint_flush()]
    torch.cuda.synchronize()
    eng.check()
    assert late[0] and sum(late) <= depth - 1 and len(got) == nb
    for b in range(nb):
        assert torch.equal(got[b], ref[b]), (b, int((got[b] != ref[b]).sum()))
    n1 = eng.launch_counts()
    assert sum(n1.get(k, 0) - n0.get(k, 0) for k in ("k_column_tp", "k_column_tp8")) > 0      # (the throughput form did run)


def test_pipelined_batches_with_different_temperatures_and_a_reset():
    """outpaint_pipelined when the temperature changes from one batch to the next: the tail wavefronts of the batch in flight run as
    launches of their own with THEIR temperature (a merged launch has one), so every batch's codes are still outpaint_planned's at its
    own temperature; outpaint_reset drops a batch in flight (a sequence cut short must not leak into the next one)."""
    m = make_model()
    V = 16
    cam = syn.demo_cameras(V)
    K, Kinv, P, Pinv = (tt(cam[k]) for k in ("K", "Kinv", "P", "Pinv"))
    batches = []
    for b, temp in enumerate((0.7, 1.0, 1.0)):
        img, depth = tt(syn.image(181 + b, V, 3, 256)), tt(syn.depth_smooth(191 + b, V, 256, 1.0, 100.0))
        yaws = np.linspace(-0.7 + 0.1 * b, 0.5 + 0.1 * b, V)
        rts = [syn.yaw_pose(cam["P"][v:v + 1], float(y)) for v, y in enumerate(yaws)]
        RT2, RT2inv = tt(np.concatenate([r[1] for r in rts])), tt(np.concatenate([r[0] for r in rts]))
        codes, uni = tt(syn.codes(201 + b, V)), tt(np.random.RandomState(211 + b).rand(V, 1024).astype(np.float32))
        batches.append(((img, depth, K, Kinv, P, Pinv, RT2, RT2inv), codes, uni, temp))
    ref = [m.outpaint_planned(m.plan_views(*a), c, temperature=t, uniforms=u)["codes"].clone() for a, c, u, t in batches]
    other = m.outpaint_planned(m.plan_views(*batches[0][0]), batches[0][1], temperature=1.0, uniforms=batches[0][2])["codes"]
    assert not torch.equal(other, ref[0])        # (the temperature matters for these draws)
    got = []
    for a, c, u, t in batches:
        done = m.outpaint_pipelined(m.plan_views(*a), c, temperature=t, uniforms=u)
        if done is not None:
            got.append(done["codes"].clone())
    got += [o["codes"].clone() for o in m.outpaint_flush()]
    torch.cuda.synchronize()
    m.outpaint2.engine(32, 32, m.pipe_frames(V)).check()
    assert len(got) == 3
    for b in range(3):
        assert torch.equal(got[b], ref[b]), (b, int((got[b] != ref[b]).sum()))
    # a sequence cut short: the batch in flight is dropped, the next sequence starts clean
    a, c, u, t = batches[0]
    assert m.outpaint_pipelined(m.plan_views(*a), c, temperature=t, uniforms=u) is None
    m.outpaint_reset()
    assert m.outpaint_flush() == []
    a, c, u, t = batches[1]
    assert m.outpaint_pipelined(m.plan_views(*a), c, temperature=t, uniforms=u) is None      # (not the dropped batch's dict)
    left = m.outpaint_flush()
    assert len(left) == 1 and torch.equal(left[0]["codes"], ref[1])
    m.outpaint2.engine(32, 32, m.pipe_frames(V)).check()


def test_plan_views_then_outpaint_planned_equals_outpaint_views_also_across_streams():
    """The two halves of outpaint_views (host planning / device AR run) are what bench.py and the driver overlap across
    batches: planning on a side stream while another AR run is in flight must give the same views."""
    m = make_model()
    V = 3
    cam = syn.demo_cameras(V)
    img, depth = tt(syn.image(51, V, 3, 256)), tt(syn.depth_smooth(52, V, 256, 1.0, 100.0))
    rts = [syn.yaw_pose(cam["P"][v:v + 1], y) for v, y in enumerate((0.6, -0.3, 0.45))]
    RT2, RT2inv = tt(np.concatenate([r[1] for r in rts])), tt(np.concatenate([r[0] for r in rts]))
    codes, uni = tt(syn.codes(53, V)), tt(np.random.RandomState(54).rand(V, 1024).astype(np.float32))
    args = (img, depth, tt(cam["K"]), tt(cam["Kinv"]), tt(cam["P"]), tt(cam["Pinv"]), RT2, RT2inv)
    ref = m.outpaint_views(*args, codes, temperature=0.7, uniforms=uni)
    ref_codes = ref["codes"].clone()
    side, main = torch.cuda.Stream(), torch.cuda.current_stream()
    busy = m.outpaint_planned(m.plan_views(*args), codes, temperature=0.7, uniforms=uni)      # an AR run in flight ...
    with torch.cuda.stream(side):
        planned = m.plan_views(*args)                                                        # ... while the next is planned
    main.wait_stream(side)
    out = m.outpaint_planned(planned, codes, temperature=0.7, uniforms=uni)
    torch.cuda.synchronize()
    m.outpaint2.engine(32, 32, V).check()
    assert torch.equal(out["codes"], ref_codes) and torch.equal(busy["codes"], ref_codes)
    assert torch.equal(out["gen_fs"], ref["gen_fs"]) and torch.equal(out["background_mask"], ref["background_mask"])


def test_outpaint_planned_with_a_between_callback_runs_it_once_after_the_prefix_pass():
    """outpaint_planned(between=...) is the split form of the AR run -- ps_pixelcnn_ar_prefix, the callback, ps_pixelcnn_ar_columns --
    that bench.py takes under torch.distributed (the stream waits there for the previous step's gathers).  The callback runs
    exactly once, on the current stream, after the whole-grid prefix pass and before the first column launch, and the codes are
    those of the unsplit run (between=None)."""
    m = make_model()
    V = 3
    cam = syn.demo_cameras(V)
    img, depth = tt(syn.image(61, V, 3, 256)), tt(syn.depth_smooth(62, V, 256, 1.0, 100.0))
    rts = [syn.yaw_pose(cam["P"][v:v + 1], y) for v, y in enumerate((0.6, -0.45, 0.3))]
    RT2, RT2inv = tt(np.concatenate([r[1] for r in rts])), tt(np.concatenate([r[0] for r in rts]))
    codes, uni = tt(syn.codes(63, V)), tt(np.random.RandomState(64).rand(V, 1024).astype(np.float32))
    args = (img, depth, tt(cam["K"]), tt(cam["Kinv"]), tt(cam["P"]), tt(cam["Pinv"]), RT2, RT2inv)
    ref = m.outpaint_planned(m.plan_views(*args), codes, temperature=0.7, uniforms=uni)
    ref_codes = ref["codes"].clone()
    eng = m.outpaint2.engine(32, 32, V)
    calls, seen = [], {}
    real_prefix, real_columns = eng.ar_prefix, eng.ar_columns

    def between():
        calls.append(len(calls))
        seen["order"] = list(seen.get("order", [])) + ["between"]
        # on the current stream, behind the prefix pass: a marker the column launches must come after
        seen["marker"] = torch.cuda.Event(enable_timing=True)
        seen["marker"].record()

    def prefix(*a, **k):
        seen["order"] = list(seen.get("order", [])) + ["prefix"]
        return real_prefix(*a, **k)

    def columns(*a, **k):
        seen["order"] = list(seen.get("order", [])) + ["columns"]
        return real_columns(*a, **k)
    eng.ar_prefix, eng.ar_columns = prefix, columns
    try:
        out = m.outpaint_planned(m.plan_views(*args), codes, temperature=0.7, uniforms=uni, between=between)
    finally:
        eng.ar_prefix, eng.ar_columns = real_prefix, real_columns
    torch.cuda.synchronize()
    eng.check()
    assert calls == [0] and seen["order"] == ["prefix", "between", "columns"]
    assert seen["marker"].query()
    assert torch.equal(out["codes"], ref_codes)
    assert (out["codes"].cpu().numpy() != syn.codes(63, V)).any()     # the region was outpainted at all


def test_get_best_sample_batches_the_candidates_without_changing_them():
    """The num_samples candidates of a view run through the sampler together (sample-parallel frames); every candidate
    must be exactly what a run of its own produces."""
    from pixelsynth_amd.z_buffermodel import build_ar_plan
    m = _scene_model(num_samples=5)
    img = tt(syn.image(31, 1, 3, 256))
    cam = {k: tt(v) for k, v in syn.demo_cameras(1).items()}
    RTinv, RT = m.get_rt_from_rot("R", cam["P"], 2, 2)
    gen_fs, bg = m.pts_transformer.forward_justpts(img, syn.depth_from_image(img), cam["K"], cam["Kinv"], cam["P"],
                                                   cam["Pinv"], RT, RTinv)
    plan = build_ar_plan(bg, 32)
    codes = m.vqvae.encode_codes(gen_fs)
    uni = torch.rand(5, 1, 1024, generator=torch.Generator(device="cpu").manual_seed(9)).to(DEV)
    seen = []

    class D:
        def run_discriminator_one_step(self, fake, real):
            seen.append(fake)
            return {"D_Fake": fake.mean().reshape(1)}
    class C(torch.nn.Module):   # a stand-in classifier: ten "logits" from the image mean
        def forward(self, x):
            return torch.cat([x.mean().reshape(1, 1) * k for k in range(1, 11)], 1)
    m.classifier = C()
    m.sample_batch = 3                      # 5 candidates -> engine runs of 3 and 2 frames
    m.get_best_sample(plan, codes, bg, gen_fs, D(), img, uniforms=uni)
    batched = [s.clone() for s in seen]
    seen.clear()
    m.sample_batch = 1                      # one candidate per run
    m.get_best_sample(plan, codes, bg, gen_fs, D(), img, uniforms=uni)
    assert len(batched) == len(seen) == 5
    for a, b in zip(batched, seen):
        assert torch.equal(a, b)
    assert not torch.equal(batched[0], batched[1])


def test_get_best_sample_takes_the_reference_signature():
    """z_buffermodel.py:244: get_best_sample(gen_order, masks, downsampled_fs, background_mask, gen_fs, netD, input_img) with
    gen_order / masks as get_masks_for_batch returns them (the reference's layouts) gives exactly what the compact-plan form
    gives: same candidates, same winner."""
    from pixelsynth_amd.z_buffermodel import build_ar_plan
    m = _scene_model(num_samples=3)
    img = tt(syn.image(31, 1, 3, 256))
    cam = {k: tt(v) for k, v in syn.demo_cameras(1).items()}
    RTinv, RT = m.get_rt_from_rot("R", cam["P"], 2, 2)
    gen_fs, bg = m.pts_transformer.forward_justpts(img, syn.depth_from_image(img), cam["K"], cam["Kinv"], cam["P"],
                                                   cam["Pinv"], RT, RTinv)
    codes = m.vqvae.encode_codes(gen_fs)
    seen = []

    class D:
        def run_discriminator_one_step(self, fake, real):
            seen.append(fake.clone())
            return {"D_Fake": fake.mean().reshape(1)}
    class C(torch.nn.Module):
        def forward(self, x):
            return torch.cat([x.mean().reshape(1, 1) * k for k in range(1, 11)], 1)
    m.classifier = C()
    best_plan = m.get_best_sample(build_ar_plan(bg, 32), codes, bg, gen_fs, D(), img)
    first = [s for s in seen]
    seen.clear()
    mi, mu, md, gen_order = m.get_masks_for_batch(RT, cam["Pinv"], bg)            # the reference's return value (:641-701)
    best_ref = m.get_best_sample(gen_order, (mi, mu, md), codes, bg, gen_fs, D(), img)
    assert len(first) == len(seen) == 3
    for a, b in zip(first, seen):
        assert torch.equal(a, b)
    assert torch.equal(best_plan, best_ref)


def test_forward_gen_order_dispatch():
    """model_setting 'get_gen_order' (z_buffermodel.py:284-285, 594-639): forward() returns the generation order of the view,
    (B, L, 2) (row, col) by rank -- the oracle's order for the background mask the splat produces."""
    m = make_model(model_setting="get_gen_order")
    cam = {k: torch.from_numpy(v) for k, v in syn.demo_cameras(1).items()}
    img, depth = torch.from_numpy(syn.image(4, 1, 3, 256)), torch.from_numpy(syn.depth_smooth(5, 1, 256, 1.0, 100.0))
    batch = {"images": [img], "cameras": [cam], "depths": [depth]}
    loss, out = m(batch)
    assert loss is None and set(out) == {"gen_order"}
    go = out["gen_order"]
    assert go.is_cuda and tuple(go.shape) == (1, 1024, 2) and go.dtype == torch.int64
    RTinv, RT = m.get_rt_from_rot("R", tt(syn.demo_cameras(1)["P"]))
    c = {k: tt(v) for k, v in syn.demo_cameras(1).items()}
    _, bg = m.pts_transformer.forward_justpts(img.to(DEV), depth.to(DEV), c["K"], c["Kinv"], c["P"], c["Pinv"], RT, RTinv)
    want = c_oracle.masks_for_background(bg[0].cpu().numpy(), 32)["order"]
    assert np.array_equal(go[0].cpu().numpy(), want)
    # a batch that carries the target camera (process_batch's form) uses it
    batch2 = {"images": [img, img], "cameras": [cam, {"P": RT.cpu(), "Pinv": RTinv.cpu()}], "depths": [depth]}
    assert torch.equal(m(batch2)[1]["gen_order"], go)


def test_sharded_sample_ranking_two_ranks_equals_one(tmp_path):
    """get_best_sample(shard=True) under two ranks (both on cuda:0, gloo): candidates dealt round-robin, two scalars per
    candidate gathered, the winner broadcast from its owner (shape learnt from the owner) -- every rank ends with the image the
    single-process ranking keeps; forward_image routes there with opt.shard_samples; and the winner's transport alone hands a
    rank that holds nothing a tensor whose shape and dtype it could not have guessed."""
    import subprocess
    import sys
    import socket
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, PS_DRYRUN_ONE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "tests", "_shard_samples_worker.py"), str(tmp_path)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-3000:]




def test_prefix_pass_dealt_to_two_streams_gives_the_codes_of_one_stream(monkeypatch):
    """outpaint_planned deals the whole-grid prefix pass of a batch of >= 64 views to PREFIX_STREAMS frame ranges, each on a stream
    of its own (ps_pixelcnn_ar_prefix with a frame range; every range has its part of the engine's scratch and of the item sort's
    buffers).  64 views, three times over to give a race between the ranges its chance: the codes are those of one range on one
    stream, and a batch that does not divide (72 = 2 x 36 frames, not 8 per XCD) falls back to one range."""
    m = make_model()
    assert m.PREFIX_STREAMS == 2 and m._prefix_split(64) == 2 and m._prefix_split(72) == 1 and m._prefix_split(32) == 1
    V = 64
    cam = syn.demo_cameras(V)
    img, depth = tt(syn.image(71, V, 3, 256)), tt(syn.depth_smooth(72, V, 256, 1.0, 100.0))
    yaws = np.linspace(-0.7, 0.7, V)
    rts = [syn.yaw_pose(cam["P"][v:v + 1], float(y)) for v, y in enumerate(yaws)]
    RT2, RT2inv = tt(np.concatenate([r[1] for r in rts])), tt(np.concatenate([r[0] for r in rts]))
    codes, uni = tt(syn.codes(73, V)), tt(np.random.RandomState(74).rand(V, 1024).astype(np.float32))
    args = (img, depth, tt(cam["K"]), tt(cam["Kinv"]), tt(cam["P"]), tt(cam["Pinv"]), RT2, RT2inv)
    eng = m.outpaint2.engine(32, 32, V)
    ranges = []
    real_prefix = eng.ar_prefix

    def prefix(*a, **k):
        ranges.append((k.get("frame_begin"), k.get("frame_end"), torch.cuda.current_stream().cuda_stream))
        return real_prefix(*a, **k)
    monkeypatch.setenv("PS_PREFIX_STREAMS", "1")
    ref = m.outpaint_planned(m.plan_views(*args), codes, temperature=0.7, uniforms=uni)["codes"].clone()
    monkeypatch.delenv("PS_PREFIX_STREAMS")
    eng.ar_prefix = prefix
    try:
        for _ in range(3):
            del ranges[:]
            out = m.outpaint_planned(m.plan_views(*args), codes, temperature=0.7, uniforms=uni)
            torch.cuda.synchronize()
            eng.check()
            assert sorted(r[:2] for r in ranges) == [(0, 32), (32, 64)] and ranges[0][2] != ranges[1][2]
            assert torch.equal(out["codes"], ref)
    finally:
        eng.ar_prefix = real_prefix


def test_outpaint_pipelined_gives_the_codes_of_outpaint_planned_batch_by_batch(monkeypatch):
    """outpaint_pipelined with two batches in flight (both resident in one 2 V-frame handle: the wavefronts a batch has left when the next
    one arrives share its launches, lmconv.model.pack_launches); outpaint_flush runs what is left of the last batch.  Three
    different batches of 64 views in a row (the frame halves of the handle alternate: the third batch reuses the first one's), twice over:
    every batch's codes are those of outpaint_planned, bit for bit, and a batch comes back exactly one call late."""
    monkeypatch.setenv("PS_PIPE_DEPTH", "2")      # two batches in flight: a batch comes back exactly one call late
    m = make_model()
    V = 64
    assert m.pipe_depth(V) == 2
    cam = syn.demo_cameras(V)
    K, Kinv, P, Pinv = (tt(cam[k]) for k in ("K", "Kinv", "P", "Pinv"))
    batches = []
    for b in range(3):
        img, depth = tt(syn.image(81 + b, V, 3, 256)), tt(syn.depth_smooth(91 + b, V, 256, 1.0, 100.0))
        yaws = np.linspace(-0.7 + 0.1 * b, 0.5 + 0.1 * b, V)
        rts = [syn.yaw_pose(cam["P"][v:v + 1], float(y)) for v, y in enumerate(yaws)]
        RT2, RT2inv = tt(np.concatenate([r[1] for r in rts])), tt(np.concatenate([r[0] for r in rts]))
        codes, uni = tt(syn.codes(101 + b, V)), tt(np.random.RandomState(111 + b).rand(V, 1024).astype(np.float32))
        batches.append(((img, depth, K, Kinv, P, Pinv, RT2, RT2inv), codes, uni))
    ref = []
    for args, codes, uni in batches:
        m.PER_FRAME_PREFIX = False      # the reference: one prefix for the batch, one AR run per batch
        out = m.outpaint_planned(m.plan_views(*args), codes, temperature=0.7, uniforms=uni)
        ref.append(out["codes"].clone())
        m.PER_FRAME_PREFIX = True       # ... and outpaint_planned with per-frame prefixes
        assert torch.equal(m.outpaint_planned(m.plan_views(*args), codes, temperature=0.7, uniforms=uni)["codes"], ref[-1])
    torch.cuda.synchronize()
    assert not torch.equal(ref[0], ref[1]) and not torch.equal(ref[1], ref[2])
    for rep in range(3):
        # per-frame prefixes (the whole-grid pass takes every frame up to ITS first sampled position, ps_pixelcnn_ar_prefix_frames; the
        # schedule of ps_ar_wavefronts_frames) twice, then one prefix for the batch: the same codes
        m.PER_FRAME_PREFIX = rep < 2
        got = []
        for args, codes, uni in batches:
            planned = m.plan_views(*args)
            fs = planned["plan"].first_steps
            assert fs.min() == planned["plan"].first_step and fs.max() > fs.min() and planned["plan"].waves_frames[0].shape[0] == int((1024 - fs).sum())
            done = m.outpaint_pipelined(planned, codes, temperature=0.7, uniforms=uni)
            assert (done is None) == (len(got) == 0 and done is None)      # (two batches in flight: a batch comes back one call late)
            if done is not None:
                got.append(done["codes"].clone())
        got += [o["codes"].clone() for o in m.outpaint_flush()]
        assert m.outpaint_flush() == []
        torch.cuda.synchronize()
        m.outpaint2.engine(32, 32, 2 * V).check()
        assert len(got) == 3
        for b in range(3):
            assert torch.equal(got[b], ref[b]), (rep, b, int((got[b] != ref[b]).sum()))
