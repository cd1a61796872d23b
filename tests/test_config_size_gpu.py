"""GPU: results ASSERTED at the sizes BASELINE.json's configurations name (round-3 review: "the configs are exercised, not verified at
size").  C5's 128-view batch: views equal the same views run alone, splat against the oracle, teacher-forced logits against the torch-fp32
twin of the network; C2 in idx-emitting mode at B = 32 against the oracle's rasterizer; gen_paired_img / gen_two_imgs on the device
against what the reference's own loops produced (tests/golden/poses.npz); column launches under a second stream's back-to-back GEMMs."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import c_oracle, lmconv_oracle as lo
from pixelsynth_amd import synthetic as syn

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DEV = torch.device("cuda", 0)
HERE = os.path.dirname(os.path.abspath(__file__))


def tt(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.fixture(scope="module")
def c5():
    """BASELINE config 5 as bench.py builds it: 8 Matterport-shaped sources x 16 views = 128 views, one pass of the hot path."""
    import bench
    model = bench.build_model(DEV)
    d, host = bench.make_inputs(0, 128, DEV)
    out = model.outpaint_views(d["img"], d["depth"], d["K"], d["Kinv"], d["P"], d["Pinv"], d["RT2"], d["RT2inv"], d["codes"], temperature=0.7,
                               uniforms=d["uniforms"])
    return model, d, host, out


def test_c5_views_of_the_128_view_batch_equal_the_views_run_alone(c5):
    """Four views of C5's batch (throughput-form column launches, one wavefront schedule for 128 frames, the batch's common prefix)
    against the same views rendered alone (latency form, their own schedule and prefix): sampled codes, reprojected features and
    background masks bit for bit."""
    model, d, host, out = c5
    assert out["codes"].shape == (128, 32, 32)
    for v in (0, 15, 77, 127):
        sl = slice(v, v + 1)
        one = model.outpaint_views(d["img"][sl], d["depth"][sl], d["K"][sl], d["Kinv"][sl], d["P"][sl], d["Pinv"][sl], d["RT2"][sl], d["RT2inv"][sl],
                                   d["codes"][sl], temperature=0.7, uniforms=d["uniforms"][sl])
        assert torch.equal(one["codes"][0], out["codes"][v]), v
        assert torch.equal(one["gen_fs"][0], out["gen_fs"][v]) and torch.equal(one["background_mask"][0], out["background_mask"][v]), v
        assert one["plan"].first_step >= out["plan"].first_step
    n = out["plan"].n_sampled
    assert n.min() >= 1 and n.max() > 300 and (out["codes"].cpu().numpy() != host["codes"]).any()   # the sweep's ends outpaint a lot


def test_c5_splat_of_two_views_vs_the_oracle(c5):
    """Two views of the batch against the C oracle (project -> rasterize K = 128, r = 4 -> alpha-composite -> 13x13 dilation)."""
    model, d, host, out = c5
    for v in (3, 127):
        cam = {k: a[v:v + 1] for k, a in host["cam"].items()}
        sampler = c_oracle.project_pts(host["depth"][v:v + 1], cam["K"], cam["Kinv"], cam["Pinv"], host["RT2"][v:v + 1], 256)
        ref = c_oracle.splat_forward(np.ascontiguousarray(sampler.transpose(0, 2, 1)), host["img"][v:v + 1].reshape(1, 3, -1), 256)
        assert np.array_equal(out["background_mask"][v].cpu().numpy(), ref["bg"][0]), v
        np.testing.assert_allclose(out["gen_fs"][v].cpu().numpy(), ref["feat"][0], rtol=0, atol=1e-6)


def test_c5_teacher_forced_logits_of_three_views_vs_the_torch_twin(c5):
    """The whole 128-view AR run again, teacher-forced with the codes it sampled, returning the logits every location was decided
    from: for three views -- the first of the batch, one from the middle of a source's sweep (its columns sit in the middle of the
    chain tiles of a wavefront), the last -- they must equal ONE full forward of the torch-fp32 twin on the completed grid (the causality property the
    incremental evaluation rests on: tests/golden/ar_trace.npz, causal_maxdiff = 0) within 1e-4, at every one of the 1024 locations."""
    model, d, host, out = c5
    plan = out["plan"]
    eng = model.outpaint2.engine(32, 32, 128)
    done = out["codes"].reshape(128, 1024).to(torch.int32).contiguous()
    c = d["codes"].reshape(128, 1024).to(torch.int32).contiguous().clone()
    logits = eng.ar_run(c, plan.order_loc, plan.region, plan.mask_init, plan.mask_undilated, plan.mask_dilated, temperature=0.7, forced=done,
                        first_step=plan.first_step, want_logits=True, waves=plan.waves)
    eng.check()
    assert torch.equal(c, done)
    sd = {k: torch.from_numpy(a) for k, a in syn.pixelcnn_state_dict(0).items()}
    for v in (0, 70, 127):
        x = torch.nn.functional.one_hot(done[v].long().cpu().view(1, 32, 32), 512).permute(0, 3, 1, 2).float()
        masks = [m[v:v + 1].cpu() for m in (plan.mask_init, plan.mask_undilated, plan.mask_dilated)]
        with torch.no_grad():
            ref = lo.pixelcnn_forward(sd, x, *masks)[0].reshape(512, 1024).t().numpy()     # (location, class)
        np.testing.assert_allclose(logits[v].cpu().numpy(), ref, rtol=1e-4, atol=1e-4, err_msg=f"view {v}")


def test_c5_the_path_the_headline_times_at_its_size(c5):
    """bench.py's timed region runs z_buffermodel.outpaint_pipelined: up to four 128-view batches resident in one 512-frame handle, every
    column launch taking what is left of each batch's current wavefront, per-frame prefixes, the prefix pass on two streams,
    `between=` set (where bench.py collects the previous step's gathers).  Three DIFFERENT C5 batches (bench.make_inputs, ranks 0 .. 2)
    through it at C5's size: every batch's codes equal outpaint_planned's bit for bit, batch 0's are the fixture's -- the codes whose
    logits test_c5_teacher_forced_logits_of_three_views_vs_the_torch_twin holds against the torch twin and whose splat the oracle
    checks -- and the kernels the bench line's roofline names are the ones that ran (launch counters of the 512-frame engine)."""
    import bench
    model, d0, host0, out0 = c5
    V = 128
    batches = [d0] + [bench.make_inputs(r, V, DEV)[0] for r in (1, 2)]
    front = lambda d: model.plan_views(d["img"], d["depth"], d["K"], d["Kinv"], d["P"], d["Pinv"], d["RT2"], d["RT2inv"])
    ref = []
    for d in batches:
        ref.append(model.outpaint_planned(front(d), d["codes"], temperature=0.7, uniforms=d["uniforms"])["codes"].clone())
    model.outpaint2.engine(32, 32, V).check()
    assert torch.equal(ref[0], out0["codes"])                      # (outpaint_views is plan_views + outpaint_planned)
    assert not torch.equal(ref[0], ref[1]) and not torch.equal(ref[1], ref[2])
    eng = model.outpaint2.engine(32, 32, model.pipe_frames(V))
    before = eng.launch_counts()
    streams, between_calls = set(), []
    real_prefix = eng.ar_prefix

    def prefix(*a, **k):
        streams.add(torch.cuda.current_stream().cuda_stream)
        return real_prefix(*a, **k)
    eng.ar_prefix = prefix
    try:
        for rep in range(2):      # twice: the second time every half of the handle has held another batch before
            got = []
            for d in batches:
                done = model.outpaint_pipelined(front(d), d["codes"], temperature=0.7, uniforms=d["uniforms"],
                                                between=lambda: between_calls.append(torch.cuda.current_stream().cuda_stream))
                if done is not None:
                    got.append(done["codes"].clone())
            got += [o["codes"].clone() for o in model.outpaint_flush()]
            torch.cuda.synchronize()
            eng.check()
            assert len(got) == 3
            for b in range(3):
                assert torch.equal(got[b], ref[b]), (rep, b, int((got[b] != ref[b]).sum()))
    finally:
        eng.ar_prefix = real_prefix
    assert len(streams) == 2 and len(between_calls) == 6           # two frame ranges on two streams; between= ran in every step
    ran = {k: v - before[k] for k, v in eng.launch_counts().items()}
    assert ran["k_column_tp"] > 0 and ran["k_column_tp"] + ran["k_column_tp8"] > 90, ran           # the throughput form: full launches of 16-column tiles (+ the fill's and the flush's smaller ones)
    assert ran["k_gemm_ws<0>"] == ran["k_gemm_ws<1>"] == 6 * 2 * 14 and ran["k_gemm_ws<2>"] == 6 * 2 * 4, ran   # 6 prefix passes x 2 ranges
    assert ran["k_gemm_wg"] == 0 and ran["k_gemm"] == 0 and ran["k_column"] == 0, ran


def test_c2_idx_emitting_mode_at_batch_32_vs_the_oracle():
    """BASELINE config 2 at its size: 32 clouds of 65 536 points through the splatter with the PyTorch3D-shaped debug tensors
    (B,S,S,K=128): idx / zbuf / dist of four frames bit-exact against the oracle's rasterizer fed the same projected points
    (idx in the packed numbering b * N + n)."""
    from pixelsynth_amd.projection.z_buffer_manipulator import PtsManipulator
    import types
    B, S = 32, 256
    opt = types.SimpleNamespace(W=S, use_rgb_features=True, splatter="xyblending", learn_default_feature=True, radius=4, pp_pixel=128, tau=1.0,
                                rad_pow=2, accumulation="alphacomposite", background_smoothing_kernel_size=13)
    pm = PtsManipulator(S, C=3, opt=opt).to(DEV)
    cam = syn.demo_cameras(B)
    img, depth = syn.image(41, B, 3, S), syn.depth_smooth(42, B, S, 1.0, 100.0)
    RT2, RT2inv = np.empty((B, 4, 4), np.float32), np.empty((B, 4, 4), np.float32)
    for b in range(B):
        inv, rt = syn.yaw_pose(cam["P"][b:b + 1], -0.6 + 1.2 * b / (B - 1))
        RT2[b], RT2inv[b] = rt[0], inv[0]
    s = pm.project_pts(tt(depth).view(B, 1, -1), tt(cam["K"]), tt(cam["Kinv"]), tt(cam["P"]), tt(cam["Pinv"]), tt(RT2), tt(RT2inv))
    pc = s.permute(0, 2, 1).contiguous()
    feat, bg, idx, zbuf, dist = pm.splatter(pc.clone(), tt(img).view(B, 3, -1), return_debug=True)
    assert tuple(idx.shape) == (B, S, S, 128) and idx.dtype == torch.int32
    N = S * S
    for b in (0, 9, 22, 31):
        sampler = c_oracle.project_pts(depth[b:b + 1], cam["K"][b:b + 1], cam["Kinv"][b:b + 1], cam["Pinv"][b:b + 1], RT2[b:b + 1], S)
        assert np.array_equal(sampler, s[b:b + 1].cpu().numpy())                      # the projection itself, bit for bit
        ref = c_oracle.splat_forward(np.ascontiguousarray(sampler.transpose(0, 2, 1)), img[b:b + 1].reshape(1, 3, -1), S)
        want = ref["idx"][0].astype(np.int64)
        want = np.where(want >= 0, want + b * N, want)                               # packed index of frame b of the batch
        assert np.array_equal(idx[b].cpu().numpy(), want.astype(np.int32)), b
        assert np.array_equal(dist[b].cpu().numpy(), ref["dist"][0]), b
        assert np.array_equal(zbuf[b].cpu().numpy(), ref["zbuf"][0]), b
        assert np.array_equal(bg[b].cpu().numpy(), ref["bg"][0]), b
        np.testing.assert_allclose(feat[b].cpu().numpy(), ref["feat"][0], rtol=0, atol=1e-6)


def _model(**kw):
    from test_zbuffermodel_gpu import make_model
    return make_model(**kw)


def test_gen_paired_img_on_the_device():
    """model_setting gen_paired_img (models/z_buffermodel.py:294-295, process_batch :127-130): the TARGET view comes with the batch --
    its image is handed back as OutputImg and its pose replaces get_rt_from_rot.  With the pose the reference records for direction
    'R' at rotation 0.6 (tests/golden/poses.npz) as the batch's second camera, every output equals the gen_img run that derives
    that pose itself."""
    fx = np.load(os.path.join(HERE, "golden", "poses.npz"), allow_pickle=True)
    rows = [str(r) for r in fx["pose_cases"]]
    ci = rows.index("gen_img|0|0.6|R||")
    cam = syn.demo_cameras(1)
    RT, RTinv = fx[f"pose{ci}_demo_RT"], fx[f"pose{ci}_demo_RTinv"]
    img, target = syn.image(4, 1, 3, 256), syn.image(9, 1, 3, 256)
    common = {"depths": [torch.from_numpy(syn.depth_smooth(5, 1, 256, 1.0, 100.0))], "codes": torch.from_numpy(syn.codes(6, 1))}
    cam0 = {k: torch.from_numpy(v) for k, v in cam.items()}
    m1 = _model(model_setting="gen_img", direction="R", rotation=0.6)
    _, want = m1.forward_image({"images": [torch.from_numpy(img)], "cameras": [cam0], **common})
    np.testing.assert_allclose(m1.get_rt_from_rot("R", cam0["P"].to(DEV))[1].cpu().numpy(), RT, rtol=0, atol=1e-7)   # the mirror derives that very pose
    m2 = _model(model_setting="gen_paired_img")
    cam1 = {"P": torch.from_numpy(RT), "Pinv": torch.from_numpy(RTinv), "K": cam0["K"], "Kinv": cam0["Kinv"]}
    _, got = m2.forward_image({"images": [torch.from_numpy(img), torch.from_numpy(target)], "cameras": [cam0, cam1], **common})
    assert torch.equal(got["OutputImg"].cpu(), torch.from_numpy(target)) and "OutputImg" not in want
    for k in ("InputImg", "PredDepthImg", "ForegroundImg", "FeaturesImg", "PredCodes"):
        assert torch.equal(got[k], want[k]), k
    m2.outpaint2.engine(32, 32, 1).check()
    assert 0.05 < float(got["ForegroundImg"].mean()) < 0.95


def test_gen_two_imgs_on_the_device_follows_the_reference_schedule():
    """model_setting gen_two_imgs (models/z_buffermodel.py:425-453): the direction comes with the batch (mapping[5] = 'UR'), the far
    view is rendered first, then the view half way back.  The real renderer, VQ-VAE and AR sampler run on the device; the poses
    handed to the cumulative reprojection are those the REFERENCE's own forward_scene recorded (tests/golden/poses.npz, 'two_imgs'),
    every frame is rendered from the frame the reference renders it from, and the output keys are the reference's."""
    from test_zbuffermodel_gpu import _record_scene, _scene_batch, _scene_model
    fx = np.load(os.path.join(HERE, "golden", "poses.npz"), allow_pickle=True)
    tag = "two_imgs"
    seq, dirs, split, setting = [str(v) for v in fx[f"scene_{tag}_opts"]]
    assert setting == "gen_two_imgs"
    m = _scene_model(model_setting="gen_two_imgs", directions=None, num_split=int(split), sequential_outpainting=False)
    batch = _scene_batch()
    batch["direction"] = torch.tensor(5)
    calls, out = _record_scene(m, batch)
    assert len(calls) == int(fx[f"scene_{tag}_n"])
    frames = [batch["images"][0].to(DEV)]
    for k, (a, r) in enumerate(calls):
        src1, depth, K, Kinv, RT1, RT1inv, RT2, RT2inv, prior, src2, last_bg, RT3inv = a
        np.testing.assert_allclose(RT1.cpu().numpy(), fx[f"scene_{tag}_{k}_RT1"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(RT2.cpu().numpy(), fx[f"scene_{tag}_{k}_RT2"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(RT2inv.cpu().numpy(), fx[f"scene_{tag}_{k}_RT2inv"], rtol=1e-5, atol=1e-5)
        ref3 = fx[f"scene_{tag}_{k}_RT3inv"]
        assert (RT3inv is None) == (ref3.size == 0)
        if RT3inv is not None:
            np.testing.assert_allclose(RT3inv.cpu().numpy(), ref3, rtol=1e-5, atol=1e-5)
        assert torch.equal(src1, frames[int(fx[f"scene_{tag}_{k}_src"])])           # rendered from the frame the reference renders it from
        made = [t for t in out if t.startswith("FeaturesImg_") and torch.equal(out[t], r[0])]
        assert len(made) == 1
        frames.append(out["PredImg_" + made[0][len("FeaturesImg_"):]])
    keys = sorted(k for k in out if k.startswith("PredImg_"))
    assert keys == [str(k) for k in fx[f"scene_{tag}_pred_keys"]]
    assert sorted(out.keys()) == [str(k) for k in fx[f"scene_{tag}_all_keys"]]
    for k in keys:
        assert tuple(out[k].shape) == (1, 3, 256, 256) and torch.isfinite(out[k]).all()
    m.outpaint2.engine(32, 32, 1).check()


def test_column_launches_under_a_foreign_stream_of_gemms(c5):
    """A column launch keeps one workgroup per compute unit resident, and its workgroups wait (bounded) for each other inside the
    launch: kernels of ANOTHER stream that hold compute units may delay it, they must not break it.  C5's AR run -- ~90 column
    launches, twice, so ~180 and with the prefix passes > 200 launches -- while a second stream runs back-to-back 4096^3 fp32 GEMMs:
    no bounded wait gives up (engine.check()) and the codes are those of the undisturbed run."""
    model, d, host, out = c5
    plan = out["plan"]
    eng = model.outpaint2.engine(32, 32, 128)
    side = torch.cuda.Stream()
    a, b = torch.randn(4096, 4096, device=DEV), torch.randn(4096, 4096, device=DEV)
    ev = lambda: torch.cuda.Event(enable_timing=True)
    g0, g1, r0, r1 = ev(), ev(), ev(), ev()
    main = torch.cuda.current_stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        g0.record(side)              # in front of the first GEMM
        for _ in range(100):         # ~100 x 1 ms of GEMM kernels alone: as long as the two AR runs even when they share the device
            c_ = a @ b
        g1.record(side)              # behind the last one
    got = []
    r0.record(main)
    for _ in range(2):
        c = d["codes"].reshape(128, 1024).to(torch.int32).contiguous().clone()
        eng.ar_run(c, plan.order_loc, plan.region, plan.mask_init, plan.mask_undilated, plan.mask_dilated, temperature=0.7, uniforms=d["uniforms"],
                   first_step=plan.first_step, waves=plan.waves)
        got.append(c)
    r1.record(main)
    eng.check()
    torch.cuda.synchronize()
    # the two intervals on the device's own clock: [g0, g1] the GEMMs, [r0, r1] the AR runs.  They must have shared the device for
    # at least 80 % of the AR runs' duration (a host-side `not stop.query()` right after enqueueing proves nothing: the host runs
    # far ahead of the GPU).
    ar = r0.elapsed_time(r1)
    start_gap = g0.elapsed_time(r0)              # AR start relative to GEMM start (ms; >= 0 up to enqueue jitter)
    gemm = g0.elapsed_time(g1)
    overlap = max(0.0, min(start_gap + ar, gemm) - max(start_gap, 0.0))
    assert overlap >= 0.8 * ar, (overlap, ar, gemm, start_gap)
    assert torch.isfinite(c_).all()
    for c in got:
        assert torch.equal(c.view(128, 32, 32), out["codes"])
