"""Pin the CPU oracle against the golden vectors produced by the reference itself
(tests/golden/make_golden.py) and against oracle/_ref (the reference's own Cython C, compiled).
CPU only."""
import importlib.util
import os

import numpy as np
import pytest
import torch

import cases
from oracle import build_oracle, c_oracle, lmconv_oracle as lo
from pixelsynth_amd import synthetic as syn


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


# ---------------------------------------------------------------- projection (a1, a2, a4)
def test_grid_exact(golden_dir):
    fx = load(golden_dir, "projection.npz")
    for W in (8, 16):
        assert np.array_equal(c_oracle.make_grid(W), fx[f"xyzs_W{W}"])
    assert np.array_equal(c_oracle.make_grid(256)[:, :, ::97], fx["xyzs_W256_stride97"])


def _poses():
    out = {}
    cam = syn.demo_cameras(1)
    for name in ("demo_L", "demo_R", "demo_small", "demo_identity", "demo_circle5"):
        out[name] = (cam, (1.0, 100.0))
    cam = syn.mp3d_cameras(1)
    for name in ("mp3d_yaw", "mp3d_back"):
        out[name] = (cam, (0.5, 10.0))
    return out


def test_project_pts_vs_reference(golden_dir):
    fx = load(golden_dir, "projection.npz")
    for name, (cam, (lo_, hi_)) in _poses().items():
        RT2 = fx[f"pose_{name}_RT2"]
        for W, stride in ((16, 1), (256, 61)):
            d = syn.depth_uniform(7, 2, W, lo_, hi_)
            rep = lambda m: np.repeat(m, 2, 0)
            s = c_oracle.project_pts(d, rep(cam["K"]), rep(cam["Kinv"]), rep(cam["Pinv"]), rep(RT2), W)
            ref = fx[f"proj_{name}_W{W}"]
            got = s[:, :, ::stride]
            # tolerance: torch.bmm accumulation order is unspecified: 1e-5 relative
            np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-5, err_msg=f"{name} W{W}")


def test_project_pts_eps_branch(golden_dir):
    fx = load(golden_dir, "projection.npz")
    cam = syn.demo_cameras(1)
    s = c_oracle.project_pts(fx["proj_epscase_depth"], cam["K"], cam["Kinv"], cam["Pinv"], fx["proj_epscase_RT2"], 16)
    ref = fx["proj_epscase_out"]
    bad_ref = ref[:, 2] == 10.0
    assert bad_ref.sum() > 10 and (~bad_ref).sum() > 10  # both branches exercised
    bad = s[:, 2] == 10.0
    # points within float noise of the |z| < EPS threshold may land on either side
    assert (bad != bad_ref).sum() <= 2
    ok = bad == bad_ref
    np.testing.assert_allclose(s[:, :, ok[0]], ref[:, :, ok[0]], rtol=1e-4, atol=1e-4)
    assert np.all(s[0][:, bad[0]] == np.array([[-10.0], [10.0], [10.0]], np.float32))


def test_project_pts_cumulative(golden_dir):
    fx = load(golden_dir, "projection.npz")
    cam = syn.demo_cameras(1)
    s, cloud = c_oracle.project_pts_cumulative(fx["cum_depth_new"], fx["cum_last_bg"], fx["cum_prior"], cam["K"],
                                               cam["Kinv"], cam["Pinv"], fx["cum_RT2"], fx["cum_RT3inv"], 16)
    np.testing.assert_allclose(s, fx["cum_sampler"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(cloud, fx["cum_cloud"], rtol=2e-5, atol=2e-5)
    assert (fx["cum_cloud"][0, 2] == np.float32(0.01)).sum() >= 3  # the in-place EPS write is visible
    assert (cloud[0, 2] == np.float32(0.01)).sum() == (fx["cum_cloud"][0, 2] == np.float32(0.01)).sum()
    s0, cloud0 = c_oracle.project_pts_cumulative(fx["cum0_depth"], None, None, cam["K"], cam["Kinv"], cam["Pinv"],
                                                 fx["cum_RT2"], None, 16)
    np.testing.assert_allclose(s0, fx["cum0_sampler"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(cloud0, fx["cum0_cloud"], rtol=2e-5, atol=2e-5)


# ---------------------------------------------------------------- order + masks (a8, a9)
def test_custom_idx_exact(golden_dir):
    fx = load(golden_dir, "orders_masks.npz")
    dmaps = dict(syn.distance_maps())
    assert len(fx["names"]) >= 20
    for name in fx["names"]:
        order, d = c_oracle.custom_idx(32, 32, dmaps[str(name)])
        assert np.array_equal(order, fx[f"order_{name}"].astype(np.int32)), name
        assert np.array_equal(d, dmaps[str(name)] * 10000)
        assert len({(int(r), int(c)) for r, c in order}) == 1024  # a permutation of the grid


def test_custom_idx_vs_ref_build():
    """oracle/_ref = the reference's own get_custom_order.c compiled in place."""
    so = build_oracle.build_ref()
    if so is None or not os.path.exists(so):
        pytest.skip("oracle/_ref not built and /root/reference absent")
    spec = importlib.util.spec_from_file_location("get_custom_order", so)
    gco = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gco)
    rs = np.random.RandomState(77)
    for k in range(12):
        n = [4, 8, 16, 32][k % 4]
        D = rs.randint(-4, 5, size=(n, n)).astype(np.int64)
        ref = np.asarray(gco.custom_idx(n, n, D.copy(), (n // 2, n // 2)))
        got, _ = c_oracle.custom_idx(n, n, D)
        assert np.array_equal(got, ref.astype(np.int32))


def test_unfolded_masks_exact(golden_dir):
    fx = load(golden_dir, "orders_masks.npz")
    for name in ("halfplane_x", "island", "rand0", "all_fg", "ring", "rand4"):
        order = fx[f"order_{name}"].astype(np.int32)
        for tag, dil, typ in (("A1", 1, "A"), ("B1", 1, "B"), ("B2", 2, "B")):
            m = c_oracle.unfolded_masks(order, 32, 32, 3, dil, typ)
            ref = np.unpackbits(fx[f"mask_{name}_{tag}"], axis=1)[:, :1024].astype(np.float32)
            assert np.array_equal(m[0], ref), (name, tag)
    o8 = fx["order8"].astype(np.int32)
    got, _ = c_oracle.custom_idx(8, 8, fx["order8_D"])
    assert np.array_equal(got, o8)
    for tag, dil, typ in (("A1", 1, "A"), ("B1", 1, "B"), ("B2", 2, "B")):
        assert np.array_equal(c_oracle.unfolded_masks(o8, 8, 8, 3, dil, typ), fx[f"mask8_{tag}"])


# ---------------------------------------------------------------- lmconv layers / blocks / network
def test_lmconv_layers_vs_reference(golden_dir):
    fx = load(golden_dir, "lmconv_layers.npz")
    for name, ci, co, dil, H in cases.LAYER_CASES:
        c = cases.layer_case(name)
        y = lo.lmconv(t(c["x"]), t(c["m"]), t(c["w"]), t(c["b"]), dilation=dil).numpy()
        np.testing.assert_allclose(y, fx[f"{name}_y"], rtol=1e-5, atol=1e-5, err_msg=name)


def test_blocks_vs_reference(golden_dir):
    fx = load(golden_dir, "blocks.npz")
    for skip in (0, 1):
        c = cases.gated_case(skip)
        sd = {"p." + k: t(v) for k, v in c["sd"].items()}
        y = lo.gated_resnet(sd, "p.", t(c["x"]), None if c["a"] is None else t(c["a"]), t(c["m"]))
        np.testing.assert_allclose(y.numpy(), fx[f"gr{skip}_y"], rtol=1e-5, atol=1e-5)
    c = cases.small_case()
    np.testing.assert_allclose(lo.pono(t(c["x"])).numpy(), fx["pono_y"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(lo.concat_elu(t(c["x"])).numpy(), fx["celu_y"], rtol=1e-6, atol=1e-6)
    sd = c["nin_sd"]
    y = lo.nin(t(c["x"]), t(sd["lin_a.weight_v"]), t(sd["lin_a.weight_g"]), t(sd["lin_a.bias"]))
    np.testing.assert_allclose(y.numpy(), fx["nin_y"], rtol=1e-5, atol=1e-5)


def _masks(order):
    return tuple(t(c_oracle.unfolded_masks(order, 32, 32, 3, dil, typ))
                 for dil, typ in ((1, "A"), (1, "B"), (2, "B")))


def test_network_logits_vs_reference(golden_dir):
    fx = load(golden_dir, "network.npz")
    dmaps = dict(syn.distance_maps())
    pos = fx["positions"]
    for wi in range(2):
        sd = {k: t(v) for k, v in syn.pixelcnn_state_dict(int(fx[f"net{wi}_wseed"])).items()}
        order, _ = c_oracle.custom_idx(32, 32, dmaps[str(fx[f"net{wi}_order_name"])])
        codes = syn.codes(int(fx[f"net{wi}_codes_seed"]), 1)
        x = torch.nn.functional.one_hot(t(codes), 512).permute(0, 3, 1, 2).float()
        with torch.no_grad():
            lg = lo.pixelcnn_forward(sd, x, *_masks(order))[0].reshape(512, 1024).numpy()
        # tolerance: 33 fp32 layers deep (1e-4 abs on logits)
        np.testing.assert_allclose(lg[:, pos], fx[f"net{wi}_logits_sub"], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(lg.astype(np.float64).sum(0), fx[f"net{wi}_logits_sum"], rtol=0, atol=2e-2)


def test_ar_sample_vs_reference_sample(golden_dir):
    """oracle AR loop == the reference's own sample() run on CPU (same seeds, same draws)."""
    fx = load(golden_dir, "ar_trace.npz")
    assert float(fx["causal_maxdiff"]) == 0.0  # per-step logits == one full forward on the completed grid
    sd = {k: t(v) for k, v in syn.pixelcnn_state_dict(int(fx["wseed"])).items()}
    bg32 = fx["bg32"]
    D = c_oracle.signed_distance((bg32 == 0).astype(np.uint8), (bg32 == 1).astype(np.uint8))
    assert np.array_equal(D, fx["D"])
    order, _ = c_oracle.custom_idx(32, 32, D)
    assert np.array_equal(order, fx["order"].astype(np.int32))
    masks = _masks(order)
    codes = t(syn.codes(int(fx["codes_seed"]), 1))
    n_steps = int(fx["n_steps"])
    region = lo.sample_region(order, bg32)
    assert len(region) == n_steps
    # teacher-forced: ONE forward on the reference's completed grid reproduces its per-step logits
    final = t(fx["final_codes"].astype(np.int64))[None]
    x = torch.nn.functional.one_hot(final, 512).permute(0, 3, 1, 2).float()
    with torch.no_grad():
        full = lo.pixelcnn_forward(sd, x, *masks)[0]
    got = np.stack([full[:, i, j].numpy() for i, j in region])[::4]
    np.testing.assert_allclose(got, fx["step_logits"], rtol=1e-4, atol=1e-4)
    # free-running with the reference's seeding: identical draws for the first steps (bounded: 24 steps)
    with torch.no_grad():
        data, _, chosen = lo.ar_sample_reference(sd, codes, order, bg32, masks, temperature=float(fx["temperature"]),
                                                 seed=int(fx["seed"]), max_steps=24)
    ref_codes = np.array([fx["final_codes"][i, j] for i, j in region[:24]])
    assert np.array_equal(chosen, ref_codes)


def test_wavefront_schedule_respects_every_dependency():
    """ps_ar_wavefronts_capped (host): every walked column appears once and every column it reads -- the 3x3 neighbours at
    dilation 1 and 2 that are earlier in the order, i.e. the open taps of the three kernel masks, restated here with
    plain loops -- sits in an EARLIER wave.  Without a capacity a column's wave is exactly one more than the latest
    wave among those; with the capacity of a launch (128) no wave is larger, and the list scheduler still needs no
    more waves than the dependency depth or the column count force.  Far shallower than the walk either way."""
    from pixelsynth_amd.lmconv.model import wavefronts
    bgs = syn.background_masks(256)
    names = ["right_half", "half_plus_island", "ragged", "all", "top_band"]
    infos = [c_oracle.masks_for_background(bgs[n], 32) for n in names] * 3          # 15 frames: waves above the capacity
    F_ = len(infos)
    order_loc = np.stack([(i["order"][:, 0] * 32 + i["order"][:, 1]) for i in infos]).astype(np.int32)
    G, L = 32, 1024
    for first in (0, 320, 1000, 1024):
        depth = None
        for cap in (0, 128):
            cols, wave_start = wavefronts(order_loc, G, G, first, max_cols=cap)
            cols = cols.numpy()
            assert cols.shape == (F_ * (L - first), 2) and wave_start[0] == 0 and wave_start[-1] == cols.shape[0]
            sizes = np.diff(wave_start)
            assert (sizes > 0).all() and (cap == 0 or sizes.size == 0 or sizes.max() <= cap)
            wave_of = {}
            for w in range(len(wave_start) - 1):
                for f, i in cols[wave_start[w]:wave_start[w + 1]]:
                    assert (f, i) not in wave_of
                    wave_of[(int(f), int(i))] = w
            assert len(wave_of) == cols.shape[0]
            for f in range(F_):
                rank = np.empty(L, int)
                rank[order_loc[f]] = np.arange(L)
                masks = [infos[f][k].reshape(9, L) for k in ("mask_init", "mask_undilated", "mask_dilated")]
                for i in range(first, L):
                    q = int(order_loc[f][i])
                    r, c = divmod(q, G)
                    dep = -1
                    for dil, mk in ((1, masks[0]), (1, masks[1]), (2, masks[2])):
                        for t in range(9):
                            rr, cc = r + (t // 3 - 1) * dil, c + (t % 3 - 1) * dil
                            if t == 4 or not (0 <= rr < G and 0 <= cc < G):
                                continue
                            p = rr * G + cc
                            assert (mk[t, q] != 0) == (rank[p] < i)          # open tap <=> earlier in the order
                            if rank[p] < i and rank[p] >= first:
                                dep = max(dep, wave_of[(f, int(rank[p]))])
                    assert wave_of[(f, i)] == dep + 1 if cap == 0 else wave_of[(f, i)] > dep
            n_waves = len(wave_start) - 1
            if cap == 0:
                depth = n_waves
            else:
                assert n_waves <= max(depth, -(-cols.shape[0] // cap)) + depth // 8   # close to the lower bound
        if first == 320:
            assert depth < (L - first) // 4
