"""GPU parity tests of the locally-masked PixelCNN path (through the C ABI).

Tolerances (fp32 throughout, MFMA f32 = fma-chain numerics): single layer 2e-5 abs; gated block 5e-5;
full-network logits 1e-4 abs against the reference golden vectors (33 fp32 layers deep);
incremental (column) evaluation vs whole-grid evaluation of the SAME kernels: bit-exact."""
import os
import types

import numpy as np
import pytest
import torch

import cases
from oracle import c_oracle, lmconv_oracle as lo
from pixelsynth_amd import synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def tt(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def norm_op(_c):
    from pixelsynth_amd.lmconv.layers import PONO
    return PONO()


def make_net(seed=0):
    from pixelsynth_amd.lmconv.model import OurPixelCNN
    net = OurPixelCNN(nr_resnet=2, nr_filters=80, input_channels=512, nr_logistic_mix=10, kernel_size=(3, 3),
                      max_dilation=2, weight_norm=False, feature_norm_op=norm_op, dropout_prob=0, conv_bias=True,
                      conv_mask_weight=False, rematerialize=False, binarize=False).eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in syn.pixelcnn_state_dict(seed).items()}, strict=True)
    return net.to(DEV)


def masks_for(order):
    return tuple(tt(c_oracle.unfolded_masks(order, 32, 32, 3, dil, typ)) for dil, typ in ((1, "A"), (1, "B"), (2, "B")))


def test_lmconv_layers_vs_reference_golden(golden_dir):
    from pixelsynth_amd.lmconv.locally_masked_convolution import locally_masked_conv2d
    fx = np.load(os.path.join(golden_dir, "lmconv_layers.npz"))
    for name, ci, co, dil, H in cases.LAYER_CASES:
        c = cases.layer_case(name)
        layer = locally_masked_conv2d(ci, co, kernel_size=(3, 3), dilation=dil, bias=True).to(DEV)
        with torch.no_grad():
            layer.weight.copy_(tt(c["w"]))
            layer.bias.copy_(tt(c["b"]))
            # the reference layout: mask repeated over input channels, (B*Ci, 9, L)
            mrep = tt(c["m"]).unsqueeze(1).repeat(1, ci, 1, 1).view(c["B"] * ci, 9, H * H)
            y = layer(tt(c["x"]), mrep)
            y2 = layer(tt(c["x"]), tt(c["m"]))  # compact (B,9,L) form
        np.testing.assert_allclose(y.cpu().numpy(), fx[f"{name}_y"], rtol=1e-5, atol=2e-5, err_msg=name)
        assert torch.equal(y, y2)


def test_lmconv_broadcast_mask_and_no_bias():
    from pixelsynth_amd.lmconv.locally_masked_convolution import lmconv_forward
    rs = np.random.RandomState(0)
    x = rs.randn(3, 7, 6, 9).astype(np.float32)  # non-square grid, ragged channels
    w = rs.randn(5, 7, 3, 3).astype(np.float32)
    m = (rs.rand(1, 9, 54) > 0.5).astype(np.float32)
    y = lmconv_forward(tt(x), tt(m), tt(w), None, dilation=2).cpu()
    ref = lo.lmconv(torch.from_numpy(x), torch.from_numpy(m), torch.from_numpy(w), None, dilation=2)
    np.testing.assert_allclose(y.numpy(), ref.numpy(), rtol=1e-5, atol=2e-5)


def test_lmconv_fractional_mask_values():
    """Mask values other than 0 / 1 (the reference multiplies the unfolded input by whatever the mask holds,
    locally_masked_convolution.py:24-27): the kernel's zero-row shortcut for 0/1 masks must not be taken."""
    from pixelsynth_amd.lmconv.locally_masked_convolution import lmconv_forward
    rs = np.random.RandomState(3)
    for ci, co in ((160, 80), (7, 5)):
        x = rs.randn(2, ci, 8, 8).astype(np.float32)
        w = (rs.randn(co, ci, 3, 3) * 0.1).astype(np.float32)
        b = rs.randn(co).astype(np.float32)
        m = rs.rand(2, 9, 64).astype(np.float32)
        m[m < 0.3] = 0.0     # closed taps among fractional ones
        m[0, :, :16] = 1.0   # and one tile whose values are all 0 / 1 next to tiles whose values are not
        m[0, 4, :16] = 0.0
        y = lmconv_forward(tt(x), tt(m), tt(w), tt(b), dilation=1).cpu()
        ref = lo.lmconv(torch.from_numpy(x), torch.from_numpy(m), torch.from_numpy(w), torch.from_numpy(b), dilation=1)
        np.testing.assert_allclose(y.numpy(), ref.numpy(), rtol=1e-5, atol=5e-5)


def test_blocks_vs_reference_golden(golden_dir):
    from pixelsynth_amd.lmconv.layers import PONO, gated_resnet, nin
    from pixelsynth_amd.lmconv.locally_masked_convolution import locally_masked_conv2d
    from pixelsynth_amd.lmconv.utils import concat_elu
    fx = np.load(os.path.join(golden_dir, "blocks.npz"))
    conv_op = lambda cin, cout: locally_masked_conv2d(cin, cout, kernel_size=(3, 3), bias=True, mask_weight=False)
    for skip in (0, 1):
        c = cases.gated_case(skip)
        blk = gated_resnet(80, conv_op, norm_op, concat_elu, skip_connection=skip, dropout_prob=0).eval()
        blk.load_state_dict({k: torch.from_numpy(v) for k, v in c["sd"].items()}, strict=True)
        blk = blk.to(DEV)
        with torch.no_grad():
            y = blk(tt(c["x"]), a=None if c["a"] is None else tt(c["a"]), mask=tt(c["m"]))
        np.testing.assert_allclose(y.cpu().numpy(), fx[f"gr{skip}_y"], rtol=1e-5, atol=5e-5)
    c = cases.small_case()
    np.testing.assert_allclose(PONO()(tt(c["x"])).cpu().numpy(), fx["pono_y"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(concat_elu(tt(c["x"])).cpu().numpy(), fx["celu_y"], rtol=1e-6, atol=1e-6)
    lin = nin(80, 512).eval()
    lin.load_state_dict({k: torch.from_numpy(v) for k, v in c["nin_sd"].items()}, strict=True)
    with torch.no_grad():
        y = lin.to(DEV)(tt(c["x"]))
    np.testing.assert_allclose(y.cpu().numpy(), fx["nin_y"], rtol=1e-5, atol=2e-5)


def test_network_logits_vs_reference_golden(golden_dir):
    fx = np.load(os.path.join(golden_dir, "network.npz"))
    dmaps = dict(syn.distance_maps())
    pos = fx["positions"]
    for wi in range(2):
        net = make_net(int(fx[f"net{wi}_wseed"]))
        order, _ = c_oracle.custom_idx(32, 32, dmaps[str(fx[f"net{wi}_order_name"])])
        mi, mu, md = masks_for(order)
        codes = syn.codes(int(fx[f"net{wi}_codes_seed"]), 1)
        x = torch.nn.functional.one_hot(tt(codes), 512).permute(0, 3, 1, 2).float()
        # reference calling convention: list input, masks repeated per input channel
        rep = lambda m, c: m[0:1].repeat(c, 1, 1).view(-1, 9, 1024)
        with torch.no_grad():
            logits = net([x, rep(mi, 513), rep(mu, 160), rep(md, 80)], sample=True)
            layered = net._forward_layers(x, True, mi, mu, md)
        lg = logits[0].reshape(512, 1024).cpu().numpy()
        np.testing.assert_allclose(lg[:, pos], fx[f"net{wi}_logits_sub"], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(lg.astype(np.float64).sum(0), fx[f"net{wi}_logits_sum"], rtol=0, atol=2e-2)
        np.testing.assert_allclose(layered.cpu().numpy(), logits.cpu().numpy(), rtol=1e-4, atol=1e-4)


def test_network_zero_input_and_batch():
    """All-zero (not yet sampled) locations + a batch of different images/orders vs the torch twin."""
    net = make_net(2)
    sd = {k: torch.from_numpy(v) for k, v in syn.pixelcnn_state_dict(2).items()}
    dmaps = dict(syn.distance_maps())
    names = ["corner", "rand2", "island_l1"]
    eng = net.engine(32, 32, 3)
    codes = syn.codes(5, 3).reshape(3, 1024).astype(np.int32)
    codes[0, 500:] = -1
    codes[2, ::3] = -1
    orders = [c_oracle.custom_idx(32, 32, dmaps[n])[0] for n in names]
    ms = [np.concatenate([c_oracle.unfolded_masks(o, 32, 32, 3, dil, typ) for o in orders]) for dil, typ in
          ((1, "A"), (1, "B"), (2, "B"))]
    logits = eng.forward(tt(codes), *[tt(m) for m in ms]).cpu().numpy()
    for b in range(3):
        x = torch.zeros(1, 512, 1024)
        valid = codes[b] >= 0
        x[0, codes[b][valid], np.nonzero(valid)[0]] = 1
        with torch.no_grad():
            ref = lo.pixelcnn_forward(sd, x.view(1, 512, 32, 32), *[torch.from_numpy(m[b:b + 1]) for m in ms])
        np.testing.assert_allclose(logits[b], ref[0].numpy(), rtol=1e-4, atol=1e-4)


def test_whole_grid_launch_forms_agree_bit_for_bit():
    """The launch forms of the whole-grid pass -- one wave per slot with the slots added up by the post op; one wave
    walking all slots and adding them up itself (taken from 8192 (tile, channel block) pairs on); the workgroup form with
    the input rows shared through LDS and the post op in its tail, at every tile size -- give identical logits, and match
    the torch twin."""
    net = make_net(4)
    sd = {k: torch.from_numpy(v) for k, v in syn.pixelcnn_state_dict(4).items()}
    dmaps = dict(syn.distance_maps())
    eng = net.engine(32, 32, 2)
    codes = syn.codes(9, 2).reshape(2, 1024).astype(np.int32)
    codes[1, 700:] = -1
    orders = [c_oracle.custom_idx(32, 32, dmaps[n])[0] for n in ("corner", "rand2")]
    ms = [np.concatenate([c_oracle.unfolded_masks(o, 32, 32, 3, dil, typ) for o in orders]) for dil, typ in
          ((1, "A"), (1, "B"), (2, "B"))]
    run = lambda: eng.forward(tt(codes), *[tt(m) for m in ms]).cpu()
    big = 1 << 30
    eng.set_tuning(gemm_merge_min=0, gemm_wg_min=big)
    merged = run()
    eng.set_tuning(gemm_merge_min=big)
    split = run()
    assert torch.equal(merged, split)
    eng.set_tuning(gemm_merge_min=0, gemm_wg_min=1, gemm_ws=0)   # k_gemm_wg: four-wave workgroups, input rows shared through LDS, post op fused (large launches)
    wg = run()
    assert torch.equal(wg, split)
    eng.set_tuning(gemm_ws=7, gemm_ws_min=1)          # k_gemm_ws: 64 items per workgroup, the WEIGHTS shared through LDS
    n0 = eng.launch_counts()
    assert torch.equal(run(), split)
    n1 = eng.launch_counts()
    assert n1["k_gemm_ws<0>"] - n0["k_gemm_ws<0>"] == 14 and n1["k_gemm_ws<1>"] - n0["k_gemm_ws<1>"] == 14 and n1["k_gemm_ws<2>"] - n0["k_gemm_ws<2>"] == 4
    assert n1["k_gemm_wg"] == n0["k_gemm_wg"] > 0     # (the form under test is the one that ran; k_gemm_wg ran for `wg` above)
    eng.set_tuning(gemm_ws=0)
    for ti in ((2, 2, 2), (1, 4, 4), (2, 4, 2)):      # item tiles per workgroup: conv_out / conv_input / dilated
        eng.set_tuning(wg_ti_out=ti[0], wg_ti_in=ti[1], wg_ti_dil=ti[2])
        assert torch.equal(run(), split), ti
    # the products' items in natural (frame, rank) order instead of grouped by open-tap set: a closed tap adds an exact zero, so
    # which items share a tile changes no bit -- in either gemm form
    eng.set_tuning(item_sort=0)
    assert torch.equal(run(), split)
    eng.set_tuning(gemm_wg_min=big)
    assert torch.equal(run(), split)
    eng.set_tuning(item_sort=1)
    assert torch.equal(run(), split)
    eng.set_tuning(gemm_wg_min=1, gemm_ws=7)
    assert torch.equal(run(), split)                  # (k_gemm_ws over one sort of all frames)
    eng.set_tuning(gemm_merge_min=8192, gemm_wg_min=256, gemm_ws_min=1024, wg_ti_out=1, wg_ti_in=2, wg_ti_dil=2, item_sort=2, gemm_ws=7)   # (the defaults)
    x = torch.zeros(1, 512, 1024)
    x[0, codes[0], np.arange(1024)] = 1
    with torch.no_grad():
        ref = lo.pixelcnn_forward(sd, x.view(1, 512, 32, 32), *[torch.from_numpy(m[0:1]) for m in ms])
    np.testing.assert_allclose(merged[0].numpy(), ref[0].numpy(), rtol=1e-4, atol=1e-4)


def test_workgroup_form_with_fractional_masks_loads_them():
    """The sorted item list carries every item's SET of open taps, and the workgroup form takes a 0 / 1 mask's values from it
    instead of loading them.  Masks with values that are neither 0 nor 1 (the generic layer supports them; the reference never
    produces them) are flagged per item and loaded: k_gemm_wg / k_gemm_ws against k_gemm, which always loads, bit for bit."""
    net = make_net(4)
    dmaps = dict(syn.distance_maps())
    eng = net.engine(32, 32, 2)
    codes = syn.codes(19, 2).reshape(2, 1024).astype(np.int32)
    orders = [c_oracle.custom_idx(32, 32, dmaps[n])[0] for n in ("rand2", "corner")]
    ms = [np.concatenate([c_oracle.unfolded_masks(o, 32, 32, 3, dil, typ) for o in orders]) for dil, typ in
          ((1, "A"), (1, "B"), (2, "B"))]
    rs = np.random.RandomState(3)
    for m_ in ms[1:]:                                   # a third of the open taps of frame 1 get a fractional value; frame 0 stays 0 / 1
        scale = np.where(rs.rand(*m_[1].shape) < 0.33, 0.25 + 0.5 * rs.rand(*m_[1].shape), 1.0).astype(np.float32)
        m_[1] = m_[1] * scale
    run = lambda: eng.forward(tt(codes), *[tt(m) for m in ms]).cpu()
    big = 1 << 30
    eng.set_tuning(gemm_merge_min=0, gemm_wg_min=big)
    ref = run()
    eng.set_tuning(gemm_wg_min=1, gemm_ws=0, gemm_ws_min=1)
    assert torch.equal(run(), ref)
    eng.set_tuning(gemm_ws=7)
    assert torch.equal(run(), ref)
    eng.set_tuning(gemm_ws=2)   # (one kind through k_gemm_ws, the others through k_gemm_wg)
    assert torch.equal(run(), ref)
    eng.set_tuning(gemm_ws=0, item_sort=0)
    assert torch.equal(run(), ref)
    eng.set_tuning(gemm_merge_min=8192, gemm_wg_min=256, gemm_ws_min=1024, item_sort=2, gemm_ws=7)   # (the defaults)
    frac = ms[1][1][ms[1][1] > 0]
    assert ((frac != 1.0).mean() > 0.2) and torch.isfinite(ref).all()


@pytest.mark.parametrize("F_,first", [(7, 600), (3, 905), (33, 333), (16, 500)])
def test_workgroup_gemm_form_under_the_prefix_cone_is_bit_identical(F_, first):
    """k_gemm_wg (the whole-grid products with the receptive-field rows staged in LDS once per workgroup) against k_gemm on
    the prefix pass of an AR run -- rank-ordered items, ragged last tiles, per-stage start ranks of the dependency cone,
    frames with different orders: sampled codes and the logits of every walked location identical bit for bit."""
    from pixelsynth_amd.lmconv.model import wavefronts
    net = make_net(5)
    eng = net.engine(32, 32, F_)
    bgs = syn.background_masks(256)
    names = ["right_half", "half_plus_island", "ragged", "top_band"]
    infos = [c_oracle.masks_for_background(bgs[names[b % 4]], 32) for b in range(F_)]
    order_loc = np.stack([(i["order"][:, 0] * 32 + i["order"][:, 1]) for i in infos]).astype(np.int32)
    reg = np.zeros((F_, 1024), np.uint8)
    rs = np.random.RandomState(F_)
    for b in range(F_):
        walked = order_loc[b][first:]
        reg[b, walked[rs.rand(walked.size) < 0.7]] = 1
        reg[b, order_loc[b][first]] = 1
    ms = [tt(np.concatenate([i[k] for i in infos])) for k in ("mask_init", "mask_undilated", "mask_dilated")]
    codes0 = syn.codes(23, F_).reshape(F_, 1024).astype(np.int32)
    u = tt(np.random.RandomState(5).rand(F_, 1024).astype(np.float32))
    waves = wavefronts(order_loc, 32, 32, first, DEV)
    eng.set_tuning(prefix_cone_force=1, gemm_merge_min=0, gemm_ws_min=1)

    def run(wg_min, item_sort=2, gemm_ws=0):
        eng.set_tuning(gemm_wg_min=wg_min, item_sort=item_sort, gemm_ws=gemm_ws)
        c = tt(codes0.copy())
        lg = eng.ar_run(c, tt(order_loc), tt(reg), *ms, temperature=0.7, uniforms=u, first_step=first, want_logits=True, waves=waves)
        eng.check()
        return c, lg
    c_wg, l_wg = run(1)
    c_ref, l_ref = run(1 << 30)
    c_nat, l_nat = run(1, item_sort=0)        # items in natural order / one sort over all frames (16 frames: default = one per XCD share)
    c_one, l_one = run(1 << 30, item_sort=1)
    c_old, l_old = run(1, gemm_ws=1)          # k_gemm_ws (weights through LDS, 64 items per workgroup) instead of k_gemm_wg (rows through LDS)
    eng.set_tuning(prefix_cone_force=0, gemm_merge_min=8192, gemm_wg_min=256, gemm_ws_min=1024, item_sort=2, gemm_ws=0)
    assert torch.equal(c_wg, c_ref) and torch.equal(c_nat, c_ref) and torch.equal(c_one, c_ref) and torch.equal(c_old, c_ref)
    walked = np.zeros((F_, 1024), bool)
    for b in range(F_):
        walked[b, order_loc[b][first:]] = True
    sel = torch.from_numpy(walked).to(DEV)
    assert torch.equal(l_wg[sel], l_ref[sel]) and torch.equal(l_nat[sel], l_ref[sel]) and torch.equal(l_one[sel], l_ref[sel])
    assert torch.equal(l_old[sel], l_ref[sel])
    assert (c_wg.cpu().numpy()[reg == 1] != codes0[reg == 1]).any()


def _ar_setup(fx):
    bg32 = fx["bg32"]
    order = fx["order"].astype(np.int32)
    region = lo.sample_region(order, bg32)
    order_loc = (order[:, 0] * 32 + order[:, 1]).astype(np.int32)[None]
    reg = np.zeros((1, 1024), np.uint8)
    reg[0, region[:, 0] * 32 + region[:, 1]] = 1
    first = int(np.nonzero(reg[0][order_loc[0]])[0][0])
    return order, region, order_loc, reg, first


def test_ar_teacher_forced_vs_reference_sample_trace(golden_dir):
    """Teacher-force the codes the REFERENCE's own sample() drew (CPU run, tests/golden/ar_trace.npz) and compare
    the logits each position was decided from with the logits the reference's loop saw at that step."""
    fx = np.load(os.path.join(golden_dir, "ar_trace.npz"))
    net = make_net(int(fx["wseed"]))
    eng = net.engine(32, 32, 1)
    order, region, order_loc, reg, first = _ar_setup(fx)
    assert len(region) == int(fx["n_steps"]) and first > 0
    mi, mu, md = masks_for(order)
    codes0 = syn.codes(int(fx["codes_seed"]), 1).reshape(1, 1024).astype(np.int32)
    final = fx["final_codes"].astype(np.int32).reshape(1, 1024)
    assert np.array_equal(final[reg == 0], codes0[reg == 0])  # observed codes untouched by the reference
    outs = []
    for fs in (first, 0):
        c = tt(codes0.copy())
        out = eng.ar_run(c, tt(order_loc), tt(reg), mi, mu, md, temperature=0.7, forced=tt(final), first_step=fs,
                         want_logits=True)
        eng.check()
        torch.cuda.synchronize()
        assert np.array_equal(c.cpu().numpy(), final)
        outs.append(out.cpu().numpy())
    got = np.stack([outs[0][0, i * 32 + j] for i, j in region])[::4]
    np.testing.assert_allclose(got, fx["step_logits"], rtol=1e-4, atol=1e-4)
    # positions walked by both runs carry identical bits (first_step only skips a prefix)
    walked = order_loc[0][first:]
    assert np.array_equal(outs[0][0, walked], outs[1][0, walked])
    # incremental column evaluation == ONE whole-grid forward on the completed grid, bit for bit
    full = eng.forward(tt(final), mi, mu, md)[0].reshape(512, 1024).t().cpu().numpy()
    assert np.array_equal(outs[1][0], full)


@pytest.mark.parametrize("cap", [128, 0, 1024])
def test_ar_wavefront_run_teacher_forced_vs_reference_sample_trace(golden_dir, cap):
    """The product path itself (ps_pixelcnn_ar_run_waves, what sample() / outpaint_views run) teacher-forced with the codes the
    REFERENCE's own sample() drew: the logits every sampled position was decided from against the logits the reference's
    loop saw at that step (1e-4), directly -- not through the position-by-position walk."""
    from pixelsynth_amd.lmconv.model import wavefronts
    fx = np.load(os.path.join(golden_dir, "ar_trace.npz"))
    net = make_net(int(fx["wseed"]))
    eng = net.engine(32, 32, 1)
    order, region, order_loc, reg, first = _ar_setup(fx)
    mi, mu, md = masks_for(order)
    codes0 = syn.codes(int(fx["codes_seed"]), 1).reshape(1, 1024).astype(np.int32)
    final = fx["final_codes"].astype(np.int32).reshape(1, 1024)
    c = tt(codes0.copy())
    waves = wavefronts(order_loc, 32, 32, first, DEV, max_cols=cap)
    out = eng.ar_run(c, tt(order_loc), tt(reg), mi, mu, md, temperature=0.7, forced=tt(final), first_step=first,
                     want_logits=True, waves=waves)
    eng.check()
    torch.cuda.synchronize()
    assert np.array_equal(c.cpu().numpy(), final)
    got = np.stack([out.cpu().numpy()[0, i * 32 + j] for i, j in region])[::4]
    np.testing.assert_allclose(got, fx["step_logits"], rtol=1e-4, atol=1e-4)
    assert len(waves[1]) - 1 < len(region)      # fewer launches than sampled positions: the schedule is a real wavefront one


@pytest.mark.parametrize("copies", [64, 72])
def test_throughput_form_teacher_forced_vs_reference_sample_trace(golden_dir, copies):
    """The HEADLINE kernel against the reference directly: the trace frame replicated `copies` times (same order, same forced
    codes) makes every wavefront wider than the latency form takes, so the column launches run as k_column_tp (16-column MFMA
    chain tiles; 72 copies leave ragged tiles); the logits every sampled position of EVERY copy was decided from are compared
    with the logits the reference's own sample() loop saw at that step (tests/golden/ar_trace.npz, models/lmconv/sample.py:54-66)."""
    from pixelsynth_amd.lmconv.model import wavefronts
    fx = np.load(os.path.join(golden_dir, "ar_trace.npz"))
    net = make_net(int(fx["wseed"]))
    eng = net.engine(32, 32, copies)
    order, region, order_loc, reg, first = _ar_setup(fx)
    mi, mu, md = masks_for(order)
    rep = lambda a: np.ascontiguousarray(np.repeat(a, copies, 0))
    codes0 = rep(syn.codes(int(fx["codes_seed"]), 1).reshape(1, 1024).astype(np.int32))
    final = rep(fx["final_codes"].astype(np.int32).reshape(1, 1024))
    c = tt(codes0.copy())
    waves = wavefronts(rep(order_loc), 32, 32, first, DEV, max_cols=1024)
    widths = np.diff(waves[1])
    assert widths[widths > 128].sum() > 0.8 * widths.sum()   # (nearly) all columns of this run go through throughput-form launches
    out = eng.ar_run(c, tt(rep(order_loc)), tt(rep(reg)), mi, mu, md, temperature=0.7, forced=tt(final), first_step=first,
                     want_logits=True, waves=waves)
    eng.check()
    torch.cuda.synchronize()
    assert np.array_equal(c.cpu().numpy(), final)
    got = out.cpu().numpy()
    sel = np.array([i * 32 + j for i, j in region])[::4]
    for f in range(copies):
        np.testing.assert_allclose(got[f, sel], fx["step_logits"], rtol=1e-4, atol=1e-4, err_msg=f"copy {f}")
    assert np.array_equal(got[0], got[copies - 1])   # and the copies agree bit for bit, wherever they sit in a tile


def test_ar_fused_sampling_inverse_cdf_and_determinism():
    net = make_net(3)
    F_ = 3
    eng = net.engine(32, 32, F_)
    bgs = syn.background_masks(256)
    names = ["right_half", "half_plus_island", "ragged"]
    infos = [c_oracle.masks_for_background(bgs[n], 32) for n in names]
    order_loc = np.stack([(i["order"][:, 0] * 32 + i["order"][:, 1]) for i in infos]).astype(np.int32)
    reg = np.stack([i["bg32"].reshape(-1) for i in infos]).astype(np.uint8)
    ms = [tt(np.concatenate([i[k] for i in infos])) for k in ("mask_init", "mask_undilated", "mask_dilated")]
    first = min(int(np.nonzero(reg[b][order_loc[b]])[0][0]) for b in range(F_))
    codes0 = syn.codes(9, F_).reshape(F_, 1024).astype(np.int32)
    u = np.random.RandomState(4).rand(F_, 1024).astype(np.float32)
    runs = []
    for _ in range(2):
        c = tt(codes0.copy())
        out = eng.ar_run(c, tt(order_loc), tt(reg), *ms, temperature=0.7, uniforms=tt(u), first_step=first,
                         want_logits=True)
        eng.check()
        torch.cuda.synchronize()
        runs.append((c.cpu().numpy(), out.cpu().numpy()))
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1])  # deterministic
    codes, logits = runs[0]
    assert np.array_equal(codes[reg == 0], codes0[reg == 0])
    assert (codes[reg == 1] >= 0).all() and (codes[reg == 1] < 512).all()
    # each draw is the inverse CDF of softmax(logits/T) at u (float64 host check, 1e-5 slack at bin edges)
    bad = 0
    for b in range(F_):
        for q in np.nonzero(reg[b])[0]:
            p = np.exp((logits[b, q].astype(np.float64) / np.float32(0.7)) - (logits[b, q] / np.float32(0.7)).max())
            cdf = np.cumsum(p) / p.sum()
            k = codes[b, q]
            lo_, hi_ = (cdf[k - 1] if k > 0 else 0.0), cdf[k]
            if not (lo_ - 1e-5 <= u[b, q] <= hi_ + 1e-5):
                bad += 1
    assert bad == 0
    # batch independence: image 1 alone gives the same codes
    eng1 = make_net(3).engine(32, 32, 1)
    c = tt(codes0[1:2].copy())
    eng1.ar_run(c, tt(order_loc[1:2]), tt(reg[1:2]), *[m[1:2].contiguous() for m in ms], temperature=0.7,
                uniforms=tt(u[1:2]), first_step=first)
    eng1.check()
    assert np.array_equal(c.cpu().numpy()[0], codes[1])
    # the sampled grid is self-consistent: one whole-grid forward reproduces the logits it was drawn from
    full = eng.forward(tt(codes), *ms).reshape(F_, 512, 1024).permute(0, 2, 1).cpu().numpy()
    for b in range(F_):
        walked = order_loc[b][first:]
        assert np.array_equal(full[b][walked], logits[b][walked])


def test_sample_dropin_modes_agree():
    """sample() with the reference's signature: 'reference' (full forward per step) and 'multinomial'
    (incremental) modes see identical logits, hence draw identical codes with torch.multinomial."""
    from pixelsynth_amd.lmconv.sample import sample
    net = make_net(1)
    bg = np.zeros((256, 256), bool)
    bg[:, 232:] = True  # 3 code columns -> 96 sampled positions
    info = c_oracle.masks_for_background(bg, 32)
    order = [info["order"].astype(np.int64)]
    masks = [tt(info[k]).repeat(c, 1, 1) for k, c in (("mask_init", 513), ("mask_undilated", 160), ("mask_dilated", 80))]
    codes = tt(syn.codes(31, 1))
    bm = tt(info["bg32"].astype(np.float32))[None]
    res = {}
    for mode in ("reference", "multinomial", "fused"):
        args = types.SimpleNamespace(num_classes=512, dataloader_seed=0, ar_mode=mode)
        with torch.no_grad():
            data, loss = sample(net, order, *masks, codes, [3, 32, 32], args, seed=1, temperature=0.7, background_mask=bm)
        assert data.shape == (1, 512, 32, 32) and torch.all(data.sum(1) == 1)
        res[mode] = data.argmax(1).cpu().numpy()
        assert np.isfinite(float(loss))
    keep = info["bg32"] == 0
    for mode in res:
        assert np.array_equal(res[mode][0][keep], syn.codes(31, 1)[0][keep])  # observed codes are never touched
    assert np.array_equal(res["reference"], res["multinomial"])
    assert (res["fused"][0][~keep] != syn.codes(31, 1)[0][~keep]).mean() > 0.9  # really resampled


@pytest.mark.parametrize("F_", [17, 40, 300])
def test_ar_many_frames_layouts(F_):
    """More frames than one 16-frame tile / than the XCD-split layout takes (32) / than the chip has CUs (256):
    the column launches must stay exact (sampled grid reproduced bit for bit by one whole-grid forward) and no
    in-launch wait may run out.  Short walk (last 24 order positions) to keep the test quick."""
    net = make_net(3)
    eng = net.engine(32, 32, F_)
    bgs = syn.background_masks(256)
    names = ["right_half", "half_plus_island", "ragged"]
    infos = [c_oracle.masks_for_background(bgs[names[b % 3]], 32) for b in range(F_)]
    order_loc = np.stack([(i["order"][:, 0] * 32 + i["order"][:, 1]) for i in infos]).astype(np.int32)
    first = 1000
    reg = np.zeros((F_, 1024), np.uint8)
    for b in range(F_):
        reg[b, order_loc[b][first:]] = 1                      # sample the last 24 positions of every order
    ms = [tt(np.concatenate([i[k] for i in infos])) for k in ("mask_init", "mask_undilated", "mask_dilated")]
    codes0 = syn.codes(11, F_).reshape(F_, 1024).astype(np.int32)
    u = np.random.RandomState(6).rand(F_, 1024).astype(np.float32)
    c = tt(codes0.copy())
    out = eng.ar_run(c, tt(order_loc), tt(reg), *ms, temperature=0.7, uniforms=tt(u), first_step=first, want_logits=True)
    eng.check()
    codes, logits = c.cpu().numpy(), out.cpu().numpy()
    assert np.array_equal(codes[reg == 0], codes0[reg == 0]) and (codes[reg == 1] >= 0).all() and (codes[reg == 1] < 512).all()
    full = eng.forward(tt(codes), *ms).reshape(F_, 512, 1024).permute(0, 2, 1).cpu().numpy()
    for b in range(F_):
        walked = order_loc[b][first:]
        assert np.array_equal(full[b][walked], logits[b][walked]), b
    # frame independence across layouts: frame 3 of this batch = the same frame run alone
    eng1 = make_net(3).engine(32, 32, 1)
    c1 = tt(codes0[3:4].copy())
    eng1.ar_run(c1, tt(order_loc[3:4]), tt(reg[3:4]), *[m[3:4].contiguous() for m in ms], temperature=0.7,
                uniforms=tt(u[3:4]), first_step=first)
    eng1.check()
    assert np.array_equal(c1.cpu().numpy()[0], codes[3])


@pytest.mark.parametrize("H,W", [(16, 16), (8, 12), (64, 64)])
def test_other_grid_sizes(H, W):
    """The engine is not tied to PixelSynth's 32x32 code grid: whole-grid logits against the torch-fp32 oracle and
    incremental == whole-grid (bit for bit) on a square and a non-square grid, random generation order.  64 x 64 = 4096
    locations is the largest grid the item sort and the prefix cone take (ranks and locations in 12 bits)."""
    net = make_net(5)
    F_, L = 2, H * W
    eng = net.engine(H, W, F_)
    rs = np.random.RandomState(H * 100 + W)
    orders = [np.stack(np.unravel_index(rs.permutation(L), (H, W)), 1).astype(np.int32) for _ in range(F_)]
    masks = [np.concatenate([c_oracle.unfolded_masks(o, H, W, 3, dil, typ) for o in orders]) for dil, typ in ((1, "A"), (1, "B"), (2, "B"))]
    ms = [tt(m) for m in masks]
    codes0 = rs.randint(0, 512, size=(F_, L)).astype(np.int32)
    logits = eng.forward(tt(codes0), *ms).cpu().numpy()
    sd = {k: torch.from_numpy(v) for k, v in syn.pixelcnn_state_dict(5).items()}
    x = torch.nn.functional.one_hot(torch.from_numpy(codes0.astype(np.int64)).view(F_, H, W), 512).permute(0, 3, 1, 2).float()
    with torch.no_grad():
        want = lo.pixelcnn_forward(sd, x, *[torch.from_numpy(m) for m in masks]).numpy()
    np.testing.assert_allclose(logits, want, rtol=1e-4, atol=1e-4)
    order_loc = np.stack([o[:, 0] * W + o[:, 1] for o in orders]).astype(np.int32)
    first = L // 2
    reg = np.zeros((F_, L), np.uint8)
    for b in range(F_):
        reg[b, order_loc[b][first:]] = 1
    u = rs.rand(F_, L).astype(np.float32)
    c = tt(codes0.copy())
    out = eng.ar_run(c, tt(order_loc), tt(reg), *ms, temperature=0.7, uniforms=tt(u), first_step=first, want_logits=True)
    eng.check()
    codes = c.cpu().numpy()
    full = eng.forward(tt(codes), *ms).reshape(F_, 512, L).permute(0, 2, 1).cpu().numpy()
    got = out.cpu().numpy()
    for b in range(F_):
        walked = order_loc[b][first:]
        assert np.array_equal(full[b][walked], got[b][walked])


@pytest.mark.parametrize("F_,first,cap", [(1, 512, 128), (5, 320, 128), (20, 600, 128), (40, 900, 128), (40, 900, 0), (3, 700, 16),
                                          # waves of more than 128 columns run in the throughput form (k_column_tp: 16-column MFMA
                                          # chain tiles, one wave per neighbour item): full and ragged tiles, 1 .. 64 tiles per launch
                                          (40, 600, 1024), (33, 800, 200), (64, 700, 1024), (24, 560, 130), (100, 960, 1024)])
def test_ar_wavefront_schedule_is_bit_identical_to_the_walk(F_, first, cap):
    """The wavefront schedule (ps_ar_wavefronts_capped + ps_pixelcnn_ar_run_waves: columns that do not depend on each
    other in one launch) must reproduce the position-by-position walk bit for bit -- sampled codes AND the logits each
    location was decided from -- for frames with different orders, wave counts and first positions: list-scheduled waves
    of a launch's capacity (128), of a smaller one, and the pure dependency levels (cap 0), whose oversized waves the
    launcher splits."""
    from pixelsynth_amd.lmconv.model import wavefronts
    net = make_net(3)
    eng = net.engine(32, 32, F_)
    bgs = syn.background_masks(256)
    names = ["right_half", "half_plus_island", "ragged", "all", "top_band"]
    infos = [c_oracle.masks_for_background(bgs[names[b % 5]], 32) for b in range(F_)]
    order_loc = np.stack([(i["order"][:, 0] * 32 + i["order"][:, 1]) for i in infos]).astype(np.int32)
    reg = np.zeros((F_, 1024), np.uint8)
    rs = np.random.RandomState(F_)
    for b in range(F_):
        walked = order_loc[b][first:]
        reg[b, walked[rs.rand(walked.size) < 0.8]] = 1        # some walked positions stay observed, as in real views
        reg[b, order_loc[b][first]] = 1
    ms = [tt(np.concatenate([i[k] for i in infos])) for k in ("mask_init", "mask_undilated", "mask_dilated")]
    codes0 = syn.codes(13, F_).reshape(F_, 1024).astype(np.int32)
    u = tt(np.random.RandomState(8).rand(F_, 1024).astype(np.float32))
    c_walk, c_wave = tt(codes0.copy()), tt(codes0.copy())
    l_walk = eng.ar_run(c_walk, tt(order_loc), tt(reg), *ms, temperature=0.7, uniforms=u, first_step=first, want_logits=True)
    eng.check()
    waves = wavefronts(order_loc, 32, 32, first, DEV, max_cols=cap)   # cap 0: pure levels, oversized ones split by the launcher
    n_waves = len(waves[1]) - 1
    ncols = F_ * (1024 - first)
    assert waves[0].shape[0] == ncols and n_waves < max((1024 - first) // 3 + 16, -(-ncols // max(cap, 1)) + 8)   # depth- or capacity-bound
    l_wave = eng.ar_run(c_wave, tt(order_loc), tt(reg), *ms, temperature=0.7, uniforms=u, first_step=first, want_logits=True,
                        waves=waves)
    eng.check()
    assert torch.equal(c_walk, c_wave)
    assert torch.equal(l_walk, l_wave)
    assert (c_wave.cpu().numpy()[reg == 1] != codes0[reg == 1]).any()


@pytest.mark.parametrize("ahead", ["0", "5", "12", "31"])
def test_throughput_form_look_ahead_depths_are_bit_identical_to_the_walk(ahead):
    """k_column_tp's neighbour role computes the slots of the first tp_ahead stages of launch i + 1 during launch i, behind
    launch i's chain tiles (double-buffered slots and counters, `done` counters published by the chain tiles, write-through
    stores).  Whatever the depth -- none, the default, all but the last stage -- and for full, ragged and oversized (split)
    wavefronts, codes and logits must equal the position-by-position walk bit for bit; the run is repeated on the same handle,
    so the never-reset counters of both parities are exercised past their first use."""
    from pixelsynth_amd.lmconv.model import wavefronts
    net = make_net(3)
    for F_, first, cap in [(72, 640, 1024), (40, 820, 200), (100, 990, 0)]:
        eng = net.engine(32, 32, F_, slot=100 + int(ahead))     # a handle of its own: the depth is fixed by the first column launch
        if eng.get_tuning("tp_ahead") != int(ahead):
            eng.set_tuning(tp_ahead=int(ahead))
        bgs = syn.background_masks(256)
        names = ["right_half", "half_plus_island", "ragged", "all", "top_band"]
        infos = [c_oracle.masks_for_background(bgs[names[b % 5]], 32) for b in range(F_)]
        order_loc = np.stack([(i["order"][:, 0] * 32 + i["order"][:, 1]) for i in infos]).astype(np.int32)
        reg = np.zeros((F_, 1024), np.uint8)
        rs = np.random.RandomState(F_)
        for b in range(F_):
            walked = order_loc[b][first:]
            reg[b, walked[rs.rand(walked.size) < 0.8]] = 1
            reg[b, order_loc[b][first]] = 1
        ms = [tt(np.concatenate([i[k] for i in infos])) for k in ("mask_init", "mask_undilated", "mask_dilated")]
        codes0 = syn.codes(13, F_).reshape(F_, 1024).astype(np.int32)
        u = tt(np.random.RandomState(8).rand(F_, 1024).astype(np.float32))
        c_walk = tt(codes0.copy())
        l_walk = net.engine(32, 32, F_).ar_run(c_walk, tt(order_loc), tt(reg), *ms, temperature=0.7, uniforms=u, first_step=first,
                                                want_logits=True, waves=wavefronts(order_loc, 32, 32, first, DEV, max_cols=128))
        waves = wavefronts(order_loc, 32, 32, first, DEV, max_cols=cap)
        assert (np.diff(waves[1]) > 128).sum() >= 3          # several throughput-form launches in a row
        for rep in range(2):
            c_wave = tt(codes0.copy())
            l_wave = eng.ar_run(c_wave, tt(order_loc), tt(reg), *ms, temperature=0.7, uniforms=u, first_step=first, want_logits=True,
                                waves=waves)
            eng.check()
            assert torch.equal(c_walk, c_wave), (F_, first, cap, rep)
            assert torch.equal(l_walk, l_wave), (F_, first, cap, rep)
        # stage-affine neighbour XCDs (every neighbour XCD owns a fixed share of the stages; the default) against all XCDs walking all
        # stages: scheduling only -- the same items, the same counters -- so it can be switched on a live handle, and nothing may change
        eng.set_tuning(tp_affine=1 - eng.get_tuning("tp_affine"))   # (the other setting than the default's)
        c_aff = tt(codes0.copy())
        l_aff = eng.ar_run(c_aff, tt(order_loc), tt(reg), *ms, temperature=0.7, uniforms=u, first_step=first, want_logits=True, waves=waves)
        eng.check()
        eng.set_tuning(tp_affine=1 - eng.get_tuning("tp_affine"))
        assert torch.equal(c_walk, c_aff) and torch.equal(l_walk, l_aff), (F_, first, cap, "affine")
        # chain tiles of 8 columns (k_column_tp8: a wave's post phase is one column; launches whose tiles fit tp_ct8_xcds XCDs) against
        # tiles of 16 (k_column_tp) -- and both with the spare compute units of the chain XCDs dealt to the neighbour shares (tp_fill)
        ct8 = eng.get_tuning("tp_ct8_xcds")
        for other, fill in ((0 if ct8 else 3, 0), (3, 1), (4, 0)):
            eng.set_tuning(tp_ct8_xcds=other, tp_fill=fill)
            c_ct = tt(codes0.copy())
            l_ct = eng.ar_run(c_ct, tt(order_loc), tt(reg), *ms, temperature=0.7, uniforms=u, first_step=first, want_logits=True, waves=waves)
            eng.check()
            assert torch.equal(c_walk, c_ct) and torch.equal(l_walk, l_ct), (F_, first, cap, "tiles of 8 columns", other, fill)
        eng.set_tuning(tp_ct8_xcds=ct8, tp_fill=0)


@pytest.mark.parametrize("ahead", ["0", "5", "28"])
def test_latency_form_look_ahead_depths_are_bit_identical_to_the_walk(ahead):
    """The same look-ahead in the latency form (k_column_la: the four-wave items of nbr_role for the NEXT launch's columns behind this
    launch's chain workgroups, which publish their stores through the store / control waves): depths other than the default,
    launches of 1 .. 128 columns, oversized wavefronts split by the launcher, a run repeated on the same handle, and a walk
    position by position (k_column, one use count for all stages) on that handle in between."""
    from pixelsynth_amd.lmconv.model import wavefronts
    net = make_net(3)
    for F_, first, cap in [(5, 320, 128), (20, 700, 16), (40, 900, 0)]:
        eng = net.engine(32, 32, F_, slot=200 + int(ahead))     # a handle of its own: the depth is fixed by the first column launch
        if eng.get_tuning("col_ahead") != int(ahead):
            eng.set_tuning(col_ahead=int(ahead))
        bgs = syn.background_masks(256)
        names = ["right_half", "half_plus_island", "ragged", "all", "top_band"]
        infos = [c_oracle.masks_for_background(bgs[names[b % 5]], 32) for b in range(F_)]
        order_loc = np.stack([(i["order"][:, 0] * 32 + i["order"][:, 1]) for i in infos]).astype(np.int32)
        reg = np.zeros((F_, 1024), np.uint8)
        rs = np.random.RandomState(F_)
        for b in range(F_):
            walked = order_loc[b][first:]
            reg[b, walked[rs.rand(walked.size) < 0.8]] = 1
            reg[b, order_loc[b][first]] = 1
        ms = [tt(np.concatenate([i[k] for i in infos])) for k in ("mask_init", "mask_undilated", "mask_dilated")]
        codes0 = syn.codes(13, F_).reshape(F_, 1024).astype(np.int32)
        u = tt(np.random.RandomState(8).rand(F_, 1024).astype(np.float32))
        waves = wavefronts(order_loc, 32, 32, first, DEV, max_cols=cap)
        outs = []
        for kind in ("waves", "walk", "waves"):
            c = tt(codes0.copy())
            lg = eng.ar_run(c, tt(order_loc), tt(reg), *ms, temperature=0.7, uniforms=u, first_step=first, want_logits=True,
                            waves=waves if kind == "waves" else None)
            eng.check()
            outs.append((c, lg))
        for c, lg in outs[1:]:
            assert torch.equal(outs[0][0], c), (F_, first, cap)
            assert torch.equal(outs[0][1], lg), (F_, first, cap)


def test_ar_run_waves_rejects_a_schedule_of_another_run():
    """The host checks the schedule's shape (column count, monotone wave_start); entries that name frames / positions
    outside the run are caught on the device and reported by check(), without touching memory out of bounds."""
    from pixelsynth_amd.lmconv.model import wavefronts
    net = make_net(3)
    F_, first = 2, 1000
    eng = net.engine(32, 32, F_)
    info = c_oracle.masks_for_background(syn.background_masks(256)["right_half"], 32)
    order_loc = np.stack([(info["order"][:, 0] * 32 + info["order"][:, 1])] * F_).astype(np.int32)
    reg = np.zeros((F_, 1024), np.uint8)
    reg[:, order_loc[0][first:]] = 1
    ms = [tt(np.concatenate([info[k]] * F_)) for k in ("mask_init", "mask_undilated", "mask_dilated")]
    u = tt(np.random.RandomState(1).rand(F_, 1024).astype(np.float32))
    cols, wave_start = wavefronts(order_loc, 32, 32, first, DEV)
    with pytest.raises(RuntimeError, match="schedule holds"):
        eng.ar_run(tt(syn.codes(1, F_).reshape(F_, 1024).astype(np.int32)), tt(order_loc), tt(reg), *ms, temperature=0.7,
                   uniforms=u, first_step=first, waves=(cols, np.ascontiguousarray(wave_start[:-1])))   # a wave short
    bad = cols.clone()
    bad[0, 0] = 7                                                       # a frame this run does not have
    eng.ar_run(tt(syn.codes(1, F_).reshape(F_, 1024).astype(np.int32)), tt(order_loc), tt(reg), *ms, temperature=0.7,
               uniforms=u, first_step=first, waves=(bad, wave_start))
    with pytest.raises(RuntimeError, match="outside this run"):
        eng.check()


@pytest.mark.parametrize("F_,first", [(5, 320), (12, 600), (3, 905), (40, 640)])   # (40 frames: throughput-form column launches)
def test_prefix_pass_evaluates_only_what_is_read(F_, first):
    """The whole-grid pass over the observed prefix skips, stage by stage, the items nobody reads (k_prefix_starts): the
    device's (33, F) table of start ranks equals the numpy restatement (oracle/prefix_cone_oracle.py), and codes and
    logits are bit-identical to a run that evaluates the whole prefix at every stage (tuning value prefix_full)."""
    import ctypes
    from oracle import prefix_cone_oracle as pc
    from pixelsynth_amd import _lib
    from pixelsynth_amd.lmconv.model import wavefronts
    net = make_net(5)
    eng = net.engine(32, 32, F_)
    bgs = syn.background_masks(256)
    names = ["right_half", "half_plus_island", "ragged", "top_band"]
    infos = [c_oracle.masks_for_background(bgs[names[b % 4]], 32) for b in range(F_)]
    order_loc = np.stack([(i["order"][:, 0] * 32 + i["order"][:, 1]) for i in infos]).astype(np.int32)
    reg = np.zeros((F_, 1024), np.uint8)
    rs = np.random.RandomState(F_)
    for b in range(F_):
        walked = order_loc[b][first:]
        reg[b, walked[rs.rand(walked.size) < 0.7]] = 1
        reg[b, order_loc[b][first]] = 1
    ms = [tt(np.concatenate([i[k] for i in infos])) for k in ("mask_init", "mask_undilated", "mask_dilated")]
    codes0 = syn.codes(21, F_).reshape(F_, 1024).astype(np.int32)
    u = tt(np.random.RandomState(4).rand(F_, 1024).astype(np.float32))
    waves = wavefronts(order_loc, 32, 32, first, DEV, max_cols=1024 if F_ >= 24 else None)

    def run():
        c = tt(codes0.copy())
        lg = eng.ar_run(c, tt(order_loc), tt(reg), *ms, temperature=0.7, uniforms=u, first_step=first, want_logits=True, waves=waves)
        eng.check()
        return c, lg
    eng.set_tuning(prefix_cone_force=1)   # (a run that returns logits evaluates the whole prefix otherwise)
    c_cone, l_cone = run()
    eng.set_tuning(prefix_cone_force=0)
    # the table the prefix pass just used
    L = _lib.lib()
    L.ps_pixelcnn_debug_cache.restype = ctypes.c_void_p
    ptr = L.ps_pixelcnn_debug_cache(eng.handle, 5, 0)
    assert ptr
    raw = type("Raw", (), {"__cuda_array_interface__": {"shape": (pc.N_EVAL, F_), "typestr": "<i4", "data": (ptr, False), "version": 2}})()
    torch.cuda.synchronize()
    got = torch.as_tensor(raw, device=DEV).clone().cpu().numpy()
    for b in range(F_):
        want = pc.prefix_starts(order_loc[b].astype(np.int64), infos[b]["mask_undilated"][0], infos[b]["mask_dilated"][0], 32, 32, first)
        assert np.array_equal(got[:, b], want), b
    assert (got < first).any() and (got > 0).any()   # something is evaluated, something is skipped
    eng.set_tuning(prefix_full=1)
    c_full, l_full = run()
    eng.set_tuning(prefix_full=0)
    assert torch.equal(c_cone, c_full)
    walked = np.zeros((F_, 1024), bool)
    for b in range(F_):
        walked[b, order_loc[b][first:]] = True
    assert torch.equal(l_cone[torch.from_numpy(walked).to(DEV)], l_full[torch.from_numpy(walked).to(DEV)])


@pytest.mark.parametrize("H,W", [(16, 16), (8, 12)])
def test_prefix_cone_on_other_grids_and_random_orders(H, W):
    """k_prefix_starts on a square and a non-square grid with RANDOM generation orders (where the cone of a stage is far
    from a suffix of the prefix, so the start ranks give away a lot -- but never too little): device table == numpy
    restatement, codes and walked logits identical to the full prefix."""
    import ctypes
    from oracle import prefix_cone_oracle as pc
    from pixelsynth_amd import _lib
    net = make_net(6)
    F_, L = 3, H * W
    eng = net.engine(H, W, F_)
    rs = np.random.RandomState(H * 10 + W)
    orders = [np.stack(np.unravel_index(rs.permutation(L), (H, W)), 1).astype(np.int32) for _ in range(F_)]
    masks = [np.concatenate([c_oracle.unfolded_masks(o, H, W, 3, dil, typ) for o in orders]) for dil, typ in ((1, "A"), (1, "B"), (2, "B"))]
    ms = [tt(m) for m in masks]
    order_loc = np.stack([o[:, 0] * W + o[:, 1] for o in orders]).astype(np.int32)
    first = (2 * L) // 3
    reg = np.zeros((F_, L), np.uint8)
    for b in range(F_):
        reg[b, order_loc[b][first:]] = 1
    codes0 = rs.randint(0, 512, size=(F_, L)).astype(np.int32)
    u = tt(rs.rand(F_, L).astype(np.float32))

    def run():
        c = tt(codes0.copy())
        lg = eng.ar_run(c, tt(order_loc), tt(reg), *ms, temperature=0.9, uniforms=u, first_step=first, want_logits=True)
        eng.check()
        return c, lg
    eng.set_tuning(prefix_cone_force=1)
    c_cone, l_cone = run()
    eng.set_tuning(prefix_cone_force=0)
    lib = _lib.lib()
    lib.ps_pixelcnn_debug_cache.restype = ctypes.c_void_p
    ptr = lib.ps_pixelcnn_debug_cache(eng.handle, 5, 0)
    raw = type("Raw", (), {"__cuda_array_interface__": {"shape": (pc.N_EVAL, F_), "typestr": "<i4", "data": (ptr, False), "version": 2}})()
    torch.cuda.synchronize()
    got = torch.as_tensor(raw, device=DEV).clone().cpu().numpy()
    for b in range(F_):
        want = pc.prefix_starts(order_loc[b].astype(np.int64), masks[1][b], masks[2][b], H, W, first)
        assert np.array_equal(got[:, b], want), b
    eng.set_tuning(prefix_full=1)
    c_full, l_full = run()
    eng.set_tuning(prefix_full=0)
    assert torch.equal(c_cone, c_full)
    walked = np.zeros((F_, L), bool)
    for b in range(F_):
        walked[b, order_loc[b][first:]] = True
    sel = torch.from_numpy(walked).to(DEV)
    assert torch.equal(l_cone[sel], l_full[sel])


def test_ar_prefix_and_columns_as_separate_calls_equal_the_run():
    """ps_pixelcnn_ar_prefix over two disjoint frame ranges -- on two streams, each confined to its own compute units
    (ps_stream_create_cu_range) -- followed by ps_pixelcnn_ar_columns on a handle told how many compute units its stream has,
    gives the codes of ps_pixelcnn_ar_run_waves bit for bit."""
    from pixelsynth_amd.lmconv.model import wavefronts
    from pixelsynth_amd.lmconv.model import CuRangeStream
    net = make_net(3)
    F_, first = 70, 800
    bgs = syn.background_masks(256)
    names = ["right_half", "half_plus_island", "ragged", "top_band"]
    infos = [c_oracle.masks_for_background(bgs[names[b % 4]], 32) for b in range(F_)]
    order_loc = np.stack([(i["order"][:, 0] * 32 + i["order"][:, 1]) for i in infos]).astype(np.int32)
    reg = np.zeros((F_, 1024), np.uint8)
    rs = np.random.RandomState(2)
    for b in range(F_):
        walked = order_loc[b][first:]
        reg[b, walked[rs.rand(walked.size) < 0.8]] = 1
        reg[b, order_loc[b][first]] = 1
    ms = [tt(np.concatenate([i[k] for i in infos])) for k in ("mask_init", "mask_undilated", "mask_dilated")]
    codes0 = syn.codes(17, F_).reshape(F_, 1024).astype(np.int32)
    u = tt(np.random.RandomState(9).rand(F_, 1024).astype(np.float32))
    o, r = tt(order_loc), tt(reg)
    waves = wavefronts(order_loc, 32, 32, first, DEV)
    eng = net.engine(32, 32, F_)
    c_ref = tt(codes0.copy())
    eng.ar_run(c_ref, o, r, *ms, temperature=0.7, uniforms=u, first_step=first, waves=waves)
    eng.check()
    A, B = CuRangeStream(0, 160), CuRangeStream(160, 96)
    eng2 = net.engine(32, 32, F_, slot=1)
    eng2.set_compute_units(160)
    c = tt(codes0.copy())
    torch.cuda.synchronize()
    with torch.cuda.stream(B.stream):
        eng2.ar_prefix(c, o, r, *ms, first, 0, 41)
        done = torch.cuda.Event()
        done.record(B.stream)
    with torch.cuda.stream(A.stream):
        eng2.ar_prefix(c, o, r, *ms, first, 41, F_)
        A.stream.wait_event(done)
        eng2.ar_columns(c, o, r, *ms, waves, temperature=0.7, uniforms=u, first_step=first)
    A.stream.synchronize()
    eng2.check()
    eng2.set_compute_units(0)
    assert torch.equal(c, c_ref)
    # a frame range that is not a range of the run is an error, not "all frames" (round-3 advice)
    for lo, hi in ((0, -1), (5, 3), (0, F_ + 1), (-1, 4)):
        with pytest.raises(RuntimeError, match="not a range"):
            eng2.ar_prefix(c, o, r, *ms, first, lo, hi)
    assert (c.cpu().numpy()[reg == 1] != codes0[reg == 1]).any()
    # tuning values by name (include/pixelsynth_hip_debug.h): unknown names and values outside a value's range are errors; the
    # look-ahead depths are fixed by the handle's first column launch; the results-invalid switches do not exist in the product build
    with pytest.raises(RuntimeError, match="no tuning value named"):
        eng2.set_tuning(no_such_value=1)
    with pytest.raises(RuntimeError, match="no tuning value named"):
        eng2.set_tuning(column_debug=1)
    with pytest.raises(RuntimeError, match="outside"):
        eng2.set_tuning(wg_ti_out=3)
    with pytest.raises(RuntimeError, match="before the handle's first column launch"):
        eng2.set_tuning(tp_ahead=(eng2.get_tuning("tp_ahead") + 1) % 8)
    eng2.set_tuning(tp_ahead=eng2.get_tuning("tp_ahead"))   # (the same value: nothing to change, no error)
    assert eng2.get_tuning("tp_min_cols") == 257
