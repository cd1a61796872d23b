"""GPU tests of the hard z-buffer: the z-test scatter kernel bit-exact against the sequential definition, the
DepthManipulator mirror against the reference's own outputs (tests/golden/zbuffer.npz)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from oracle import zbuffer_oracle as zo  # noqa: E402
from pixelsynth_amd import _lib  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def tt(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def test_zbuffer_scatter_kernel_is_the_sequential_last_write():
    """ps_zbuffer_scatter_f32 against numpy's in-order assignment (last write to a pixel stays) on heavy pile-ups."""
    rs = np.random.RandomState(0)
    B, N, H, W = 3, 50000, 64, 48
    ys = rs.randint(0, H, size=(B, N)).astype(np.int32)
    xs = (rs.randint(0, W, size=(B, N)) // 3 * 3 % W).astype(np.int32)      # many points per pixel, some pixels never hit
    v0, v1 = rs.randn(B, N).astype(np.float32), rs.randn(B, N).astype(np.float32)
    want = np.full((B, 2, H, W), -2.0, np.float32)
    for b in range(B):
        want[b, 0, ys[b], xs[b]] = v0[b]
        want[b, 1, ys[b], xs[b]] = v1[b]
    out = torch.full((B, 2, H, W), -2.0, device=DEV)
    winner = torch.empty(B, H, W, dtype=torch.int32, device=DEV)
    dev = [tt(a) for a in (ys, xs, v0, v1)]     # (kept alive: a temporary's block would be handed to the next upload)
    rc = _lib.lib().ps_zbuffer_scatter_f32(*[_lib.ptr(a) for a in dev], B, N, H, W, _lib.ptr(out), _lib.ptr(winner),
                                           _lib.ptr(_lib.status_word(DEV)), _lib.current_stream())
    _lib.check(rc, "ps_zbuffer_scatter_f32")
    _lib.read_status("ps_zbuffer_scatter_f32", DEV)      # synchronises; nothing was dropped
    assert np.array_equal(out.cpu().numpy(), want)
    assert (want == -2).any() and (want != -2).any()


def test_status_word_reports_dropped_points_and_bad_orders():
    """Data errors only the device can see -- a pixel outside the image in the z-buffer scatter, a generation order that is no
    permutation -- raise a bit in the CALLER's status word; ps_read_status turns it into an error (message through
    ps_last_error) and clears it.  The library holds no flag of its own."""
    B, N, H, W = 1, 64, 8, 8
    ys = np.zeros((B, N), np.int32)
    xs = np.zeros((B, N), np.int32)
    ys[0, 5] = H                                           # outside
    v = np.zeros((B, N), np.float32)
    out = torch.full((B, 2, H, W), -2.0, device=DEV)
    winner = torch.empty(B, H, W, dtype=torch.int32, device=DEV)
    dev = [tt(a) for a in (ys, xs, v, v)]
    st = _lib.status_word(DEV)
    _lib.check(_lib.lib().ps_zbuffer_scatter_f32(*[_lib.ptr(a) for a in dev], B, N, H, W, _lib.ptr(out), _lib.ptr(winner), _lib.ptr(st),
                                                 _lib.current_stream()), "ps_zbuffer_scatter_f32")
    with pytest.raises(RuntimeError, match="outside the image"):
        _lib.read_status("scatter", DEV)
    _lib.read_status("scatter", DEV)                       # cleared by the read
    order = torch.arange(16, dtype=torch.int32, device=DEV).view(1, 16).clone()
    order[0, 3] = 99                                       # no permutation of the 4x4 grid
    masks = [torch.empty(1, 9, 16, device=DEV) for _ in range(3)]
    _lib.check(_lib.lib().ps_order_masks_f32(_lib.ptr(order), 1, 4, 4, *[_lib.ptr(m) for m in masks], _lib.ptr(st),
                                             _lib.current_stream()), "ps_order_masks_f32")
    with pytest.raises(RuntimeError, match="no permutation"):
        _lib.read_status("order_masks", DEV)
    _lib.read_status("order_masks", DEV)


def test_depth_manipulator_refuses_sizes_the_reference_literals_do_not_cover():
    from pixelsynth_amd.projection.depth_manipulator import DepthManipulator
    dm = DepthManipulator(128)
    eye = torch.eye(4, device=DEV)[None]
    with pytest.raises(ValueError, match="256x256"):
        dm.project_zbuffer(torch.ones(1, 1, 128, 128, device=DEV), eye, eye, eye, eye)


def test_depth_manipulator_matches_the_reference_outputs():
    import make_golden_inputs as mgi
    from pixelsynth_amd.projection.depth_manipulator import DepthManipulator
    fx = np.load(os.path.join(GOLD, "zbuffer.npz"))
    for name, W, d, cams, RT2 in mgi.zbuffer_cases({str(n): int(fx[f"{n}_seed"]) for n in fx["names"]}):
        dm = DepthManipulator(W)
        smp, dep = dm.project_zbuffer(tt(d), tt(cams["K"]), tt(cams["Kinv"]), tt(cams["Pinv"]), tt(RT2))
        smp, dep = smp.cpu().numpy(), dep.cpu().numpy()
        np.testing.assert_allclose(dep[:, :, ::5, ::5], fx[f"{name}_depth_sub"], rtol=2e-5, atol=2e-5)
        # which point stays on a pixel is an integer decision taken on float products: the device's bmm may round a product
        # the other way than the CPU's, which moves a point on a pixel boundary next door or flips a near-tie in z -- a few
        # entries in a thousand; everything else is exactly the reference's value (source-grid coordinates, not arithmetic)
        got, ref = smp[:, :, ::3, ::3], fx[f"{name}_sampler_sub"]
        differ = (got != ref).mean()
        assert differ < 5e-3, f"{name}: {differ:.4f} of the sampler entries differ from the reference"
        filled = np.array([(smp[b, 0] != -2).sum() for b in range(smp.shape[0])])
        assert np.all(np.abs(filled - fx[f"{name}_filled"]) <= 0.005 * fx[f"{name}_filled"] + 8)
        # and with the same projected points the mirror's scatter IS the oracle's, bit for bit
        want, _ = zo.project_zbuffer(d, cams["K"], cams["Kinv"], cams["Pinv"], RT2)
        assert (smp != want).mean() < 5e-3
