"""CPU-only self-consistency of the rasterizer/compositor oracle (PyTorch3D semantics, parity
unpinned -- see oracle/pixelsynth_oracle.c): accelerated == literal restatement, and the
size-independent properties the GPU tests rely on."""
import numpy as np

from oracle import c_oracle


def cloud(seed, B, N, spread=1.2, zlo=-0.2):
    rs = np.random.RandomState(seed)
    pts = np.empty((B, N, 3), np.float32)
    pts[..., :2] = (rs.rand(B, N, 2) * 2 - 1) * spread
    pts[..., 2] = rs.rand(B, N) * 5 + zlo
    return pts


def test_rowbin_equals_naive():
    pts = cloud(1, 2, 700)
    pts[0, :40, 2] = 1.5  # z ties -> index tie-break
    a = c_oracle.rasterize(pts, 32, 4.0 / 32 * 2, 8, naive=True)
    b = c_oracle.rasterize(pts, 32, 4.0 / 32 * 2, 8, naive=False)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    idx, zbuf, dist = a
    hit = idx >= 0
    assert hit.any() and (~hit).any()
    assert (zbuf[hit] >= 0).all() and (dist[hit] < (4.0 / 32 * 2) ** 2).all()
    z = np.where(hit, zbuf, np.float32(1e30))
    assert (np.diff(z, axis=-1) >= 0).all()  # ascending z, pads last


def test_splat_properties():
    pts = cloud(2, 1, 900)
    feat = np.random.RandomState(3).rand(1, 3, 900).astype(np.float32)
    r = c_oracle.splat_forward(pts, feat, 32, radius_px=4.0, K=16)
    assert np.array_equal(r["pts_after"][..., :2], -pts[..., :2])  # in-place negation of x,y only
    assert np.array_equal(r["pts_after"][..., 2], pts[..., 2])
    # linearity in the features (alphas depend on geometry only)
    r2 = c_oracle.splat_forward(pts, feat * 2, 32, radius_px=4.0, K=16)
    np.testing.assert_allclose(r2["feat"], 2 * r["feat"], rtol=1e-6, atol=1e-7)
    # constant features + wsumnorm -> the constant wherever something was hit
    ones = np.ones_like(feat)
    r3 = c_oracle.splat_forward(pts, ones, 32, radius_px=4.0, K=16, accumulation="wsumnorm", bg_ksize=1)
    hit = r3["idx"][..., 0] >= 0
    np.testing.assert_allclose(r3["feat"][0, 0][hit[0]], 1.0, rtol=1e-5)
    assert np.array_equal(r3["bg"], ~hit)  # ksize 1 = no dilation
    # dilation only ever grows the mask
    r13 = c_oracle.splat_forward(pts, feat, 32, radius_px=4.0, K=16, bg_ksize=13)
    assert (r13["bg"] | ~hit == r13["bg"]).all() and r13["bg"].sum() >= (~hit).sum()


def test_all_behind_camera_is_all_background():
    pts = cloud(4, 1, 50)
    pts[..., 2] = -1.0
    r = c_oracle.splat_forward(pts, np.ones((1, 3, 50), np.float32), 16, radius_px=4.0, K=4)
    assert r["bg"].all() and (r["idx"] == -1).all() and (r["feat"] == 0).all()
