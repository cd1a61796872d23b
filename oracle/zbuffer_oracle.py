"""TEST INFRASTRUCTURE (not shipped, never imported by pixelsynth_amd/): numpy restatement of the reference's hard z-buffer,
DepthManipulator.project_zbuffer (models/projection/depth_manipulator.py:37-104), pinned against outputs of the reference itself
(tests/golden/zbuffer.npz, generated single-threaded on the CPU, where torch's index_put_ assigns in order and the last write
to a pixel stays -- numpy's fancy assignment has the same documented rule)."""
import numpy as np

EPS = 1e-2


def grid(W):
    """:20-26 -- rows (x, y, -1, 1), x left to right in [-1, 1], y top to bottom from +1 to -1."""
    xs, ys = np.meshgrid(np.linspace(-1, 1, W), np.linspace(1, -1, W))
    return np.vstack((xs.reshape(1, W, W), ys.reshape(1, W, W), -np.ones((1, W, W)), np.ones((1, W, W)))).astype(np.float32)[None]


def project_zbuffer(depth, K, K_inv, RTinv_cam1, RT_cam2, ties=None):
    """depth (B,1,w,h) f32, cameras (B,4,4) f32 -> (bilinear_sampler (B,2,w,h), projected depth (B,1,w,h)), float32 arithmetic
    in the reference's association order (:52-61).  ties: optional list that receives, per frame, the number of z ties whose
    order would change the result (equal z at neighbouring sorted positions with different out-of-range flags at those
    positions, or on the same pixel) -- the reference's sort is not stable, so a fixture must have none."""
    bs, _, w, h = depth.shape
    f = np.float32
    orig = np.repeat(grid(w), bs, 0)
    xys = (orig * depth).astype(f)
    xys[:, -1] = 1
    xys = xys.reshape(bs, 4, -1)
    mm = lambda a, b: np.stack([(a[i].astype(f) @ b[i].astype(f)).astype(f) for i in range(bs)])
    cam1 = mm(K_inv, xys)
    RT = mm(RT_cam2, RTinv_cam1)
    wrld = mm(RT, cam1)
    proj = mm(K, wrld)
    z = proj[:, 2:3]
    mask = np.abs(z) < EPS
    with np.errstate(divide="ignore", invalid="ignore"):
        sampler = (proj[:, 0:2] / -z).astype(f)
    sampler[np.repeat(mask, 2, 1)] = -10
    sampler[:, 1] = -sampler[:, 1]
    ts = ((sampler + 1) * 128).astype(f)
    out = np.full((bs, 2, w, h), -2.0, f)
    oxy = orig[:, :2].reshape(bs, 2, -1)
    for b in range(bs):
        order = np.argsort(-z[b, 0], kind="stable")                       # z descending, ties in original order
        xs = np.clip(np.trunc(ts[b, 0, order]).astype(np.int64), 0, 255)   # .long() truncates toward zero
        ys = np.clip(np.trunc(ts[b, 1, order]).astype(np.int64), 0, 255)
        flag = (((ts[b] < 0) | (ts[b] > 255)).astype(f).max(0) * 4).astype(f)   # by ORIGINAL position, added to sorted values
        if ties is not None:
            zs = z[b, 0, order]
            eq = np.nonzero(zs[1:] == zs[:-1])[0]
            ties.append(int(((flag[eq] != flag[eq + 1]) | ((xs[eq] == xs[eq + 1]) & (ys[eq] == ys[eq + 1]))).sum()))
        out[b, 0, ys, xs] = oxy[b, 0, order] + flag                         # duplicates: the last assignment stays
        out[b, 1, ys, xs] = -oxy[b, 1, order] + flag
    return out, (-z).reshape(bs, 1, w, h)
