"""ctypes/numpy wrapper over oracle/libpixelsynth_oracle.so (TEST INFRASTRUCTURE).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Each wrapper names the reference function it checks (see pixelsynth_oracle.c for file:line).
"""
import ctypes
import os

import numpy as np

from . import build_oracle as _b

_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libpixelsynth_oracle.so")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(
                os.path.join(os.path.dirname(path), "pixelsynth_oracle.c")):
            _b.build_oracle()
        _LIB = ctypes.CDLL(path)
    return _LIB


def _p(a, t=ctypes.c_void_p):
    return None if a is None else a.ctypes.data_as(t)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def make_grid(W):
    """PtsManipulator.xyzs buffer -> (1,4,W*W) f32."""
    out = np.empty((4, W * W), np.float32)
    lib().ps_oracle_make_grid(ctypes.c_int(W), _p(out))
    return out[None]


def project_pts(depth, K, Kinv, RT1inv, RT2, W):
    """PtsManipulator.project_pts: depth (B,1,N) -> sampler (B,3,N)."""
    depth = _f32(depth).reshape(depth.shape[0], -1)
    B = depth.shape[0]
    out = np.empty((B, 3, W * W), np.float32)
    lib().ps_oracle_project_pts(_p(depth), _p(_f32(K)), _p(_f32(Kinv)), _p(_f32(RT1inv)),
                                _p(_f32(RT2)), ctypes.c_int(B), ctypes.c_int(W), _p(out))
    return out


def project_pts_cumulative(depth_new, last_bg, prior, K, Kinv, RT1inv, RT2, RT3inv, W):
    """PtsManipulator.project_pts_cumulative -> (sampler (B,3,NT), cloud (B,4,NT))."""
    B = depth_new.shape[0]
    depth_new = _f32(depth_new).reshape(B, -1)
    n_new = depth_new.shape[1]
    n_prior = 0 if prior is None else prior.shape[2]
    NT = n_new + n_prior
    sampler = np.empty((B, 3, NT), np.float32)
    cloud = np.empty((B, 4, NT), np.float32)
    lb = None if last_bg is None else np.ascontiguousarray(last_bg.reshape(B, -1), dtype=np.uint8)
    pr = None if prior is None else _f32(prior)
    r3 = None if RT3inv is None else _f32(RT3inv)
    lib().ps_oracle_project_pts_cumulative(
        _p(depth_new), _p(lb), _p(pr), _p(_f32(K)), _p(_f32(Kinv)), _p(_f32(RT1inv)), _p(_f32(RT2)),
        _p(r3), ctypes.c_int(B), ctypes.c_int(W), ctypes.c_int(n_new), ctypes.c_int(n_prior),
        _p(sampler), _p(cloud))
    return sampler, cloud


def rasterize(pts, S, radius, K, naive=False):
    """PyTorch3D rasterize_points semantics on pts (B,N,3) -> idx,zbuf,dist (B,S,S,K)."""
    pts = _f32(pts)
    B, N, _ = pts.shape
    idx = np.empty((B, S, S, K), np.int32)
    zbuf = np.empty((B, S, S, K), np.float32)
    dist = np.empty((B, S, S, K), np.float32)
    fn = lib().ps_oracle_rasterize_naive if naive else lib().ps_oracle_rasterize
    fn(_p(pts), ctypes.c_int(B), ctypes.c_int(N), ctypes.c_int(S), ctypes.c_float(radius),
       ctypes.c_int(K), _p(idx), _p(zbuf), _p(dist))
    return idx, zbuf, dist


ACCUMULATION = {"alphacomposite": 0, "wsum": 1, "wsumnorm": 2}


def splat_forward(pts, feat, S, radius_px=4.0, K=128, tau=1.0, rad_pow=2,
                  accumulation="alphacomposite", bg_ksize=13, naive=False):
    """RasterizePointsXYsBlending.forward(pts3D (B,N,3), src (B,C,N)).

    Returns dict(feat (B,C,S,S), bg (B,S,S) bool, idx, zbuf, dist (B,S,S,K), pts_after (negated))."""
    pts = _f32(pts).copy()
    feat = _f32(feat)
    B, N, _ = pts.shape
    C = feat.shape[1]
    out = np.empty((B, C, S, S), np.float32)
    bg = np.empty((B, S, S), np.uint8)
    idx = np.empty((B, S, S, K), np.int32)
    zbuf = np.empty((B, S, S, K), np.float32)
    dist = np.empty((B, S, S, K), np.float32)
    lib().ps_oracle_splat_forward(
        _p(pts), _p(feat), ctypes.c_int(B), ctypes.c_int(N), ctypes.c_int(C), ctypes.c_int(S),
        ctypes.c_double(radius_px), ctypes.c_int(K), ctypes.c_float(tau), ctypes.c_int(rad_pow),
        ctypes.c_int(ACCUMULATION[accumulation]), ctypes.c_int(bg_ksize), ctypes.c_int(int(naive)),
        _p(out), _p(bg), _p(idx), _p(zbuf), _p(dist))
    return dict(feat=out, bg=bg.astype(bool), idx=idx, zbuf=zbuf, dist=dist, pts_after=pts)


def block_all(mask, blk=8, invert=False):
    """AvgPool2d(blk)(mask.float()).astype(uint8): 1 iff the whole block is set (or clear if invert)."""
    mask = np.ascontiguousarray(mask, dtype=np.uint8)
    S = mask.shape[-1]
    out = np.empty((S // blk, S // blk), np.uint8)
    lib().ps_oracle_block_all(_p(mask), ctypes.c_int(S), ctypes.c_int(blk), ctypes.c_int(int(invert)),
                              _p(out))
    return out


def chamfer_dt5(src):
    """cv2.distanceTransform(src, DIST_L2, 5) portable fixed-point path (parity unpinned)."""
    src = np.ascontiguousarray(src, dtype=np.uint8)
    H, W = src.shape
    out = np.empty((H, W), np.float32)
    lib().ps_oracle_chamfer_dt5(_p(src), ctypes.c_int(H), ctypes.c_int(W), _p(out))
    return out


def signed_distance(fgb, bgb):
    """(DT(fg) - DT(bg)).astype(int) of models/z_buffermodel.py:673-675 -> int64 (H,W)."""
    fgb = np.ascontiguousarray(fgb, dtype=np.uint8)
    bgb = np.ascontiguousarray(bgb, dtype=np.uint8)
    H, W = fgb.shape
    out = np.empty((H, W), np.int64)
    lib().ps_oracle_signed_distance(_p(fgb), _p(bgb), ctypes.c_int(H), ctypes.c_int(W), _p(out))
    return out


def custom_idx(rows, cols, distances):
    """get_custom_order.custom_idx: returns (order (L,2) int32, distances*10000) -- input not mutated."""
    d = np.ascontiguousarray(distances, dtype=np.int64).copy()
    order = np.empty((rows * cols, 2), np.int32)
    lib().ps_oracle_custom_idx(ctypes.c_int(rows), ctypes.c_int(cols), _p(d), _p(order))
    return order, d


def unfolded_masks(order, nrows, ncols, k=3, dilation=1, mask_type="B"):
    """masking.get_unfolded_masks(observed_idx=None) -> (1, k*k, nrows*ncols) f32."""
    order = np.ascontiguousarray(order, dtype=np.int32)
    out = np.empty((k * k, nrows * ncols), np.float32)
    lib().ps_oracle_kernel_masks(_p(order), ctypes.c_int(order.shape[0]), ctypes.c_int(nrows),
                                 ctypes.c_int(ncols), ctypes.c_int(k), ctypes.c_int(dilation),
                                 ctypes.c_int(int(mask_type == "B")), _p(out))
    return out[None]


def masks_for_background(bg, obs=32):
    """Host-side glue of ZbufferModelPts.get_masks_for_batch for ONE image: bg (S,S) bool ->
    dict(order (L,2), mask_init/mask_undilated/mask_dilated (1,9,L), bg32 (obs,obs) uint8, D int64)."""
    bg = np.ascontiguousarray(bg, dtype=np.uint8)
    blk = bg.shape[-1] // obs
    fgb = block_all(bg, blk, invert=True)
    bgb = block_all(bg, blk, invert=False)
    D = signed_distance(fgb, bgb)
    order, _ = custom_idx(obs, obs, D)
    return dict(order=order, D=D, bg32=bgb, fg32=fgb,
                mask_init=unfolded_masks(order, obs, obs, 3, 1, "A"),
                mask_undilated=unfolded_masks(order, obs, obs, 3, 1, "B"),
                mask_dilated=unfolded_masks(order, obs, obs, 3, 2, "B"))
