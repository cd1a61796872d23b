"""Soft z-buffer splatter on MI355X -- drop-in for the reference's
models/layers/z_buffer_layers.py:RasterizePointsXYsBlending (same constructor, same forward).

The reference rasterizes with PyTorch3D (rasterize_points + compositing.*, z_buffer_layers.py:81-84,
112-129) and materialises (B,S,S,K) idx/dist/alpha tensors; here one C-ABI call
(ps_splat_f32 -> pixelsynth_amd/csrc/splat.hip) bins, sorts and composites on the fly.
There is no CPU fallback: tensors must live on the ROCm device.
"""
import os

import torch
from torch import nn

from .. import _lib

ACCUMULATION = {"alphacomposite": 0, "wsum": 1, "wsumnorm": 2}


class _Workspace:
    """Scratch owned by PyTorch (the library never allocates); grown on demand, kept per device."""

    def __init__(self):
        self.buf = {}

    def get(self, device, nbytes):
        t = self.buf.get(device)
        if t is None or t.numel() < nbytes:
            t = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
            self.buf[device] = t
        return t


_WS = _Workspace()


def splat_workspace(device, B, N, S, radius_px):
    n = _lib.lib().ps_splat_workspace_bytes(B, N, S, float(radius_px))
    if n == 0:
        raise RuntimeError("ps_splat_workspace_bytes: invalid sizes")
    return _WS.get(device, n)


class RasterizePointsXYsBlending(nn.Module):
    """Same inputs/outputs as the reference class (z_buffer_layers.py:12-131).

    forward(pts3D (B,N,3), src (B,C,N)) -> (features (B,C,S,S) f32, background_mask (B,S,S) bool).
    Like the reference, x and y of the caller's pts3D are negated in place (:71-72)."""

    def __init__(self, C=64, learn_feature=True, radius=1.5, size=256, points_per_pixel=8, opts=None):
        super().__init__()
        # `default_feature` is never used by forward (in the reference neither) but checkpoints carry it:
        # a (1,C,1) parameter when learnt, a zero buffer otherwise
        if learn_feature:
            self.default_feature = nn.Parameter(torch.randn(1, C, 1))
        else:
            self.register_buffer("default_feature", torch.zeros(1, C, 1))
        self.radius, self.size, self.points_per_pixel, self.opts = radius, size, points_per_pixel, opts

    def _opt(self, name, default):
        return getattr(self.opts, name, default) if self.opts is not None else default

    def forward(self, pts3D, src, return_debug=False):
        if src.dim() > 3:    # image-shaped input: (B,C,w,w) features with a (B,3,N) cloud, one row of points per w
            bs, c, w = src.shape[:3]
            image_size = w
            pts3D = pts3D.permute(0, 2, 1)
            src = src.unsqueeze(2).expand(bs, c, w, *src.shape[2:]).reshape(bs, c, -1)
        else:
            bs, image_size = src.size(0), self.size
        # cloud and features must be arranged alike: (B,N,3) against (B,C,N)
        if pts3D.size(2) != 3 or pts3D.size(1) != src.size(2):
            raise AssertionError(f"splat: points {tuple(pts3D.shape)} do not match features {tuple(src.shape)}")
        _lib.require_cuda(pts3D, src)
        os.environ.get("DEBUG")  # the reference reads os.environ["DEBUG"] (KeyError if unset); tolerated here

        B, N, C = bs, pts3D.size(1), src.size(1)
        caller_pts = pts3D
        pts = pts3D if (pts3D.is_contiguous() and pts3D.dtype == torch.float32) else pts3D.float().contiguous()
        feat = src.float().contiguous()
        S = int(image_size)
        K = int(self.points_per_pixel)
        out = torch.empty(B, C, S, S, dtype=torch.float32, device=pts.device)
        bg = torch.empty(B, S, S, dtype=torch.uint8, device=pts.device)
        idx = zbuf = dist = None
        if return_debug:
            idx = torch.empty(B, S, S, K, dtype=torch.int32, device=pts.device)
            zbuf = torch.empty(B, S, S, K, dtype=torch.float32, device=pts.device)
            dist = torch.empty(B, S, S, K, dtype=torch.float32, device=pts.device)
        ws = splat_workspace(pts.device, B, N, S, self.radius)
        rc = _lib.lib().ps_splat_f32(
            _lib.ptr(pts), _lib.ptr(feat), B, N, C, S, float(self.radius), K,
            float(self._opt("tau", 1.0)), int(self._opt("rad_pow", 2)),
            ACCUMULATION[self._opt("accumulation", "alphacomposite")],
            int(self._opt("background_smoothing_kernel_size", 13)),
            _lib.ptr(out), _lib.ptr(bg), _lib.ptr(idx), _lib.ptr(zbuf), _lib.ptr(dist),
            _lib.ptr(ws), ws.numel(), _lib.current_stream())
        _lib.check(rc, "ps_splat_f32")
        if pts is not caller_pts:  # keep the reference's visible side effect on the caller's tensor
            caller_pts[:, :, 0:2] = pts[:, :, 0:2].to(caller_pts.dtype)
        background_mask = bg.view(torch.bool)    # (k_dilate writes 0 / 1: the same bytes are the boolean mask)
        if return_debug:
            return out, background_mask, idx, zbuf, dist
        return out, background_mask
