"""Hard z-buffer reprojection on MI355X -- drop-in for the reference's models/projection/depth_manipulator.py:DepthManipulator
(same constructor and method signature).  The reference uses it in its legacy `depth_model` baseline to find
the visible / invisible regions of a target view: every source pixel is unprojected with its depth, moved to the target camera,
and the source coordinate of the point that wins the z-test is written at the pixel it lands on -- a backward sampler for
grid_sample.

The projection is a handful of (B,4,4) products, done with torch on the device in the reference's association order; the
z-test scatter -- the part with a data-dependent write order -- is the HIP kernel ps_zbuffer_scatter_f32.  What "wins" is
defined by what the reference computes on the CPU (index_put_ processes the sorted points in order, the last write stays):
points sorted by projected z, descending, stably; of several points on one pixel the one sorted LAST stays.  (That is the
FARTHEST point -- z is negative in front of the camera -- whatever the reference's comment intends; its CUDA path leaves the
winner undefined.)  Two more quirks are kept as they are: the image size in the pixel mapping is the literal 128 / 255 of the
reference, and the out-of-range flag is added by ORIGINAL point position to values gathered in SORTED order (:86-97)."""
import numpy as np
import torch
import torch.nn as nn

from .. import _lib

EPS = 1e-2


class DepthManipulator(nn.Module):
    def __init__(self, W=256):
        super().__init__()
        xs, ys = np.meshgrid(np.linspace(-1, 1, W), np.linspace(1, -1, W))      # depth_manipulator.py:20-26
        xys = np.vstack((xs.reshape(1, W, W), ys.reshape(1, W, W), -np.ones((1, W, W)), np.ones((1, W, W))))
        self.grid = torch.Tensor(xys).unsqueeze(0)

    def homogenize(self, xys):
        assert xys.size(1) <= 3
        ones = torch.ones(xys.size(0), 1, xys.size(2)).to(xys.device)
        return torch.cat((xys, ones), 1)

    @torch.no_grad()
    def project_zbuffer(self, depth, K, K_inv, RTinv_cam1, RT_cam2):
        """depth (B,1,w,h), cameras (B,4,4) -> (bilinear_sampler (B,2,w,h), projected depth (B,1,w,h))."""
        _lib.require_cuda(depth, K, K_inv, RTinv_cam1, RT_cam2)
        bs, _, w, h = depth.size()
        orig_xys = self.grid.to(depth.device).repeat(bs, 1, 1, 1)
        xys = orig_xys * depth
        xys[:, -1, :] = 1
        xys = xys.view(bs, 4, -1)
        cam1_X = K_inv.bmm(xys)
        RT = RT_cam2.bmm(RTinv_cam1)
        wrld_X = RT.bmm(cam1_X)
        xy_proj = K.bmm(wrld_X)
        mask = xy_proj[:, 2:3, :].abs() < EPS
        sampler = xy_proj[:, 0:2, :] / -xy_proj[:, 2:3, :]
        sampler[mask.repeat(1, 2, 1)] = -10
        sampler[:, 1, :] = -sampler[:, 1, :]
        tsampler = ((sampler + 1) * 128).view(bs, 2, -1)
        _, sampler_inds = xy_proj[:, 2:3, :].sort(dim=2, descending=True, stable=True)
        order = sampler_inds[:, 0]                                                   # (B,N) point at each sorted position
        xs = torch.gather(tsampler[:, 0], 1, order).long().clamp(min=0, max=255)
        ys = torch.gather(tsampler[:, 1], 1, order).long().clamp(min=0, max=255)
        flag = ((tsampler < 0) | (tsampler > 255)).float().max(dim=1)[0] * 4         # (B,N), by ORIGINAL position (:86-87)
        oxy = orig_xys[:, :2].reshape(bs, 2, -1)
        v0 = (torch.gather(oxy[:, 0], 1, order) + flag).contiguous()
        v1 = (-torch.gather(oxy[:, 1], 1, order) + flag).contiguous()
        out = torch.full((bs, 2, w, h), -2.0, device=depth.device, dtype=torch.float32)
        winner = torch.empty(bs, w, h, dtype=torch.int32, device=depth.device)
        ys32, xs32 = ys.to(torch.int32).contiguous(), xs.to(torch.int32).contiguous()
        rc = _lib.lib().ps_zbuffer_scatter_f32(_lib.ptr(ys32), _lib.ptr(xs32), _lib.ptr(v0), _lib.ptr(v1), bs, w * h, w, h,
                                               _lib.ptr(out), _lib.ptr(winner), _lib.current_stream())
        _lib.check(rc, "ps_zbuffer_scatter_f32")
        return out, -xy_proj[:, 2:3, :].view(bs, 1, w, h)
