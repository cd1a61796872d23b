"""Hard z-buffer reprojection on MI355X -- drop-in for the reference's models/projection/depth_manipulator.py:DepthManipulator
(same constructor and method signature).  The reference uses it in its legacy `depth_model` baseline to find
the visible / invisible regions of a target view: every source pixel is unprojected with its depth, moved to the target camera,
and the source coordinate of the point that wins the z-test is written at the pixel it lands on -- a backward sampler for
grid_sample.

Both halves are HIP kernels behind the C ABI: ps_zbuffer_project_f32 (unprojection, the (B,4,4) camera products in the
reference's association order, EPS rule, pixel mapping and out-of-range flag: one thread per source pixel instead of a dozen
small torch launches) and ps_zbuffer_scatter_sorted_f32 (the z-test, the part with a data-dependent write order); the sort by
projected z in between is torch's (stable, descending -- what the reference calls).  What "wins" is
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
        if w != 256 or h != 256:   # the literals 128 / 255 of :66-90 only mean "the image" at 256 (the reference indexes out of bounds below it)
            raise ValueError(f"DepthManipulator.project_zbuffer: {w}x{h} input; the reference's pixel mapping is written for 256x256")
        dev, N = depth.device, w * h
        grid = self.grid.to(dev).reshape(4, N).contiguous()
        f32 = lambda t: t.to(torch.float32).contiguous()
        depth, K, K_inv, RTinv_cam1, RT_cam2 = f32(depth), f32(K), f32(K_inv), f32(RTinv_cam1), f32(RT_cam2)
        zproj = torch.empty(bs, N, dtype=torch.float32, device=dev)
        ys, xs = torch.empty(bs, N, dtype=torch.int32, device=dev), torch.empty(bs, N, dtype=torch.int32, device=dev)
        flag = torch.empty(bs, N, dtype=torch.float32, device=dev)
        L = _lib.lib()
        _lib.check(L.ps_zbuffer_project_f32(_lib.ptr(depth), _lib.ptr(grid), _lib.ptr(K), _lib.ptr(K_inv), _lib.ptr(RTinv_cam1),
                                            _lib.ptr(RT_cam2), bs, w, _lib.ptr(zproj), _lib.ptr(ys), _lib.ptr(xs), _lib.ptr(flag),
                                            _lib.current_stream()), "ps_zbuffer_project_f32")
        order = zproj.sort(dim=1, descending=True, stable=True)[1].contiguous()      # (B,N) point at each sorted position (:68)
        out = torch.full((bs, 2, w, h), -2.0, device=dev, dtype=torch.float32)
        winner = torch.empty(bs, w, h, dtype=torch.int32, device=dev)
        _lib.check(L.ps_zbuffer_scatter_sorted_f32(_lib.ptr(order), _lib.ptr(ys), _lib.ptr(xs), _lib.ptr(grid), _lib.ptr(flag), bs, N,
                                                   w, h, _lib.ptr(out), _lib.ptr(winner), _lib.ptr(_lib.status_word(dev)),
                                                   _lib.current_stream()), "ps_zbuffer_scatter_sorted_f32")
        return out, (-zproj).view(bs, 1, w, h)
