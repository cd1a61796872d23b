"""Point-cloud reprojection on MI355X -- drop-in for the reference's
models/projection/z_buffer_manipulator.py:PtsManipulator (same constructor and method signatures).

project_pts / project_pts_cumulative run in csrc/splat.hip:k_project through the C ABI
(ps_project_pts_f32, ps_project_pts_cumulative_f32); forward_justpts uses the fused
ps_project_splat_f32 so the (B,N,3) cloud never leaves the scratch buffer.
"""
import torch
import torch.nn as nn

from .. import _lib
from ..layers.z_buffer_layers import ACCUMULATION, RasterizePointsXYsBlending, splat_workspace

EPS = 1e-2


def get_splatter(name, depth_values, opt=None, size=256, C=64, points_per_pixel=8):
    """z_buffer_manipulator.py:11-27."""
    if name == "xyblending":
        return RasterizePointsXYsBlending(C, learn_feature=opt.learn_default_feature, radius=opt.radius,
                                          size=size, points_per_pixel=points_per_pixel, opts=opt)
    raise NotImplementedError()


def _f32c(t):
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()


class PtsManipulator(nn.Module):
    def __init__(self, W, C=64, opt=None):
        super().__init__()
        self.opt = opt
        self.W = W
        self.splatter = get_splatter(opt.splatter, None, opt, size=W, C=C, points_per_pixel=opt.pp_pixel)
        # the reference's `xyzs` buffer (:38-48) is kept so that state_dicts line up -- rows (x, -y, -1, 1) of the
        # align-corners NDC grid, row-major; the kernels regenerate these values on the fly
        axis = torch.arange(W, dtype=torch.float32) / float(W - 1) * 2 - 1
        gx = axis.view(1, W).expand(W, W).reshape(-1)
        gy = axis.view(W, 1).expand(W, W).reshape(-1)
        one = torch.ones(W * W)
        self.register_buffer("xyzs", torch.stack((gx, -gy, -one, one)).unsqueeze(0))

    # ------------------------------------------------------------------ a2
    def project_pts(self, pts3D, K, K_inv, RT_cam1, RTinv_cam1, RT_cam2, RTinv_cam2):
        """Reference :50-83.  pts3D (B,1,N) depth -> sampler (B,3,N)."""
        _lib.require_cuda(pts3D, K, K_inv, RTinv_cam1, RT_cam2)
        B = pts3D.size(0)
        N = self.W * self.W
        assert pts3D.numel() == B * N, "project_pts expects one depth per grid point"
        depth = _f32c(pts3D)
        out = torch.empty(B, 3, N, dtype=torch.float32, device=depth.device)
        rc = _lib.lib().ps_project_pts_f32(_lib.ptr(depth), _lib.ptr(_f32c(K)), _lib.ptr(_f32c(K_inv)),
                                           _lib.ptr(_f32c(RTinv_cam1)), _lib.ptr(_f32c(RT_cam2)), B, self.W,
                                           _lib.ptr(out), _lib.current_stream())
        _lib.check(rc, "ps_project_pts_f32")
        return out

    # ------------------------------------------------------------------ a3
    def forward_justpts(self, src, pred_pts, K, K_inv, RT_cam1, RTinv_cam1, RT_cam2, RTinv_cam2):
        """Reference :85-107 -> (features (B,C,W,W), background_mask (B,W,W) bool)."""
        bs, c, w, h = src.size()
        if len(pred_pts.size()) > 3 and w == self.W and h == self.W:
            _lib.require_cuda(src, pred_pts, K, K_inv, RTinv_cam1, RT_cam2)
            sp = self.splatter
            S = self.W
            out = torch.empty(bs, c, S, S, dtype=torch.float32, device=src.device)
            bg = torch.empty(bs, S, S, dtype=torch.uint8, device=src.device)
            ws = splat_workspace(src.device, bs, S * S, S, sp.radius)
            rc = _lib.lib().ps_project_splat_f32(
                _lib.ptr(_f32c(pred_pts)), _lib.ptr(_f32c(src)), _lib.ptr(_f32c(K)), _lib.ptr(_f32c(K_inv)),
                _lib.ptr(_f32c(RTinv_cam1)), _lib.ptr(_f32c(RT_cam2)), bs, c, S, float(sp.radius),
                int(sp.points_per_pixel), float(sp._opt("tau", 1.0)), int(sp._opt("rad_pow", 2)),
                ACCUMULATION[sp._opt("accumulation", "alphacomposite")],
                int(sp._opt("background_smoothing_kernel_size", 13)), _lib.ptr(out), _lib.ptr(bg), _lib.ptr(ws),
                ws.numel(), _lib.current_stream())
            _lib.check(rc, "ps_project_splat_f32")
            return out, bg.view(torch.bool)    # (k_dilate writes 0 / 1: the same bytes are the boolean mask -- no conversion pass)
        if len(pred_pts.size()) > 3:
            pred_pts = pred_pts.view(bs, 1, -1)
            src = src.view(bs, c, -1)
        pts3D = self.project_pts(pred_pts, K, K_inv, RT_cam1, RTinv_cam1, RT_cam2, RTinv_cam2)
        pointcloud = pts3D.permute(0, 2, 1).contiguous()
        return self.splatter(pointcloud, src)

    # ------------------------------------------------------------------ a5
    def forward_justpts_cumulative(self, src1, pred_pts, K, K_inv, RT_cam1, RTinv_cam1, RT_cam2, RTinv_cam2,
                                   prior_point_cloud, src2, last_background_mask, RTinv_cam3):
        """Reference :184-219 -> (features, background_mask, new_point_cloud, src)."""
        bs, c = src1.shape[:2]
        mask_flat = None if last_background_mask is None else last_background_mask.view(bs, 1, -1)
        src = src1
        if pred_pts.dim() > 3:
            pred_pts = pred_pts.reshape(bs, 1, -1)
            src = src1.reshape(bs, c, -1)
            if src2 is not None:
                # only the points that fell on background last time are new; boolean gathers keep row-major order
                # (sized explicitly: a frame with no background at all contributes zero new points, where the
                # reference's view(bs, 1, -1) cannot infer a size)
                keep = mask_flat.bool()
                new_pts = pred_pts[keep]
                n_keep = new_pts.numel() // bs
                pred_pts = new_pts.view(bs, 1, n_keep)
                src = torch.cat([src[keep.expand(bs, c, -1)].view(bs, c, n_keep), src2.reshape(bs, c, -1)], dim=2)
        last_background_mask = mask_flat
        pts3D, new_point_cloud = self.project_pts_cumulative(
            pred_pts, K, K_inv, RT_cam1, RTinv_cam1, RT_cam2, RTinv_cam2, prior_point_cloud,
            last_background_mask, RTinv_cam3)
        pointcloud = pts3D.permute(0, 2, 1).contiguous()
        result, background_mask = self.splatter(pointcloud, src)
        return result, background_mask, new_point_cloud, src

    # ------------------------------------------------------------------ a4
    def project_pts_cumulative(self, pts3D, K, K_inv, RT_cam1, RTinv_cam1, RT_cam2, RTinv_cam2,
                               prior_point_cloud=None, last_background_mask=None, RTinv_cam3=None):
        """Reference :221-266 -> (sampler (B,3,NT), xy_proj (B,4,NT))."""
        _lib.require_cuda(pts3D, K, K_inv, RTinv_cam1, RT_cam2)
        B = pts3D.size(0)
        depth = _f32c(pts3D).view(B, pts3D.numel() // B)
        n_new = depth.size(1)
        new_index = None
        if last_background_mask is not None:
            m = last_background_mask.view(B, -1)
            # boolean-mask gather keeps row-major order (:226-228); equal counts per image as in the reference
            new_index = torch.nonzero(m, as_tuple=False)[:, 1].view(B, n_new).to(torch.int32).contiguous()
        n_prior = 0 if prior_point_cloud is None else prior_point_cloud.size(2)
        NT = n_new + n_prior
        sampler = torch.empty(B, 3, NT, dtype=torch.float32, device=depth.device)
        cloud = torch.empty(B, 4, NT, dtype=torch.float32, device=depth.device)
        prior = None if prior_point_cloud is None else _f32c(prior_point_cloud)
        rt3 = None if RTinv_cam3 is None else _f32c(RTinv_cam3)
        rc = _lib.lib().ps_project_pts_cumulative_f32(
            _lib.ptr(depth), _lib.ptr(new_index), _lib.ptr(prior), _lib.ptr(_f32c(K)), _lib.ptr(_f32c(K_inv)),
            _lib.ptr(_f32c(RTinv_cam1)), _lib.ptr(_f32c(RT_cam2)), _lib.ptr(rt3), B, self.W, n_new, n_prior,
            _lib.ptr(sampler), _lib.ptr(cloud), _lib.current_stream())
        _lib.check(rc, "ps_project_pts_cumulative_f32")
        return sampler, cloud
