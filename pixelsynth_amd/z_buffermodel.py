"""View-synthesis orchestration of the hot path -- counterpart of the reference's
models/z_buffermodel.py:ZbufferModelPts for the rows of SURVEY.md 8(a): target poses
(get_rt_from_rot :202-242), reprojection + splat (forward_justpts), generation order + masks
(get_masks_for_batch :641-701), autoregressive outpainting (get_best_sample -> sample()) and the
foreground/background blend (get_combined :703-708).

The dense networks the reference runs around that path (depth Unet, VQ-VAE-2 encode/decode, refinement
decoder, discriminator / Places365 ranking) are SURVEY 8(f) "next" rows, not built here: they are
injected as callables (`pts_regressor`, `vqvae`, `projector`); when absent, the batch must carry the
tensors they would have produced (`depths`, `codes`) -- that is how the synthetic benchmark drives it.
"""
import math
import types

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .lmconv.layers import PONO
from .lmconv.model import OurPixelCNN
from .lmconv.sample import sample
from .projection.z_buffer_manipulator import PtsManipulator


class ARPlan:
    """Device-resident, compact result of get_masks_for_batch for B images (see ps_ar_plan)."""

    def __init__(self, order_loc, region, mask_init, mask_undilated, mask_dilated, first_step, order_host, G):
        self.order_loc, self.region = order_loc, region
        self.mask_init, self.mask_undilated, self.mask_dilated = mask_init, mask_undilated, mask_dilated
        self.first_step = first_step
        self._order_host, self._G = order_host, G

    @property
    def gen_order(self):
        """list of (L,2) int arrays (row, col) by rank: the reference's gen_order (built on demand)."""
        G = self._G
        return [np.stack([o // G, o % G], 1).astype(np.int64) for o in self._order_host]

    @property
    def n_sampled(self):
        return self._n_sampled


def build_ar_plan(background_mask, G=32, device=None):
    """background_mask (B,S,S) bool/uint8 tensor (device or host) -> ARPlan on `device`.
    One device->host copy of the mask (the reference does four, z_buffermodel.py:662-669), integer
    work in C++ (csrc/host_order.cpp), one host->device copy per output."""
    device = device or (background_mask.device if background_mask.is_cuda else torch.device("cuda", torch.cuda.current_device()))
    bg = background_mask.to(torch.uint8).cpu().contiguous().numpy()
    B, S, _ = bg.shape
    L = G * G
    order_loc = np.empty((B, L), np.int32)
    region = np.empty((B, L), np.uint8)
    masks = [np.empty((B, 9, L), np.float32) for _ in range(3)]
    import ctypes
    first = ctypes.c_int32(0)
    rc = _lib.lib().ps_ar_plan(_lib.ptr(bg), B, S, G, _lib.ptr(order_loc), _lib.ptr(region), _lib.ptr(masks[0]),
                               _lib.ptr(masks[1]), _lib.ptr(masks[2]), ctypes.cast(ctypes.byref(first), ctypes.c_void_p))
    _lib.check(rc, "ps_ar_plan")
    up = lambda a: torch.from_numpy(a).to(device, non_blocking=True)
    plan = ARPlan(up(order_loc), up(region), up(masks[0]), up(masks[1]), up(masks[2]), int(first.value), order_loc, G)
    plan._n_sampled = region.sum(1).astype(int)
    return plan


class ZbufferModelPts(nn.Module):
    def __init__(self, opt, pts_regressor=None, vqvae=None, projector=None):
        super().__init__()
        self.opt = opt
        self.pts_regressor = pts_regressor
        if vqvae is None and getattr(opt, "vqvae", False):  # z_buffermodel.py:81-82
            from .vqvae2 import VQVAETop
            vqvae = VQVAETop()
        self.vqvae = vqvae
        self.projector = projector
        C = 3 if getattr(opt, "use_rgb_features", True) else 64
        self.pts_transformer = PtsManipulator(opt.W, C=C, opt=opt)
        self.num_classes = 512
        self.outpaint2 = OurPixelCNN(nr_resnet=2, nr_filters=80, input_channels=self.num_classes,
                                     nr_logistic_mix=10, kernel_size=(3, 3), max_dilation=2, weight_norm=False,
                                     feature_norm_op=lambda c: PONO(), dropout_prob=0, conv_bias=True,
                                     conv_mask_weight=False, rematerialize=False, binarize=False)  # :62-74
        self.args = types.SimpleNamespace(dataloader_seed=getattr(opt, "seed", 0), num_classes=self.num_classes)
        self.obs = [3, 32, 32]
        self.downsample = nn.AvgPool2d(kernel_size=8, stride=8)
        self.rotvecs = {'R': np.array([0, .6, 0]), 'L': np.array([0, -.6, 0]), 'U': np.array([-.3, 0, 0]),
                        'D': np.array([.3, 0, 0]), 'UR': np.array([-.15, .3, 0]), 'UL': np.array([-.15, -.3, 0]),
                        'DR': np.array([.15, .3, 0]), 'DL': np.array([.15, -.3, 0])}  # :113-114
        self.mapping = ['R', 'L', 'U', 'D', 'UL', 'UR', 'DR', 'DL']

    # ---------------------------------------------------------------- a15
    def eulerAnglesToRotationMatrix(self, theta):
        """z_buffermodel.py:186-200."""
        R_x = np.array([[1, 0, 0], [0, math.cos(theta[0]), -math.sin(theta[0])],
                        [0, math.sin(theta[0]), math.cos(theta[0])]])
        R_y = np.array([[math.cos(theta[1]), 0, math.sin(theta[1])], [0, 1, 0],
                        [-math.sin(theta[1]), 0, math.cos(theta[1])]])
        R_z = np.array([[math.cos(theta[2]), -math.sin(theta[2]), 0],
                        [math.sin(theta[2]), math.cos(theta[2]), 0], [0, 0, 1]])
        return np.dot(R_z, np.dot(R_y, R_x))

    def get_rt_from_rot(self, direction, input_RT, num=None, denom=None):
        """z_buffermodel.py:202-242 -> (new_output_RTinv, new_output_RT), same device as input_RT."""
        dev = input_RT.device
        if num is None:
            num = 0
        setting = getattr(self.opt, "model_setting", "gen_img")
        if setting in ('gen_two_imgs', 'gen_scene'):
            if direction == 'S':
                new_RT = torch.zeros_like(input_RT)
                new_RT[:, :, :3] = input_RT[:, :, :3]
                new_RT[:, 3, 3] = 1
                off = torch.tensor([np.sin(2 * np.pi * num / denom), np.cos(2 * np.pi * num / denom),
                                    .4 * np.sin(2 * np.pi * (.25 + num / denom))]).to(dev)
                new_RT[0, :3, 3] = input_RT[0, :3, 3] + .35 * off
                return torch.inverse(new_RT), new_RT
            elif direction == 'C':
                rotvec = np.array([0.2 * np.cos(2 * np.pi * num / denom), 0.2 * np.sin(2 * np.pi * num / denom), 0])
            else:
                rotvec = self.rotvecs[direction] * num / denom
        else:
            rotvec = self.rotvecs[direction] * self.opt.rotation / np.linalg.norm(self.rotvecs[direction])
        mtx = torch.zeros([1, 4, 4], device=dev)
        mtx[0, 3, 3] = 1
        mtx[0, :3, :3] = torch.tensor(self.eulerAnglesToRotationMatrix(rotvec)).to(torch.float32).to(dev)
        if getattr(self.opt, "homography", False) and direction not in ('C',):
            new_RT = torch.zeros([1, 4, 4], device=dev)
            new_RT[:, :, 3] = input_RT[:, :, 3]
            new_RT[:, :3, :3] = mtx[:, :3, :3].bmm(input_RT[:, :3, :3])
        else:
            new_RT = mtx.bmm(input_RT)
        return torch.inverse(new_RT), new_RT

    # ---------------------------------------------------------------- a7
    def get_masks_for_batch(self, output_RT, input_RTinv, background_mask, compact=False):
        """z_buffermodel.py:641-701.  Default return value matches the reference: masks repeated per input
        channel, (b*513,9,L), (b*160,9,L), (b*80,9,L), plus gen_order (list of (L,2) arrays).
        compact=True returns the ARPlan the HIP sampler consumes directly (one (b,9,L) copy per mask)."""
        plan = build_ar_plan(background_mask, self.obs[1])
        if compact:
            return plan
        b, L = plan.mask_init.shape[0], self.obs[1] * self.obs[2]
        rep = lambda m, c: m.unsqueeze(1).repeat(1, c, 1, 1).view(-1, 9, L)
        return rep(plan.mask_init, 513), rep(plan.mask_undilated, 160), rep(plan.mask_dilated, 80), plan.gen_order

    # ---------------------------------------------------------------- a14
    def get_combined(self, gen_fs, ar_sample, background_mask):
        """z_buffermodel.py:703-708."""
        b, h, w = background_mask.shape
        foreground_mask = (~background_mask).float()
        background_mask = background_mask.float()
        return gen_fs * foreground_mask.view(b, -1, h, w) + ar_sample * background_mask.view(b, -1, h, w)

    # ---------------------------------------------------------------- batched hot path (C3/C4/C5)
    @torch.no_grad()
    def outpaint_views(self, fs, depth, K, K_inv, input_RT, input_RTinv, output_RT, output_RTinv, codes,
                       temperature=0.7, uniforms=None, forced=None):
        """V independent novel views in one pass: reproject + splat (a2-a6), order + masks (a7-a9),
        AR outpainting of the 32x32 code grid (a13, fused device loop).
        fs (V,C,S,S), depth (V,1,S,S), cameras (V,4,4), codes (V,32,32) int (the VQ-VAE codes of the
        reprojected view; synthetic in the benchmark).  Returns dict(gen_fs, background_mask, codes, plan)."""
        gen_fs, background_mask = self.pts_transformer.forward_justpts(fs, depth, K, K_inv, input_RT, input_RTinv,
                                                                      output_RT, output_RTinv)
        plan = build_ar_plan(background_mask, self.obs[1])
        V = fs.shape[0]
        L = self.obs[1] * self.obs[2]
        if codes is None:  # z_buffermodel.py:345: the VQ-VAE top codes of the reprojected view
            codes = self.vqvae.encode_codes(gen_fs)
        c32 = codes.reshape(V, L).to(torch.int32).contiguous().clone()
        eng = self.outpaint2.engine(self.obs[1], self.obs[2], V)
        if forced is None and uniforms is None:
            uniforms = torch.rand(V, L, device=fs.device, dtype=torch.float32)
        eng.ar_run(c32, plan.order_loc, plan.region, plan.mask_init, plan.mask_undilated, plan.mask_dilated,
                   temperature=temperature, uniforms=uniforms, forced=forced, first_step=plan.first_step)
        return dict(gen_fs=gen_fs, background_mask=background_mask, codes=c32.view(V, self.obs[1], self.obs[2]),
                    plan=plan)

    # ---------------------------------------------------------------- reference-shaped single image path
    @torch.no_grad()
    def forward_image(self, batch, netD=None):
        """Hot-path part of forward_image (z_buffermodel.py:291-419) for model_setting gen_img / gen_paired_img.
        batch: {"images": [(B,3,S,S)], "cameras": [{"P","Pinv","K","Kinv"}], optional "depths": [(B,1,S,S)],
        optional "codes": (B,32,32)} -> (None, outputs dict with the reference's keys)."""
        dev = torch.device("cuda", torch.cuda.current_device())
        input_img = batch["images"][0].to(dev)
        cam = {k: v.to(dev) for k, v in batch["cameras"][0].items() if torch.is_tensor(v)}
        K, K_inv, input_RT, input_RTinv = cam["K"], cam["Kinv"], cam["P"], cam["Pinv"]
        output_RTinv, output_RT = self.get_rt_from_rot(self.opt.direction, input_RT)
        if self.pts_regressor is not None:
            regressed_pts = torch.sigmoid(self.pts_regressor(input_img)) * (self.opt.max_z - self.opt.min_z) + self.opt.min_z
        else:
            regressed_pts = batch["depths"][0].to(dev)
        fs = input_img
        gen_fs, background_mask = self.pts_transformer.forward_justpts(fs, regressed_pts, K, K_inv, input_RT,
                                                                      input_RTinv, output_RT, output_RTinv)
        masks_init, masks_undilated, masks_dilated, gen_order = self.get_masks_for_batch(output_RT, input_RTinv,
                                                                                         background_mask)
        if self.vqvae is not None:
            enc = getattr(self.vqvae, "encode_codes", None)      # our mirror: top codes only, int32, on the device
            downsampled_fs = enc(gen_fs) if enc is not None else self.vqvae.encode(gen_fs)[3]
        else:
            downsampled_fs = batch["codes"].to(dev)
        autoreg_output, _ = sample(self.outpaint2, gen_order, masks_init, masks_undilated, masks_dilated,
                                   downsampled_fs, self.obs, self.args, 0, self.opt.temperature,
                                   self.downsample(background_mask.float()))
        codes = torch.argmax(autoreg_output, dim=1)
        outputs = {"InputImg": input_img, "PredDepthImg": regressed_pts / 5 - 1,
                   "ForegroundImg": (~background_mask).repeat(input_img.shape[0], 1, 1, 1).float(),
                   "FeaturesImg": gen_fs, "PredCodes": codes}
        if self.vqvae is not None and self.projector is not None:
            ar_sample = self.vqvae.decode_code(codes.to(torch.int64))
            outputs["PredImg"] = self.projector(self.get_combined(gen_fs, ar_sample, background_mask), background_mask)
        return None, outputs
