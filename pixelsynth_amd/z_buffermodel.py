"""View-synthesis orchestration of the hot path -- counterpart of the reference's
models/z_buffermodel.py:ZbufferModelPts for the rows of SURVEY.md 8(a): target poses
(get_rt_from_rot :202-242), reprojection + splat (forward_justpts), generation order + masks
(get_masks_for_batch :641-701), autoregressive outpainting (get_best_sample -> sample()) and the
foreground/background blend (get_combined :703-708).

The dense networks the reference runs around that path are SURVEY 8(f) "next" rows: the VQ-VAE-2 top level
(pixelsynth_amd/vqvae2), the depth Unet and the refinement decoder (pixelsynth_amd/networks) are built when the
options name them (`vqvae`, `norm_G` + `refine_model_type`) or can be injected (`pts_regressor`, `vqvae`,
`projector`); the discriminator / Places365 classifier of the sample ranking are injected only.  When a network is
absent the batch must carry the tensors it would have produced (`depths`, `codes`) -- that is how the synthetic
benchmark drives the hot path on its own.
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
from .networks.architectures import check_f16x3_overflow, clear_f16x3_overflow, decoder_conv
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

    # The wavefront schedule of ONE first step for the whole batch: (cols on the device, wave_start on the host) and the columns on the
    # host.  A plan that carries the schedule of per-frame prefixes (waves_frames: what the batched paths run) builds this one on first
    # use -- 3.6 of the 9.8 ms of host work per 128-view plan, and only the measurement / parity callers ask for it.
    def _schedule(self):
        if self._waves is None:
            from .lmconv.model import wavefronts
            w = wavefronts(self._order_host, self._G, self._G, self.first_step, self.order_loc.device, keep_host=True)
            self._waves, self._waves_host = w[:2], w[2]
        return self._waves, self._waves_host

    _waves = _waves_host = None

    @property
    def waves(self):
        return self._schedule()[0]

    @waves.setter
    def waves(self, value):
        self._waves = value

    @property
    def waves_host(self):
        return self._schedule()[1]


import collections
import os
import threading

_PREFIX_STREAMS = {}    # (device, n) -> the prefix pass's side streams (ZbufferModelPts._prefix_streams)
PER_FRAME_PREFIX = os.environ.get("PS_PER_FRAME_PREFIX", "1") != "0"   # plans also carry the schedule of per-frame prefixes (waves_frames)

_PINNED = collections.OrderedDict()
_PINNED_MAX = 12                       # staging buffers kept (four per batch shape): the oldest shapes are released
_PLAN_LOCK = threading.RLock()         # the staging buffers are shared state: one plan is staged at a time per process


def _pinned(name, shape, dtype):
    """Page-locked staging buffers (pageable copies ran at ~0.6 GB/s on the MI355X hosts), kept per (name, shape) in a small
    LRU: callers that vary their batch size do not pile up pinned host memory.  Used under _PLAN_LOCK."""
    key = (name, tuple(shape), dtype)
    t = _PINNED.pop(key, None)
    if t is None:
        t = torch.empty(shape, dtype=dtype, pin_memory=True)
    _PINNED[key] = t
    while len(_PINNED) > _PINNED_MAX:
        _PINNED.popitem(last=False)
    return t


def build_ar_plan(background_mask, G=32, device=None):
    """background_mask (B,S,S) bool/uint8 tensor (device or host) -> ARPlan on `device`.
    One device->host copy of the mask (the reference does four, z_buffermodel.py:662-669), the integer work (pooling,
    distance transforms, generation order) in C++ on the host (csrc/host_order.cpp), the orders back up, and the three
    kernel masks built from them on the device (ps_order_masks_f32) -- nothing bigger than the orders crosses PCIe."""
    with _PLAN_LOCK:
        return _build_ar_plan(background_mask, G, device)


def _build_ar_plan(background_mask, G, device):
    import ctypes
    device = device or (background_mask.device if background_mask.is_cuda else torch.device("cuda", torch.cuda.current_device()))
    B, S, _ = background_mask.shape
    L = G * G
    if background_mask.is_cuda:
        stage = _pinned("bg", (B, S, S), torch.uint8)
        as_u8 = (background_mask.view(torch.uint8) if background_mask.dtype == torch.bool and background_mask.is_contiguous()
                 else background_mask.to(torch.uint8))      # (a bool mask IS bytes of 0 / 1: no conversion pass in front of the copy)
        stage.copy_(as_u8, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        bg = stage.numpy()
    else:
        bg = background_mask.to(torch.uint8).contiguous().numpy()
    order_t, region_t = _pinned("order", (B, L), torch.int32), _pinned("region", (B, L), torch.uint8)
    order_loc, region = order_t.numpy(), region_t.numpy()
    first = ctypes.c_int32(0)
    rc = _lib.lib().ps_ar_plan(_lib.ptr(bg), B, S, G, _lib.ptr(order_loc), _lib.ptr(region), None, None, None,
                               ctypes.cast(ctypes.byref(first), ctypes.c_void_p))
    _lib.check(rc, "ps_ar_plan")
    d_order, d_region = order_t.to(device, non_blocking=True), region_t.to(device, non_blocking=True)
    masks = [torch.empty(B, 9, L, dtype=torch.float32, device=device) for _ in range(3)]
    rc = _lib.lib().ps_order_masks_f32(_lib.ptr(d_order), B, G, G, _lib.ptr(masks[0]), _lib.ptr(masks[1]), _lib.ptr(masks[2]),
                                       _lib.ptr(_lib.status_word(device)), _lib.current_stream())
    _lib.check(rc, "ps_order_masks_f32")
    order_host = order_loc.copy()       # (the staging buffer is reused by the next plan)
    plan = ARPlan(d_order, d_region, masks[0], masks[1], masks[2], int(first.value), order_host, G)
    plan._n_sampled = region.sum(1).astype(int)
    from .lmconv.model import wavefronts
    # per-frame prefixes: a frame's first SAMPLED position (L: none) -- the observed positions in front of it need no column
    sampled = np.take_along_axis(region, order_loc.astype(np.int64), 1) != 0
    plan.first_steps = np.where(sampled.any(1), sampled.argmax(1), L).astype(np.int32)
    if PER_FRAME_PREFIX and int(plan.first_steps.min()) >= plan.first_step and int(plan.first_steps.max()) > plan.first_step:
        fs_t = _pinned("first_steps", (B,), torch.int32)
        fs_t.numpy()[:] = plan.first_steps
        plan.first_steps_dev = fs_t.to(device, non_blocking=True)
        w = wavefronts(order_host, G, G, plan.first_step, device, keep_host=True, first_steps=plan.first_steps)
        plan.waves_frames = w[:2]
    from .lmconv.model import TP_MIN_FRAMES
    if getattr(plan, "waves_frames", None) is None or B < TP_MIN_FRAMES:
        plan._schedule()   # (no per-frame schedule, or a small batch, whose outpaint_planned runs this one: built here, off the AR stream)
    _lib.read_status("ps_order_masks_f32", device)   # synchronises: the staging buffers are free again, and a bad order is an error
    return plan


def plan_from_reference_args(gen_order, masks, sample_region, G=32, device=None):
    """The ARPlan of values in the REFERENCE's form (what get_masks_for_batch returns without compact=True and what
    get_best_sample / sample() are handed, z_buffermodel.py:244-248): gen_order = list of (L,2) (row, col) arrays by rank,
    masks = (masks_init (b*513,9,L), masks_undilated (b*160,9,L), masks_dilated (b*80,9,L)) or their compact (b,9,L) forms,
    sample_region (b,G,G) = self.downsample(background_mask.float()): a block is sampled where it equals 1 (sample.py:24-41)."""
    from .lmconv.locally_masked_convolution import compact_mask
    from .lmconv.model import wavefronts
    device = device or torch.device("cuda", torch.cuda.current_device())
    B, L = len(gen_order), G * G
    order_host = np.stack([np.asarray(g, np.int64)[:, 0] * G + np.asarray(g, np.int64)[:, 1] for g in gen_order]).astype(np.int32)
    region_host = (sample_region.detach().reshape(B, L).cpu().numpy() == 1).astype(np.uint8)
    first = L
    for b in range(B):
        hit = np.nonzero(region_host[b][order_host[b]])[0]
        if hit.size:
            first = min(first, int(hit[0]))
    m = [compact_mask(t.to(device), B, c).to(torch.float32) for t, c in zip(masks, (513, 160, 80))]
    m = [(t.expand(B, -1, -1) if t.size(0) == 1 and B > 1 else t).contiguous() for t in m]
    plan = ARPlan(torch.from_numpy(order_host).to(device), torch.from_numpy(region_host).to(device), m[0], m[1], m[2], first,
                  order_host, G)
    plan._n_sampled = region_host.sum(1).astype(int)
    plan.waves = wavefronts(order_host, G, G, first, device)
    return plan


def rank_samples(discrim_scores, entropy_scores):
    """Index of the sample get_best_sample keeps (z_buffermodel.py:266-276): samples are ranked by discriminator score
    (ascending) and by classifier entropy (ascending); total = .5*(n-1-entropy_rank) + .5*discrim_rank; the first
    arg-max wins.  Ranks come from numpy's default argsort, as there."""
    n = len(discrim_scores)
    by_disc, by_entr = np.argsort(np.asarray(discrim_scores)), np.argsort(np.asarray(entropy_scores))
    disc_rank, entr_rank = np.empty(n, np.int64), np.empty(n, np.int64)
    disc_rank[by_disc] = np.arange(n)
    entr_rank[by_entr] = np.arange(n)
    return int(np.argmax(.5 * (n - 1 - entr_rank) + .5 * disc_rank))


class _SceneState:
    """What forward_scene carries from one rendered frame to the next (z_buffermodel.py:436-443)."""

    def __init__(self, img):
        self.img = img                # the frame the next one is rendered from
        self.cloud = None             # (1,4,N) every point so far, in the camera of the last rendered frame
        self.feats = None             # (1,C,N) their features
        self.background = None        # (1,S,S) background mask of the last rendered frame
        self.out_RTinv = None         # inverse pose of the last rendered frame
        self.numerator = None
        self.direction = None


class ZbufferModelPts(nn.Module):
    def __init__(self, opt, pts_regressor=None, vqvae=None, projector=None, encoder=None, classifier=None):
        super().__init__()
        self.opt = opt
        if pts_regressor is None and hasattr(opt, "norm_G"):  # z_buffermodel.py:41-44
            from .networks import Unet
            extra = {"num_filters": opt.Unet_num_filters} if hasattr(opt, "Unet_num_filters") else {}
            pts_regressor = Unet(channels_in=3, channels_out=1, opt=opt, **extra)
        if projector is None and "resnet" in getattr(opt, "refine_model_type", "") and hasattr(opt, "norm_G"):  # :90
            from .networks import get_decoder
            projector = get_decoder(opt)
        self.pts_regressor = pts_regressor
        self.encoder = encoder        # feature encoder when use_rgb_features is off (injected)
        if classifier is None and max(int(getattr(opt, "num_samples", 1)), 1) > 1:   # z_buffermodel.py:88 (random init until the
            from .networks import resnet18                                             # Places365 state_dict is loaded, demo.py:233-243)
            classifier = resnet18(num_classes=365)
        self.classifier = classifier  # Places365 ResNet-18 of get_best_sample
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
        self.sample_batch = 32        # frames (candidates x views) a get_best_sample engine run takes at most

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

    # ---------------------------------------------------------------- depth of the source image (forward_image :303-311, :606-612)
    def regress_depth(self, input_img, given=None):
        """The reference's depth rule, ONE place for every entry point: the regressor's sigmoid scaled to [min_z, max_z], or
        1 / (10 sigmoid + 0.01) with opt.use_inverse_depth (landscape datasets), or the given depth with opt.use_gt_depth --
        and `given` stands in for the regressor when the model has none (the synthetic benchmark)."""
        if self.pts_regressor is None or getattr(self.opt, "use_gt_depth", False):
            if given is None:
                raise ValueError("no depth regressor (or opt.use_gt_depth): the batch must carry the depth")
            return given
        raw = torch.sigmoid(self.pts_regressor(input_img))
        if getattr(self.opt, "use_inverse_depth", False):
            return 1. / (raw * 10 + 0.01)
        return raw * (self.opt.max_z - self.opt.min_z) + self.opt.min_z

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
    def plan_views(self, fs, depth, K, K_inv, input_RT, input_RTinv, output_RT, output_RTinv):
        """First half of outpaint_views: reproject + splat (a2-a6) on the current stream, then the host part -- the
        background masks come back, generation orders, kernel masks and the wavefront schedule are built (a7-a9) and
        uploaded.  Ends with everything the AR run needs resident on the device.
        -> dict(gen_fs, background_mask, plan)."""
        gen_fs, background_mask = self.pts_transformer.forward_justpts(fs, depth, K, K_inv, input_RT, input_RTinv,
                                                                      output_RT, output_RTinv)
        return dict(gen_fs=gen_fs, background_mask=background_mask, plan=build_ar_plan(background_mask, self.obs[1]))

    @staticmethod
    def adopt_planned(planned, stream):
        """A plan made on a side stream is about to be consumed on `stream`: tell the caching allocator, so that the
        plan's buffers are not recycled on the side stream while work queued on `stream` still reads them."""
        plan = planned["plan"]
        more = (plan.waves_frames[0], plan.first_steps_dev) if getattr(plan, "waves_frames", None) is not None else ()
        for t in (planned["gen_fs"], planned["background_mask"], plan.order_loc, plan.region, plan.mask_init,
                  plan.mask_undilated, plan.mask_dilated) + ((plan._waves[0],) if plan._waves is not None else ()) + more:
            if t.numel():
                t.record_stream(stream)

    @torch.no_grad()
    def outpaint_planned(self, planned, codes, temperature=0.7, uniforms=None, forced=None, between=None):
        """Second half: AR outpainting of the 32x32 code grids (a13) of the views prepared by plan_views; asynchronous
        on the current stream.  Adds `codes` (V,32,32) int32 to the dict and returns it.
        between: a callable run on the current stream BETWEEN the whole-grid prefix pass and the first column launch (the two
        halves of the AR run, ps_pixelcnn_ar_prefix / ps_pixelcnn_ar_columns).  bench.py makes the stream wait there
        for the previous step's asynchronous frame gather: the collective's kernels then run beside the prefix pass -- whose
        small workgroups fit around them -- and are through before a column launch asks for every compute unit."""
        gen_fs, plan = planned["gen_fs"], planned["plan"]
        V = gen_fs.shape[0]
        L = self.obs[1] * self.obs[2]
        if codes is None:  # z_buffermodel.py:345: the VQ-VAE top codes of the reprojected view
            codes = self.vqvae.encode_codes(gen_fs)
        c32 = codes.reshape(V, L).to(torch.int32).contiguous().clone()
        eng = self.outpaint2.engine(self.obs[1], self.obs[2], V)
        if forced is None and uniforms is None:
            uniforms = torch.rand(V, L, device=gen_fs.device, dtype=torch.float32)
        nsplit = self._prefix_split(V, busy=between is not None)
        # per-frame prefixes where the plan carries their schedule (build_ar_plan): the whole-grid pass takes every frame up to ITS first
        # sampled position, the columns start there (ps_pixelcnn_ar_prefix_frames / ps_ar_wavefronts_frames: the same codes)
        # (batches of the throughput form only: the launches of a small batch are bound by their latency, not by their columns --
        # 16 views: 5.55 ms per step with one prefix for the batch, 5.66 with per-frame ones)
        from .lmconv.model import TP_MIN_FRAMES
        waves_f = getattr(plan, "waves_frames", None) if self.PER_FRAME_PREFIX and V >= TP_MIN_FRAMES else None
        pf = dict(first_steps=plan.first_steps_dev, max_first_step=int(plan.first_steps.max())) if waves_f is not None else {}
        waves = waves_f if waves_f is not None else plan.waves
        if (between is None and nsplit == 1 and waves_f is None) or plan.first_step >= L:   # (nothing to walk: only the whole-grid pass runs)
            eng.ar_run(c32, plan.order_loc, plan.region, plan.mask_init, plan.mask_undilated, plan.mask_dilated,
                       temperature=temperature, uniforms=uniforms, forced=forced, first_step=plan.first_step, waves=plan.waves)
            if between is not None:
                between()
        else:
            args = (c32, plan.order_loc, plan.region, plan.mask_init, plan.mask_undilated, plan.mask_dilated, plan.first_step)
            if nsplit == 1:
                eng.ar_prefix(*args, **pf)
            else:
                # The whole-grid prefix pass of disjoint frame ranges on streams of their own (ps_pixelcnn_ar_prefix is built for it: every
                # range has its part of the scratch): a launch empties over its last tenth, and the next stage's launch cannot start
                # before it has -- with a second range's launches in flight, their workgroups take the places as they fall free.
                main = torch.cuda.current_stream()
                ready = torch.cuda.Event()
                ready.record(main)
                per = V // nsplit
                for k, st in enumerate(self._prefix_streams(nsplit - 1, c32.device)):
                    st.wait_event(ready)
                    with torch.cuda.stream(st):
                        eng.ar_prefix(*args, frame_begin=(k + 1) * per, frame_end=(k + 2) * per if k + 2 < nsplit else V, **pf)
                    for t in (c32,) + args[1:6] + ((pf["first_steps"],) if pf else ()):
                        t.record_stream(st)
                eng.ar_prefix(*args, frame_begin=0, frame_end=per, **pf)
                for st in self._prefix_streams(nsplit - 1, c32.device):
                    main.wait_stream(st)
            if between is not None:
                between()
            eng.ar_columns(c32, plan.order_loc, plan.region, plan.mask_init, plan.mask_undilated, plan.mask_dilated, waves,
                           temperature=temperature, uniforms=uniforms, forced=forced, first_step=plan.first_step)
        planned["codes"] = c32.view(V, self.obs[1], self.obs[2])
        return planned

    # ---------------------------------------------------------------- the AR runs of consecutive batches, overlapped
    PIPE_CAP = 1024         # columns a merged launch takes (lmconv.model.COLUMNS_PER_LAUNCH_TP)
    PER_FRAME_PREFIX = True  # outpaint_pipelined: per-frame prefixes where the plan carries their schedule (build_ar_plan, PS_PER_FRAME_PREFIX)
    PIPE_DEPTH = 4          # batches outpaint_pipelined keeps in flight at least (pipe_depth): every launch takes what is left of each batch's
    #                         current wavefront, oldest batch first, while there is room (lmconv.model.pack_launches) -- with three to four in
    #                         flight the launches of a large batch are full (C5's 128 views: 33 launches of ~1 020 columns per step
    #                         where head / tail merging ran 45 of 750; 16 views: 33 of 128 where equal parts ran 44)
    PIPE_FRAMES = 384       # ... and as many as it takes to have about this many frames in the handle, eight at most: the throughput form's
    #                         launches take 1 024 columns, which middle-sized batches only fill with more of them in flight (ms per step with
    #                         4 / 6 / 8 in flight -- 24 views: 5.08 / 4.39 / 4.10, 32: 5.21 / 4.53 / 4.36, 48: 6.23 / 5.80 / 5.73, 64: 6.82 / 6.48,
    #                         96: 9.04 / 8.93; 128: the same from 4 on; 16, latency form: 3.38 / 3.41 / 3.52)

    def pipe_depth(self, V):
        """Batches of V views that outpaint_pipelined keeps in flight at most (PS_PIPE_DEPTH overrides): the frames of its engine handle
        are that many batches'; a batch's result comes back at most depth - 1 calls late."""
        import os
        from .lmconv.model import TP_MIN_FRAMES
        d = os.environ.get("PS_PIPE_DEPTH")
        if d:
            return max(2, min(8, int(d)))
        if V < TP_MIN_FRAMES:        # (latency form: launches of 128 columns, full at four)
            return self.PIPE_DEPTH
        return max(self.PIPE_DEPTH, min(8, int(round(self.PIPE_FRAMES / V))))

    def pipe_frames(self, V):
        """Frames of the engine handle outpaint_pipelined runs batches of V views in."""
        return self.pipe_depth(V) * V

    def _pipe_buffers(self, V, device):
        """The batches in flight live in ONE engine handle of depth x V frames: batch i in frames [V (i % depth), ...).  The per-frame
        arrays of the C ABI (codes, order, region, the three masks, uniforms) are persistent (depth x V, ...) tensors; a batch's plan is
        copied into its share on the stream of the AR run (14 MB, ~10 us), so that planning on a side stream never writes under a
        launch that still reads another batch's share."""
        st = self.__dict__.get("_pipe")
        D = self.pipe_depth(V)
        if st is not None and (st["V"] != V or st["device"] != device or st["depth"] != D):
            if st["inflight"] or st["done"]:
                raise RuntimeError("outpaint_pipelined: a batch of another size is still in flight (call outpaint_flush first)")
            st = None
        if st is None:
            L = self.obs[1] * self.obs[2]
            z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=device)
            st = self.__dict__["_pipe"] = dict(
                V=V, depth=D, device=device, inflight=[], done=[], seq=0, next_out=0, codes=z((D * V, L), torch.int32), order=z((D * V, L), torch.int32),
                region=z((D * V, L), torch.uint8), masks=[z((D * V, 9, L), torch.float32) for _ in range(3)],
                uniforms=z((D * V, L), torch.float32), first_steps=z((D * V,), torch.int32),
                offset=[torch.tensor([k * V, 0], dtype=torch.int32, device=device) for k in range(D)])
            st["order"][:] = torch.arange(L, device=device, dtype=torch.int32)   # (a frame nobody has planned yet still holds a permutation)
        return st

    @torch.no_grad()
    def outpaint_pipelined(self, planned, codes, temperature=0.7, uniforms=None, between=None):
        """outpaint_planned for callers with a STREAM of batches of V views (bench.py, driver.py).  A batch's wavefronts grow to the
        launch capacity and shrink to a handful of columns, and a launch costs its 33 dependent stages whatever it holds: up to
        pipe_depth(V) batches are resident in ONE engine handle of pipe_frames(V) frames, and every column launch takes what is left of
        each batch's current wavefront, oldest batch first, while there is room (lmconv.model.pack_launches) -- C5's 128 views: 33
        launches of ~1 020 columns per step instead of 45 of 750 (head / tail merging of two batches, round 5) or 91 of a batch alone.
        Every column still runs behind the columns it reads (a batch moves on to its next wavefront only in the launch after the one
        that took the last of the current one), so the codes are outpaint_planned's, bit for bit (tests/test_zbuffermodel_gpu.py,
        tests/test_config_size_gpu.py).  Asynchronous on the current stream.
        -> the dict (codes added) of the next batch, in submission order, that is complete -- at most pipe_depth(V) - 1 calls late -- or
        None; outpaint_flush() runs what is left.  between: as for outpaint_planned."""
        gen_fs, plan = planned["gen_fs"], planned["plan"]
        V, G = gen_fs.shape[0], self.obs[1]
        L = G * self.obs[2]
        st = self._pipe_buffers(V, gen_fs.device)
        D = st["depth"]
        if codes is None:
            codes = self.vqvae.encode_codes(gen_fs)
        if uniforms is None:
            uniforms = torch.rand(V, L, device=gen_fs.device, dtype=torch.float32)
        eng = self.outpaint2.engine(G, self.obs[2], D * V)
        args = (st["codes"], st["order"], st["region"], st["masks"][0], st["masks"][1], st["masks"][2])
        if st["inflight"] and st["inflight"][0]["temperature"] != temperature:
            # what is in flight was planned with ANOTHER temperature: it cannot ride in this batch's launches (a launch has one
            # temperature), so it is finished now, as launches of its own, with its own -- the codes stay those of outpaint_planned
            self._pipe_step(eng, st, args, V, drain=True)
        h = min(set(range(D)) - {b["slot"] for b in st["inflight"]})      # a share of the handle nobody in flight lives in
        lo, hi = h * V, (h + 1) * V
        # (as elementwise kernels, not Tensor.copy_: same-type copies go through hipMemcpyAsync, which on the stream of the AR run stalled
        # for ~60 ms every few steps)
        put = lambda dst, src: torch.add(src, 0, out=dst) if src.dtype == dst.dtype else dst.copy_(src)
        put(st["codes"][lo:hi], codes.reshape(V, L))
        put(st["order"][lo:hi], plan.order_loc)
        put(st["region"][lo:hi], plan.region)
        for dst, src in zip(st["masks"], (plan.mask_init, plan.mask_undilated, plan.mask_dilated)):
            put(dst[lo:hi], src.expand(V, -1, -1) if src.size(0) == 1 else src)
        put(st["uniforms"][lo:hi], uniforms)
        # PER-FRAME prefixes (plans that carry their schedule): the whole-grid pass takes every frame up to ITS first sampled position --
        # a location costs it half of what a column costs, and the bits are the same
        waves = getattr(plan, "waves_frames", None) if self.PER_FRAME_PREFIX else None
        pf = {}
        if waves is not None:
            put(st["first_steps"][lo:hi], plan.first_steps_dev)
            pf = dict(first_steps=st["first_steps"], max_first_step=int(plan.first_steps.max()))
        else:
            waves = plan.waves
        # the prefix pass of this batch's frames (two ranges on two streams, as in outpaint_planned)
        nsplit = self._prefix_split(V, busy=between is not None)
        if nsplit == 1:
            eng.ar_prefix(*args, plan.first_step, frame_begin=lo, frame_end=hi, **pf)
        else:
            main = torch.cuda.current_stream()
            ready = torch.cuda.Event()
            ready.record(main)
            per = V // nsplit
            for k, side in enumerate(self._prefix_streams(nsplit - 1, gen_fs.device)):
                side.wait_event(ready)
                with torch.cuda.stream(side):
                    eng.ar_prefix(*args, plan.first_step, frame_begin=lo + (k + 1) * per, frame_end=lo + (k + 2) * per if k + 2 < nsplit else hi, **pf)
            eng.ar_prefix(*args, plan.first_step, frame_begin=lo, frame_end=lo + per, **pf)
            for side in self._prefix_streams(nsplit - 1, gen_fs.device):
                main.wait_stream(side)
        if between is not None:
            between()
        # this batch's schedule, in the handle's frame numbering, joins the batches in flight.  The columns are on the device already (the
        # plan's upload); a call's launches are put together THERE, from slices of the batches' columns (one concatenation) -- nothing
        # crosses PCIe on the stream of the AR run.
        ws = np.asarray(waves[1])
        dcols = waves[0] + st["offset"][h] if h else waves[0]
        st["inflight"].append(dict(planned=planned, cols=dcols, ws=ws, w=0, off=0, first_step=plan.first_step, slot=h, temperature=temperature,
                                   seq=st["seq"]))
        st["seq"] += 1
        self._pipe_step(eng, st, args, V)
        return self._pipe_pop(st)

    @staticmethod
    def _pipe_pop(st):
        """The next batch in SUBMISSION order, if it is complete (a short batch may be through before an older, longer one: it waits)."""
        if st["done"] and st["done"][0][0] == st["next_out"]:
            st["next_out"] += 1
            return st["done"].pop(0)[1]
        return None

    def _pipe_step(self, eng, st, args, V, drain=False):
        """One call's launches out of the batches in flight (lmconv.model.pack_launches: every launch takes what is left of each batch's
        current wavefront, oldest first, under the launch capacity).  With the handle full -- `depth` batches in flight -- launches run
        until the oldest batch is complete (one batch in, one out: the steady state); while it fills, a `depth`-th of the new batch's
        wavefronts' worth; drain: until nothing is left.  Batches whose last column has been queued are complete: their codes are
        taken out of the handle behind the launches."""
        from .lmconv.model import launch_capacity, pack_launches
        infl = st["inflight"]
        cap = min(int(os.environ.get("PS_PIPE_CAP", self.PIPE_CAP)), self.PIPE_CAP, launch_capacity(V))
        while infl:
            first = min(b["first_step"] for b in infl)
            temperature = infl[0]["temperature"]
            full = len(infl) >= st["depth"]
            slices, starts = pack_launches(infl, cap, until_oldest_done=True,
                                           budget=None if (drain or full) else -(-(len(infl[-1]["ws"]) - 1) // st["depth"]))
            if slices:
                cols = torch.cat([infl[k]["cols"][a:b] for k, a, b in slices])
                self._pipe_columns(eng, st, args, cols, starts, first, temperature)
            finished = [b for b in infl if b["w"] >= len(b["ws"]) - 1]
            infl[:] = [b for b in infl if b["w"] < len(b["ws"]) - 1]
            st["done"] = sorted(st["done"] + [(b["seq"], self._pipe_done(st, b)) for b in finished], key=lambda t: t[0])
            if not drain:
                break

    def _pipe_columns(self, eng, st, args, cols, ws, first, temperature):
        if len(ws) > 1 and ws[-1] > 0:
            eng.ar_columns(*args, (cols.contiguous(), np.ascontiguousarray(ws, np.int32)), temperature=temperature, uniforms=st["uniforms"],
                           first_step=int(first))

    def _pipe_done(self, st, b):
        """The batch whose last columns have just been queued: its codes out of the handle's share."""
        V, lo = st["V"], b["slot"] * st["V"]
        out = b["planned"]
        out["codes"] = st["codes"][lo:lo + V].clone().view(V, self.obs[1], self.obs[2])
        return out

    @torch.no_grad()
    def outpaint_flush(self):
        """What outpaint_pipelined still holds: the remaining parts of the batches in flight, as launches of their own
        -> the dicts of the batches not handed back yet, oldest first ([] when there is none)."""
        st = self.__dict__.get("_pipe")
        if st is None:
            return []
        if st["inflight"]:
            eng = self.outpaint2.engine(self.obs[1], self.obs[2], st["depth"] * st["V"])
            args = (st["codes"], st["order"], st["region"], st["masks"][0], st["masks"][1], st["masks"][2])
            self._pipe_step(eng, st, args, st["V"], drain=True)
        out, st["done"] = [d for _, d in st["done"]], []
        st["next_out"] = st["seq"]
        return out

    def outpaint_reset(self):
        """Forget the batches outpaint_pipelined still holds (their remaining wavefronts never run; their codes are lost).  For a caller
        whose sequence of batches was cut short by an exception: without this the NEXT sequence of the same batch size would merge the
        stale batches into its first launches and get their dicts back as its first results (driver.render_pipelined and bench.py call
        it on their way out of a failed run)."""
        st = self.__dict__.get("_pipe")
        if st is not None:
            st["inflight"], st["done"], st["next_out"] = [], [], st["seq"]

    PREFIX_SPLIT_MIN_VIEWS = 64   # below this a launch of half the frames no longer fills the chip
    PREFIX_STREAMS = 2            # 128 views: 18.16 -> 17.95 ms per step (three alternating pairs); 4 ranges lose (18.59)

    def _prefix_split(self, V, busy=False):
        """Frame ranges the prefix pass of a V-view batch is dealt to (each on a stream of its own): PREFIX_STREAMS, or
        PS_PREFIX_STREAMS from the environment; 1 for batches too small to fill the chip twice over or not a multiple of 8 frames
        per range (a range's frames are dealt to the 8 XCDs).  busy: collectives are in flight beside the AR run (the caller passed
        between=) -- with the runtime's default of four hardware queues one more stream then shares a queue with another and the step
        gets SLOWER (19.5 against 18.4 ms), so the pass is split only when the process runs with GPU_MAX_HW_QUEUES >= 8 (18.1 ms;
        bench.py sets it, tools/hwq_ab.sh measured it)."""
        import os
        n = int(os.environ.get("PS_PREFIX_STREAMS", self.PREFIX_STREAMS))
        if busy and "PS_PREFIX_STREAMS" not in os.environ and int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) < 8:
            n = 1
        return n if n > 1 and V >= self.PREFIX_SPLIT_MIN_VIEWS and V % (8 * n) == 0 else 1

    def _prefix_streams(self, n, device):
        """n side streams for the prefix pass, created once per PROCESS and device: which hardware queue a stream lands on is dealt at
        creation, one in eight shares the main stream's (docs/LAB_NOTEBOOK.md, "Which stream the side stream is") -- a process that
        builds several models one after the other (bench.py's side configurations) must not draw a new lot with each of them
        (round 6: C4's 64-frame circle 8.8 ms per step inside the long default bench run, 6.9 as a run of its own)."""
        key = (str(device), n)
        cache = _PREFIX_STREAMS
        if key not in cache:
            import os
            skip = int(os.environ.get("PS_PREFIX_STREAM_SKIP", "0"))    # tuning: streams created (and kept) in front of them
            cache[("skip",) + key] = [torch.cuda.Stream(device=device) for _ in range(skip)]
            cache[key] = [torch.cuda.Stream(device=device) for _ in range(n)]
        return cache[key]

    def outpaint_views(self, fs, depth, K, K_inv, input_RT, input_RTinv, output_RT, output_RTinv, codes,
                       temperature=0.7, uniforms=None, forced=None, check=True):
        """V independent novel views in one pass: reproject + splat (a2-a6), order + masks (a7-a9),
        AR outpainting of the 32x32 code grid (a13, fused device loop).
        fs (V,C,S,S), depth (V,1,S,S), cameras (V,4,4), codes (V,32,32) int (the VQ-VAE codes of the
        reprojected view; synthetic in the benchmark).  Returns dict(gen_fs, background_mask, codes, plan).
        Callers with several batches can overlap the host part of the next batch with the AR run of this one:
        plan_views on a side stream while outpaint_planned runs (bench.py, driver.py do)."""
        planned = self.plan_views(fs, depth, K, K_inv, input_RT, input_RTinv, output_RT, output_RTinv)
        out = self.outpaint_planned(planned, codes, temperature, uniforms, forced)
        if check:   # synchronises; pipelined callers (plan_views / outpaint_planned) pass check=False and ask the engine once
            self.outpaint2.engine(self.obs[1], self.obs[2], fs.shape[0]).check()
        return out

    @torch.no_grad()
    def synthesize_views(self, src_imgs, view_src, K, K_inv, input_RT, input_RTinv, output_RT, output_RTinv, temperature=None,
                         uniforms=None, depths=None, check=True):
        """END TO END for V independent (source, target view) pairs in one pass -- the batched counterpart of forward_image
        (z_buffermodel.py:291-419, num_samples = 1): depth Unet on the n_src SOURCE images (once per source, not per view),
        reproject + splat, VQ-VAE top codes of the reprojected views, AR outpainting, decode_code, get_combined, refinement
        decoder.  src_imgs (n_src,3,S,S); view_src (V,) long: the source of every view; cameras / poses (V,4,4);
        depths (n_src,1,S,S) stands in for the regressor when the model has none.
        -> dict(PredImg (V,3,S,S), FeaturesImg, background_mask, codes, depth)."""
        depth_src = self.regress_depth(src_imgs, depths)   # :303-311
        fs_src = src_imgs if getattr(self.opt, "use_rgb_features", True) else self.encoder(src_imgs)
        planned = self.plan_views(fs_src[view_src].contiguous(), depth_src[view_src].contiguous(), K, K_inv, input_RT, input_RTinv,
                                  output_RT, output_RTinv)
        out = self.outpaint_planned(planned, None, self.opt.temperature if temperature is None else temperature, uniforms)
        if check:
            self.outpaint2.engine(self.obs[1], self.obs[2], K.shape[0]).check()
        pred = (self._decode_checked if check else self._decode_candidate)(out["gen_fs"], out["background_mask"], out["codes"])
        return dict(PredImg=pred, FeaturesImg=out["gen_fs"], background_mask=out["background_mask"], codes=out["codes"],
                    depth=depth_src, plan=out["plan"])

    # ---------------------------------------------------------------- reference-shaped single image path
    @torch.no_grad()
    def forward_image(self, batch, netD=None):
        """Hot-path part of forward_image (z_buffermodel.py:291-419) for model_setting gen_img / gen_paired_img.
        batch: {"images": [(B,3,S,S)], "cameras": [{"P","Pinv","K","Kinv"}], optional "depths": [(B,1,S,S)],
        optional "codes": (B,32,32)} -> (None, outputs dict with the reference's keys)."""
        dev = next(self.parameters()).device   # (the renderer itself refuses anything but the GPU)
        input_img = batch["images"][0].to(dev)
        cam = {k: v.to(dev) for k, v in batch["cameras"][0].items() if torch.is_tensor(v)}
        K, K_inv, input_RT, input_RTinv = cam["K"], cam["Kinv"], cam["P"], cam["Pinv"]
        paired = getattr(self.opt, "model_setting", "gen_img") == "gen_paired_img"
        if paired:   # :294-295 (process_batch :127-130): the target view comes with the batch
            output_img = batch["images"][-1].to(dev)
            output_RT, output_RTinv = batch["cameras"][-1]["P"].to(dev), batch["cameras"][-1]["Pinv"].to(dev)
        else:
            output_RTinv, output_RT = self.get_rt_from_rot(self.opt.direction, input_RT)
        regressed_pts = self.regress_depth(input_img, batch["depths"][0].to(dev) if "depths" in batch else None)   # :303-311
        fs = input_img if getattr(self.opt, "use_rgb_features", True) else self.encoder(input_img)
        gen_fs, background_mask = self.pts_transformer.forward_justpts(fs, regressed_pts, K, K_inv, input_RT,
                                                                      input_RTinv, output_RT, output_RTinv)
        outputs = {"InputImg": input_img, "PredDepthImg": regressed_pts / 5 - 1,
                   "ForegroundImg": (~background_mask).repeat(input_img.shape[0], 1, 1, 1).float(), "FeaturesImg": gen_fs}
        if paired:
            outputs["OutputImg"] = output_img
        if getattr(self.opt, "no_outpainting", False):   # :383-384
            outputs["PredImg"] = self._project_checked(gen_fs, None)
            return None, outputs
        if self.vqvae is not None:
            enc = getattr(self.vqvae, "encode_codes", None)      # our mirror: top codes only, int32, on the device
            downsampled_fs = enc(gen_fs) if enc is not None else self.vqvae.encode(gen_fs)[3]
        else:
            downsampled_fs = batch["codes"].to(dev)
        if max(int(getattr(self.opt, "num_samples", 1)), 1) > 1:   # :349 -> get_best_sample with opt.num_samples candidates
            plan = self.get_masks_for_batch(output_RT, input_RTinv, background_mask, compact=True)
            outputs["PredImg"] = self.get_best_sample(plan, downsampled_fs, background_mask, gen_fs, netD, input_img,
                                                      shard=bool(getattr(self.opt, "shard_samples", False)))
            return None, outputs
        masks_init, masks_undilated, masks_dilated, gen_order = self.get_masks_for_batch(output_RT, input_RTinv,
                                                                                         background_mask)
        autoreg_output, _ = sample(self.outpaint2, gen_order, masks_init, masks_undilated, masks_dilated,
                                   downsampled_fs, self.obs, self.args, 0, self.opt.temperature,
                                   self.downsample(background_mask.float()))
        codes = torch.argmax(autoreg_output, dim=1)
        outputs["PredCodes"] = codes
        if self.vqvae is not None:  # :250-252 (without a refinement net the blend itself is the prediction)
            outputs["PredImg"] = self._decode_checked(gen_fs, background_mask, codes.to(torch.int64))
        return None, outputs

    # ---------------------------------------------------------------- sample ranking (8f.3, host logic)
    def _entropy_score(self, gen_img):
        """Entropy of the scene classifier on the candidate, including the reference's reinterpretation of the
        (3,256,256) tensor as (256,256,3) (z_buffermodel.py:256-262) and its 224x224 ImageNet-normalised input."""
        from PIL import Image
        raw = ((gen_img[0].reshape([256, 256, 3]).cpu().numpy() * .5 + .5) * 255).astype(np.uint8)
        im = np.asarray(Image.fromarray(raw).resize((224, 224), Image.BILINEAR), np.float32) / 255.0
        im = (im - np.array([0.485, 0.456, 0.406], np.float32)) / np.array([0.229, 0.224, 0.225], np.float32)
        x = torch.from_numpy(im).permute(2, 0, 1)[None].to(gen_img.device)
        with torch.no_grad():
            probs = torch.softmax(self.classifier(x).float().cpu(), 1).squeeze().numpy()
        probs = np.sort(probs)[::-1]
        return float(-np.sum(probs * np.log(probs)))

    def _decode_candidate(self, gen_fs, background_mask, codes):
        """codes (B,32,32) -> image: decode, blend with the reprojected features (a14), refine if a projector exists."""
        combined = self.get_combined(gen_fs, self.vqvae.decode_code(codes), background_mask)
        return combined if self.projector is None else self.projector(combined, background_mask)

    def _run_checked(self, device, fn):
        """fn() -- any pass through the refinement decoder -- then (synchronising) the question whether one of ITS split-fp16 convolutions
        met an activation beyond fp16's range (csrc/conv_f16x3.hip raises a device flag; the image is then wrong): if so the pass is run
        again with every convolution through torch in fp32, with a warning.  The flag is cleared BEFORE the pass (what an earlier,
        unchecked pass left there is not this one's) and by the check.  Weights under spectral norm behind normalisation layers do not
        get there; a checkpoint that does should set opt.decoder_conv = "fp32" and save itself the first attempt."""
        clear_f16x3_overflow(device)
        out = fn()
        try:
            check_f16x3_overflow(device)
        except RuntimeError as err:
            import warnings
            warnings.warn(f"{err}: the decoder pass is run again in fp32")
            with decoder_conv("fp32"):
                out = fn()
        return out

    def _decode_checked(self, gen_fs, background_mask, codes):
        """_decode_candidate behind the overflow check of _run_checked: every image that is returned, ranked or fed into the next frame
        of a chain goes through here."""
        return self._run_checked(gen_fs.device, lambda: self._decode_candidate(gen_fs, background_mask, codes))

    def _project_checked(self, gen_fs, *mask):
        """The no_outpainting form: the refinement decoder on the reprojected features alone (z_buffermodel.py:383-384; the chained mode
        calls it without the mask argument), checked likewise."""
        if self.projector is None:
            return gen_fs
        return self._run_checked(gen_fs.device, lambda: self.projector(gen_fs, *mask))

    @torch.no_grad()
    def get_best_sample(self, *args, uniforms=None, shard=False):
        """z_buffermodel.py:244-276 on the fused sampler: num_samples outpaintings of the same view, the best by
        discriminator + entropy rank is kept.  Two call forms:
          get_best_sample(gen_order, masks, downsampled_fs, background_mask, gen_fs, netD, input_img)   the reference's (:244),
              gen_order / masks as get_masks_for_batch returns them;
          get_best_sample(plan, codes, background_mask, gen_fs, netD, input_img)                         with the compact ARPlan.
        `codes` / downsampled_fs (B,32,32): the VQ-VAE codes of gen_fs.  One sample needs no scorers; more need `netD`
        (pixelsynth_amd.losses.DiscriminatorLoss or the reference's) and `self.classifier`.
        uniforms: optional (num_samples,B,L) draws (otherwise torch.Generator seeded i, as sample() reseeds with i).
        shard (or opt.shard_samples through forward_image): under torch.distributed the candidates are dealt over the ranks
        (candidate i on rank i % W: SURVEY 8e), two scalars per candidate are gathered, every rank applies the rank rule and
        the owner of the winner broadcasts it."""
        from . import distributed as D
        if isinstance(args[0], ARPlan):
            plan, codes, background_mask, gen_fs, netD, input_img = args
        else:
            gen_order, masks, codes, background_mask, gen_fs, netD, input_img = args
            plan = plan_from_reference_args(gen_order, masks, self.downsample(background_mask.float()), self.obs[1], gen_fs.device)
        n = max(int(getattr(self.opt, "num_samples", 1)), 1)
        if n > 1 and (netD is None or self.classifier is None):
            raise RuntimeError("num_samples > 1 ranks candidates with the discriminator (netD: pixelsynth_amd.losses.DiscriminatorLoss "
                               "or the reference's) and the scene classifier -- pass netD or use num_samples=1")
        B, G = codes.shape[0], self.obs[1]
        L = G * self.obs[2]
        dev = codes.device
        if uniforms is None:
            uniforms = torch.stack([torch.rand(B, L, generator=torch.Generator(device="cpu").manual_seed(i)) for i in range(n)]).to(dev)
        rank, world = D.world()
        mine = D.shard_views(n, rank, world) if (shard and world > 1 and n > 1) else list(range(n))
        # The candidates are independent AR runs of the same view(s): they go through the sampler TOGETHER, as k * B frames
        # (sample-major) that share the view's order and masks and differ in their draws -- one wavefront schedule, the
        # launches of one run instead of k runs one after the other (the num_samples candidates are one of
        # the path's natural parallel axes).
        per = max(1, min(len(mine), self.sample_batch // max(B, 1)))          # candidates per engine run
        imgs, disc, entr = {}, [], []
        for s0 in range(0, len(mine), per):
            idx = mine[s0:s0 + per]
            k = len(idx)
            rep = lambda t: t.repeat((k,) + (1,) * (t.dim() - 1)).contiguous()
            waves = plan.waves
            if k > 1:
                from .lmconv.model import wavefronts
                waves = wavefronts(np.tile(plan._order_host, (k, 1)), G, self.obs[2], plan.first_step, dev)
            c = rep(codes.reshape(B, L).to(torch.int32))
            eng = self.outpaint2.engine(G, self.obs[2], k * B)
            eng.ar_run(c, rep(plan.order_loc), rep(plan.region), rep(plan.mask_init), rep(plan.mask_undilated),
                       rep(plan.mask_dilated), temperature=self.opt.temperature,
                       uniforms=uniforms[idx].reshape(k * B, L).contiguous(), first_step=plan.first_step, waves=waves)
            eng.check()
            for j, i in enumerate(idx):
                img = self._decode_checked(gen_fs, background_mask, c[j * B:(j + 1) * B].view(B, G, self.obs[2]))
                imgs[i] = img
                if n > 1:
                    disc.append(float(netD.run_discriminator_one_step(img, input_img)["D_Fake"].mean().cpu()))
                    entr.append(self._entropy_score(img))
        if n == 1:
            return imgs[0]
        if len(mine) < n:
            d_all, e_all = D.gather_scores(disc, entr, n)
            best = rank_samples(list(d_all), list(e_all))
            return D.broadcast_from(imgs.get(best), D.owner_of(best, world), gen_fs.device)   # (a rank without candidates
                                                                                               # learns the shape from the owner)
        return imgs[rank_samples(disc, entr)]

    # ---------------------------------------------------------------- chained trajectories (8f.4)
    def _scene_depth(self, img, batch):
        if self.pts_regressor is not None:  # :476-480
            return torch.sigmoid(self.pts_regressor(img)) * (self.opt.max_z - self.opt.min_z) + self.opt.min_z
        fn = batch.get("depth_fn")  # synthetic runs: a callable img -> depth stands in for the Unet
        if fn is None:
            raise RuntimeError("forward_scene regresses depth from every generated frame: give the model a "
                               "pts_regressor or the batch a 'depth_fn' callable")
        return fn(img)

    def _scene_frame(self, st, batch, K, K_inv, in_RT, in_RTinv, out_RT, out_RTinv, netD, input_img):
        """One frame of a chained trajectory: depth of the current frame, cumulative reprojection (only the points
        that were background last time are new, a5), outpainting, state hand-over (:476-522 / :540-582)."""
        depth = self._scene_depth(st.img, batch)
        fs = st.img if getattr(self.opt, "use_rgb_features", True) else self.encoder(st.img)
        gen_fs, background_mask, cloud, feats = self.pts_transformer.forward_justpts_cumulative(
            fs, depth, K, K_inv, in_RT, in_RTinv, out_RT, out_RTinv, st.cloud, st.feats, st.background, st.out_RTinv)
        if not getattr(self.opt, "no_outpainting", False):
            plan = build_ar_plan(background_mask, self.obs[1])
            gen_img = self.get_best_sample(plan, self.vqvae.encode_codes(gen_fs), background_mask, gen_fs, netD, input_img)
        else:
            gen_img = self._project_checked(gen_fs)
        st.img, st.cloud, st.feats, st.background, st.out_RTinv = gen_img, cloud, feats, background_mask, out_RTinv
        return gen_img, gen_fs, depth, background_mask

    @torch.no_grad()
    def forward_scene(self, batch, netD=None):
        """z_buffermodel.py:420-584 (model_setting gen_scene / gen_two_imgs): per direction, first the far end of the
        sweep (unless sequential_outpainting), then the views in between, every frame rendered from the previous
        one on top of the accumulated point cloud.  B = 1, as in the reference (a5 needs equal counts per image).
        -> (None, outputs) with the reference's keys PredImg_<dir>_<i>, FeaturesImg_..., PredDepthImg_..., ForegroundImg_..."""
        dev = next(self.parameters()).device   # (the renderer itself refuses anything but the GPU)
        input_img = batch["images"][0].to(dev)
        cam = {k: v.to(dev) for k, v in batch["cameras"][0].items() if torch.is_tensor(v)}
        K, K_inv, input_RT, input_RTinv = cam["K"], cam["Kinv"], cam["P"], cam["Pinv"]
        two = self.opt.model_setting == 'gen_two_imgs'
        directions = [self.mapping[int(batch["direction"])]] if two else list(self.opt.directions)
        sequential = bool(getattr(self.opt, "sequential_outpainting", False))
        outputs = {"InputImg": input_img}
        st = _SceneState(input_img)

        def pose_of(direction, numerator, denom):
            return self.get_rt_from_rot(direction, input_RT, numerator, denom)

        def summary(direction, tag, gen_fs, depth, background_mask):
            outputs[f"FeaturesImg_{direction}_{tag}"] = gen_fs
            outputs[f"PredDepthImg_{direction}_{tag}"] = depth
            outputs[f"ForegroundImg_{direction}_{tag}"] = (~background_mask).repeat(input_img.shape[0], 1, 1, 1).float()

        for direction in directions:
            base = int(self.opt.num_split)
            if two:
                num_split = 2
            elif direction in ('S', 'C'):
                num_split = base * 2
            elif direction in ('U', 'D', 'UL', 'UR', 'DR', 'DL'):
                num_split = max(base // 2, 1)
            else:
                num_split = base

            def source_pose():  # where the frame we render FROM was taken
                if st.numerator is None:
                    return input_RTinv, input_RT
                return pose_of(st.direction, st.numerator, num_split)

            if not sequential:
                # the large completion first (:470-522)
                in_RTinv, in_RT = source_pose()
                out_RTinv, out_RT = pose_of(direction, num_split, num_split)
                gen_img, gen_fs, depth, bgm = self._scene_frame(st, batch, K, K_inv, in_RT, in_RTinv, out_RT, out_RTinv,
                                                                netD, input_img)
                st.numerator, st.direction = num_split, direction
                outputs[f"PredImg_{direction}_{num_split}"] = gen_img
                summary(direction, num_split, gen_fs, depth, bgm)
                todo = range(num_split - 1, -1, -1)
            else:
                todo = range(num_split + 1)
            for i in todo:
                if sequential and i == 0:
                    in_RTinv, in_RT = source_pose()
                else:
                    in_RTinv, in_RT = pose_of(direction, st.numerator, num_split)
                out_RTinv, out_RT = pose_of(direction, i, num_split)
                gen_img, gen_fs, depth, bgm = self._scene_frame(st, batch, K, K_inv, in_RT, in_RTinv, out_RT, out_RTinv,
                                                                netD, input_img)
                outputs[f"PredImg_{direction}_{i}"] = gen_img
                outputs[f"FeaturesImg_{direction}_{i}"] = gen_fs
                if sequential and i == num_split:
                    summary(direction, num_split, gen_fs, depth, bgm)
                    st.direction = direction
                st.numerator = i
        return None, outputs

    @torch.no_grad()
    def forward_gen_order(self, batch):
        """z_buffermodel.py:594-639 (model_setting 'get_gen_order'): depth, reprojection + splat, generation order --
        -> (None, {"gen_order": (B,L,2) int64 device tensor of (row, col) by rank})."""
        dev = next(self.parameters()).device
        input_img = batch["images"][0].to(dev)
        cam = {k: v.to(dev) for k, v in batch["cameras"][0].items() if torch.is_tensor(v)}
        K, K_inv, input_RT, input_RTinv = cam["K"], cam["Kinv"], cam["P"], cam["Pinv"]
        if len(batch["cameras"]) > 1 and "P" in batch["cameras"][-1]:   # process_batch hands over the target pose (:127-130) ...
            output_RT, output_RTinv = batch["cameras"][-1]["P"].to(dev), batch["cameras"][-1]["Pinv"].to(dev)
        else:                                                           # ... a demo-style batch has the direction instead
            output_RTinv, output_RT = self.get_rt_from_rot(self.opt.direction, input_RT)
        regressed_pts = self.regress_depth(input_img, batch["depths"][0].to(dev) if "depths" in batch else None)   # :606-612
        fs = input_img if getattr(self.opt, "use_rgb_features", True) else self.encoder(input_img)
        _, background_mask = self.pts_transformer.forward_justpts(fs, regressed_pts, K, K_inv, input_RT, input_RTinv,
                                                                  output_RT, output_RTinv)
        plan = self.get_masks_for_batch(output_RT, input_RTinv, background_mask, compact=True)
        return None, {"gen_order": torch.from_numpy(np.stack(plan.gen_order)).to(dev)}

    def forward(self, batch, netD=None):
        """z_buffermodel.py:278-290."""
        if self.opt.model_setting in ('gen_scene', 'gen_two_imgs'):
            return self.forward_scene(batch, netD)
        if self.opt.model_setting == 'get_gen_order':
            return self.forward_gen_order(batch)
        return self.forward_image(batch, netD)
