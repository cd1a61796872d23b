"""OurPixelCNN on MI355X -- drop-in for the reference's models/lmconv/model.py (same constructor, same
parameter names/shapes so reference checkpoints load, same forward signature).

With PixelSynth's configuration (models/z_buffermodel.py:62-74: nr_resnet=2, nr_filters=80,
input_channels=512, 3x3, max_dilation=2, weight_norm=False, PONO norms, dropout 0, no mask weight)
and one-hot (or all-zero) input, forward() runs entirely inside the HIP engine
(ps_pixelcnn_forward_f32: csrc/lmconv.hip).  Any other configuration / input runs layer by layer
like the reference, every masked conv on the HIP lmconv kernel.
"""
import ctypes

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.utils import weight_norm as wn

from .. import _lib
from .layers import PONO, gated_resnet, identity, nin
from .locally_masked_convolution import compact_mask, locally_masked_conv2d
from .utils import concat_elu

DOWN_NR = (2, 3, 3)


def _param_keys():
    keys = []
    for i in range(3):
        for j in range(DOWN_NR[i]):
            p = f"down_layers.{i}.u_stream.{j}."
            keys += [p + "conv_input.weight", p + "conv_input.bias", p + "nin_skip.lin_a.bias",
                     p + "nin_skip.lin_a.weight_g", p + "nin_skip.lin_a.weight_v", p + "conv_out.weight",
                     p + "conv_out.bias"]
    for i in range(3):
        for j in range(2):
            p = f"up_layers.{i}.u_stream.{j}."
            keys += [p + "conv_input.weight", p + "conv_input.bias", p + "conv_out.weight", p + "conv_out.bias"]
    keys += ["u_init.weight", "u_init.bias"]
    for name in ("downsize_u_stream", "upsize_u_stream"):
        for i in range(2):
            keys += [f"{name}.{i}.weight", f"{name}.{i}.bias"]
    keys += ["nin_out.lin_a.bias", "nin_out.lin_a.weight_g", "nin_out.lin_a.weight_v"]
    return keys


PARAM_KEYS = _param_keys()  # the reference state_dict order; ps_pixelcnn_create expects exactly this order
assert len(PARAM_KEYS) == 93


class PixelCNNEngine:
    """Owns a ps_pixelcnn handle (device weights re-packed for MFMA + activation caches)."""

    def __init__(self, state_dict, H=32, W=32, max_frames=1):
        arrs = [np.ascontiguousarray(state_dict[k].detach().cpu().numpy() if hasattr(state_dict[k], "detach")
                                     else state_dict[k], dtype=np.float32) for k in PARAM_KEYS]
        self._keep = arrs
        ptrs = (ctypes.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        self.handle = ctypes.c_void_p()
        self.H, self.W, self.L, self.max_frames = H, W, H * W, max_frames
        rc = _lib.lib().ps_pixelcnn_create(ptrs, len(arrs), H, W, max_frames, ctypes.byref(self.handle))
        _lib.check(rc, "ps_pixelcnn_create")

    def check(self):
        """Synchronise and raise if any column launch of this engine gave up on an in-launch wait (ps_pixelcnn_status)."""
        _lib.check(_lib.lib().ps_pixelcnn_status(self.handle, _lib.current_stream()), "ps_pixelcnn_status")

    def set_tuning(self, **values):
        """Tuning values of the handle by name (include/pixelsynth_hip_debug.h: ps_pixelcnn_set_tuning) -- which launch form the
        whole-grid pass takes from which size on, the look-ahead depths of the column launches, ...  None of them changes
        results; the parity tests use this to run every form inside one process."""
        for k, v in values.items():
            _lib.check(_lib.lib().ps_pixelcnn_set_tuning(self.handle, k.encode(), int(v)), f"ps_pixelcnn_set_tuning({k})")
        return self

    def get_tuning(self, key):
        v = ctypes.c_int(0)
        _lib.check(_lib.lib().ps_pixelcnn_get_tuning(self.handle, key.encode(), ctypes.cast(ctypes.byref(v), ctypes.c_void_p)),
                   f"ps_pixelcnn_get_tuning({key})")
        return v.value

    # ---- which kernels carried the matrix work (include/pixelsynth_hip_debug.h; tests and bench.py, not the product path)
    @staticmethod
    def launch_kind_names():
        L = _lib.lib()
        return [L.ps_pixelcnn_launch_kind_name(k).decode() for k in range(L.ps_pixelcnn_launch_kinds())]

    def launch_counts(self):
        """{kernel: launches of this engine since its creation} (host counters)."""
        names = self.launch_kind_names()
        buf = (ctypes.c_longlong * len(names))()
        _lib.check(_lib.lib().ps_pixelcnn_launch_counts(self.handle, ctypes.cast(buf, ctypes.c_void_p), len(names)), "ps_pixelcnn_launch_counts")
        return dict(zip(names, (int(v) for v in buf)))

    def profile_begin(self):
        _lib.check(_lib.lib().ps_pixelcnn_profile_begin(self.handle), "ps_pixelcnn_profile_begin")

    def profile_end(self):
        """Synchronises -> {kernel: (launches, summed ms)} of the launches since profile_begin (HIP events on the launches' own streams)."""
        names = self.launch_kind_names()
        n, ms = (ctypes.c_int * len(names))(), (ctypes.c_float * len(names))()
        _lib.check(_lib.lib().ps_pixelcnn_profile_end(self.handle, len(names), ctypes.cast(n, ctypes.c_void_p), ctypes.cast(ms, ctypes.c_void_p)),
                   "ps_pixelcnn_profile_end")
        return {k: (int(a), float(b)) for k, a, b in zip(names, n, ms)}

    def close(self):
        if getattr(self, "handle", None):
            _lib.lib().ps_pixelcnn_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _frame_masks(self, F_, *masks):
        """The kernels index the three masks per frame, (F,9,L) f32 contiguous: a (1,9,L) mask -- the reference's broadcast
        form, what get_masks() returns -- is expanded here; anything else is refused before a kernel reads past it."""
        out = []
        for m in masks:
            if m.dim() != 3 or m.shape[1] != 9 or m.shape[2] != self.L or m.shape[0] not in (1, F_):
                raise ValueError(f"mask of shape {tuple(m.shape)}: expected ({F_}|1, 9, {self.L})")
            if m.shape[0] != F_:
                m = m.expand(F_, -1, -1)
            out.append(m.to(torch.float32).contiguous())
        return out

    def _frame_arg(self, F_, name, t, dtype):
        if t is not None and (tuple(t.shape) != (F_, self.L) or t.dtype != dtype or not t.is_contiguous()):
            raise ValueError(f"{name}: expected a contiguous ({F_}, {self.L}) {dtype} tensor, got {tuple(t.shape)} {t.dtype}")

    # masks: (F,9,L) or (1,9,L) f32 device tensors
    def forward(self, codes, mask_init, mask_undilated, mask_dilated):
        """codes (F,L) or (F,H,W) int32 (-1 = zero input) -> logits (F,512,H,W)."""
        F_ = codes.shape[0]
        if F_ > self.max_frames:
            raise ValueError(f"{F_} frames on an engine built for {self.max_frames}")
        codes = codes.reshape(F_, self.L).to(torch.int32).contiguous()
        _lib.require_cuda(codes, mask_init, mask_undilated, mask_dilated)
        mask_init, mask_undilated, mask_dilated = self._frame_masks(F_, mask_init, mask_undilated, mask_dilated)
        logits = torch.empty(F_, 512, self.H, self.W, dtype=torch.float32, device=codes.device)
        rc = _lib.lib().ps_pixelcnn_forward_f32(self.handle, _lib.ptr(codes), _lib.ptr(mask_init),
                                                _lib.ptr(mask_undilated), _lib.ptr(mask_dilated), F_,
                                                _lib.ptr(logits), _lib.current_stream())
        _lib.check(rc, "ps_pixelcnn_forward_f32")
        return logits

    def ar_run(self, codes, order, region, mask_init, mask_undilated, mask_dilated, temperature=1.0, forced=None,
               uniforms=None, first_step=0, want_logits=False, waves=None):
        """In-place AR completion of codes (F,L) int32.  order (F,L) int32 location per order position,
        region (F,L) uint8 by location.  waves: optional wavefront schedule for these orders and this first_step,
        (cols device int32 (n,2), wave_start host int32 array) as from wavefronts() -- same results, far fewer
        dependent launches.  Returns out_logits (F,L,512) or None."""
        F_ = codes.shape[0]
        _lib.require_cuda(codes, order, region, mask_init, mask_undilated, mask_dilated, forced, uniforms)
        mask_init, mask_undilated, mask_dilated = self._frame_masks(F_, mask_init, mask_undilated, mask_dilated)
        for name, t, dt in (("codes", codes, torch.int32), ("order", order, torch.int32), ("region", region, torch.uint8),
                            ("forced", forced, torch.int32), ("uniforms", uniforms, torch.float32)):
            self._frame_arg(F_, name, t, dt)
        out = torch.empty(F_, self.L, 512, dtype=torch.float32, device=codes.device) if want_logits else None
        head = (self.handle, _lib.ptr(codes), _lib.ptr(order), _lib.ptr(region), _lib.ptr(mask_init),
                _lib.ptr(mask_undilated), _lib.ptr(mask_dilated), _lib.ptr(forced), _lib.ptr(uniforms),
                float(temperature), F_, int(first_step))
        from ..distributed import shared_device_turn
        with shared_device_turn():   # (a no-op but in the single-GPU dry run of several ranks)
            if waves is None or waves[0].shape[0] == 0:  # (nothing to walk: only the whole-grid pass runs)
                rc = _lib.lib().ps_pixelcnn_ar_run(*head, _lib.ptr(out), _lib.current_stream())
                _lib.check(rc, "ps_pixelcnn_ar_run")
            else:
                cols, wave_start = waves
                _lib.require_cuda(cols)
                assert cols.dtype == torch.int32 and wave_start.dtype == np.int32
                rc = _lib.lib().ps_pixelcnn_ar_run_waves(*head, _lib.ptr(cols), _lib.ptr(wave_start), len(wave_start) - 1,
                                                         _lib.ptr(out), _lib.current_stream())
                _lib.check(rc, "ps_pixelcnn_ar_run_waves")
        return out

    def _ar_args(self, codes, order, region, mask_init, mask_undilated, mask_dilated):
        F_ = codes.shape[0]
        _lib.require_cuda(codes, order, region, mask_init, mask_undilated, mask_dilated)
        masks = self._frame_masks(F_, mask_init, mask_undilated, mask_dilated)
        for name, t, dt in (("codes", codes, torch.int32), ("order", order, torch.int32), ("region", region, torch.uint8)):
            self._frame_arg(F_, name, t, dt)
        return F_, masks

    def ar_prefix(self, codes, order, region, mask_init, mask_undilated, mask_dilated, first_step, frame_begin=0, frame_end=None,
                  first_steps=None, max_first_step=None):
        """First half of ar_run for frames [frame_begin, frame_end): sampled codes masked out, whole-grid pass over their observed
        prefix.  Asynchronous on the current stream; disjoint frame ranges may go to different streams (ps_pixelcnn_ar_prefix).
        first_steps (F,) int32 device tensor + max_first_step: PER-FRAME prefixes (ps_pixelcnn_ar_prefix_frames) -- frame f's pass
        covers its positions [0, first_steps[f]), first_step <= first_steps[f] <= max_first_step."""
        F_, (mi, mu, md) = self._ar_args(codes, order, region, mask_init, mask_undilated, mask_dilated)
        if first_steps is not None:
            _lib.require_cuda(first_steps)
            assert first_steps.dtype == torch.int32 and first_steps.shape == (F_,) and first_steps.is_contiguous()
            rc = _lib.lib().ps_pixelcnn_ar_prefix_frames(self.handle, _lib.ptr(codes), _lib.ptr(order), _lib.ptr(region), _lib.ptr(mi), _lib.ptr(mu),
                                                         _lib.ptr(md), F_, _lib.ptr(first_steps), int(first_step), int(max_first_step), int(frame_begin),
                                                         int(F_ if frame_end is None else frame_end), _lib.current_stream())
            _lib.check(rc, "ps_pixelcnn_ar_prefix_frames")
            return
        rc = _lib.lib().ps_pixelcnn_ar_prefix(self.handle, _lib.ptr(codes), _lib.ptr(order), _lib.ptr(region), _lib.ptr(mi), _lib.ptr(mu),
                                              _lib.ptr(md), F_, int(first_step), int(frame_begin), int(F_ if frame_end is None else frame_end),
                                              _lib.current_stream())
        _lib.check(rc, "ps_pixelcnn_ar_prefix")

    def ar_columns(self, codes, order, region, mask_init, mask_undilated, mask_dilated, waves, temperature=1.0, forced=None, uniforms=None,
                   first_step=0):
        """Second half of ar_run: the column launches of all frames, wavefront by wavefront (ps_pixelcnn_ar_columns); every frame's
        ar_prefix must have completed (stream order / events are the caller's)."""
        F_, (mi, mu, md) = self._ar_args(codes, order, region, mask_init, mask_undilated, mask_dilated)
        _lib.require_cuda(forced, uniforms)
        self._frame_arg(F_, "forced", forced, torch.int32)
        self._frame_arg(F_, "uniforms", uniforms, torch.float32)
        cols, wave_start = waves
        _lib.require_cuda(cols)
        from ..distributed import shared_device_turn
        with shared_device_turn():
            rc = _lib.lib().ps_pixelcnn_ar_columns(self.handle, _lib.ptr(codes), _lib.ptr(order), _lib.ptr(region), _lib.ptr(mi), _lib.ptr(mu),
                                                   _lib.ptr(md), _lib.ptr(forced), _lib.ptr(uniforms), float(temperature), F_, int(first_step),
                                                   _lib.ptr(cols), _lib.ptr(wave_start), len(wave_start) - 1, _lib.current_stream())
            _lib.check(rc, "ps_pixelcnn_ar_columns")

    def set_compute_units(self, n_cus):
        """Compute units the stream of this engine's column launches can use (0 = the whole device)."""
        _lib.check(_lib.lib().ps_pixelcnn_set_compute_units(self.handle, int(n_cus)), "ps_pixelcnn_set_compute_units")

    def ar_step(self, codes, order, mask_init, mask_undilated, mask_dilated, step, first_step):
        F_ = codes.shape[0]
        mask_init, mask_undilated, mask_dilated = self._frame_masks(F_, mask_init, mask_undilated, mask_dilated)
        self._frame_arg(F_, "codes", codes, torch.int32)
        self._frame_arg(F_, "order", order, torch.int32)
        logits = torch.empty(F_, 512, dtype=torch.float32, device=codes.device)
        rc = _lib.lib().ps_pixelcnn_ar_step(self.handle, _lib.ptr(codes), _lib.ptr(order), _lib.ptr(mask_init),
                                            _lib.ptr(mask_undilated), _lib.ptr(mask_dilated), F_, int(step),
                                            int(first_step), _lib.ptr(logits), _lib.current_stream())
        _lib.check(rc, "ps_pixelcnn_ar_step")
        return logits


# Partitions of the 256 compute units this module has been run on.  Others are refused: with 176, 208 or 216 compute units the
# column launches never got all their workgroups resident (bench runs had to be killed), with 224 they ran at a third of their
# speed -- how the dispatcher deals workgroups over a masked queue's compute units is not something to guess at.
VALIDATED_SPLITS = (128, 160, 192)


class CuRangeStream:
    """A torch stream whose kernels run on compute units [first, first + n) only."""

    def __init__(self, first, n, device=None):
        if (int(first), int(n)) not in [(0, k) for k in VALIDATED_SPLITS] + [(k, 256 - k) for k in VALIDATED_SPLITS]:
            raise ValueError(f"compute units [{first}, {first + n}): only the splits of 256 at {VALIDATED_SPLITS} have been validated")
        self.first, self.n = int(first), int(n)
        self._raw = ctypes.c_void_p()
        with torch.cuda.device(device if device is not None else torch.cuda.current_device()):
            _lib.check(_lib.lib().ps_stream_create_cu_range(self.first, self.n, ctypes.byref(self._raw)), "ps_stream_create_cu_range")
            self.stream = torch.cuda.ExternalStream(self._raw.value)

    def close(self):
        """Drain the stream.  It is NOT destroyed: torch's caching allocator may still hold events recorded on it (record_stream),
        and querying them after hipStreamDestroy crashed at interpreter exit; the runtime reclaims the stream with the process."""
        if getattr(self, "_raw", None) and self._raw.value:
            self.stream.synchronize()


def _resnet_stack(n, nr_filters, nonlinearity, conv_op, feature_norm_op, skip, dropout_prob):
    return nn.ModuleList([gated_resnet(nr_filters, conv_op, feature_norm_op, nonlinearity, skip_connection=skip,
                                       dropout_prob=dropout_prob) for _ in range(n)])


class OurPixelCNNLayer_up(nn.Module):
    """nr_resnet gated blocks in sequence; returns every intermediate u (the skip sources of the down pass)."""

    def __init__(self, nr_resnet, nr_filters, resnet_nonlinearity, conv_op, feature_norm_op=None,
                 kernel_size=(5, 5), weight_norm=True, dropout_prob=0.5, rematerialize=False):
        super(OurPixelCNNLayer_up, self).__init__()
        self.nr_resnet = nr_resnet
        self.u_stream = _resnet_stack(nr_resnet, nr_filters, resnet_nonlinearity, conv_op, feature_norm_op, 0, dropout_prob)

    def forward(self, u, mask=None):
        outs = []
        for block in self.u_stream:
            u = block(u, mask=mask)
            outs.append(u)
        return outs


class OurPixelCNNLayer_down(nn.Module):
    """nr_resnet gated blocks, each taking the most recent unused u of the up pass as its skip input."""

    def __init__(self, nr_resnet, nr_filters, resnet_nonlinearity, conv_op, feature_norm_op=None,
                 kernel_size=(5, 5), weight_norm=True, dropout_prob=0.5, rematerialize=False):
        super(OurPixelCNNLayer_down, self).__init__()
        self.nr_resnet = nr_resnet
        self.u_stream = _resnet_stack(nr_resnet, nr_filters, resnet_nonlinearity, conv_op, feature_norm_op, 1, dropout_prob)

    def forward(self, u, u_list, mask=None):
        for block in self.u_stream:
            u = block(u, a=u_list.pop(), mask=mask)
        return u


class OurPixelCNN(nn.Module):
    def __init__(self, nr_resnet=5, nr_filters=80, nr_logistic_mix=10, resnet_nonlinearity='concat_elu',
                 input_channels=3, kernel_size=(5, 5), max_dilation=2, weight_norm=True, feature_norm_op=None,
                 dropout_prob=0.5, conv_bias=True, conv_mask_weight=False, rematerialize=False, binarize=False):
        super(OurPixelCNN, self).__init__()
        assert resnet_nonlinearity == 'concat_elu'
        self.resnet_nonlinearity = lambda x: concat_elu(x)
        self.init_padding = None
        self.binarize = binarize
        mk = lambda cin, cout, dil=1: locally_masked_conv2d(cin, cout, kernel_size=kernel_size, dilation=dil,
                                                            bias=conv_bias, mask_weight=conv_mask_weight)
        if weight_norm:
            conv_op_init = lambda cin, cout: wn(mk(cin, cout))
            conv_op_dilated = lambda cin, cout: wn(mk(cin, cout, max_dilation))
            conv_op = lambda cin, cout: wn(mk(cin, cout))
        else:
            conv_op_init = lambda cin, cout: mk(cin, cout)
            conv_op_dilated = lambda cin, cout: mk(cin, cout, max_dilation)
            conv_op = lambda cin, cout: mk(cin, cout)
        down_nr_resnet = [nr_resnet] + [nr_resnet + 1] * 2
        self.down_layers = nn.ModuleList([OurPixelCNNLayer_down(down_nr_resnet[i], nr_filters,
                                                                self.resnet_nonlinearity, conv_op, feature_norm_op,
                                                                kernel_size=kernel_size, weight_norm=weight_norm,
                                                                dropout_prob=dropout_prob) for i in range(3)])
        self.up_layers = nn.ModuleList([OurPixelCNNLayer_up(nr_resnet, nr_filters, self.resnet_nonlinearity, conv_op,
                                                            feature_norm_op, kernel_size=kernel_size,
                                                            weight_norm=weight_norm, dropout_prob=dropout_prob)
                                        for _ in range(3)])
        self.u_init = conv_op_init(input_channels + 1, nr_filters)
        self.downsize_u_stream = nn.ModuleList([conv_op_dilated(nr_filters, nr_filters) for _ in range(2)])
        self.upsize_u_stream = nn.ModuleList([conv_op_dilated(nr_filters, nr_filters) for _ in range(2)])
        self.norm_init = feature_norm_op(nr_filters) if feature_norm_op else identity
        self.norm_ds = nn.ModuleList([feature_norm_op(nr_filters) for _ in range(2)]) if feature_norm_op else None
        self.norm_us = nn.ModuleList([feature_norm_op(nr_filters) for _ in range(2)]) if feature_norm_op else None
        if self.binarize:
            self.nin_out = nin(nr_filters, 2, weight_norm=True)
        else:
            self.nin_out = nin(nr_filters, 512, weight_norm=True)
        self._engine_ok = (nr_resnet == 2 and nr_filters == 80 and input_channels == 512 and
                           tuple(kernel_size) == (3, 3) and max_dilation == 2 and not weight_norm and
                           conv_bias and not conv_mask_weight and not binarize and dropout_prob == 0 and
                           feature_norm_op is not None and isinstance(feature_norm_op(nr_filters), PONO))
        self._engine = None
        self._engine_sig = None

    # ---- HIP engine -------------------------------------------------------------------------
    def engine(self, H=32, W=32, max_frames=1, slot=0):
        """The ps_pixelcnn handle for the current parameters (rebuilt when they change).  slot: callers that overlap the runs of
        two batches keep two handles -- each owns its activation caches."""
        if not self._engine_ok:
            raise RuntimeError("the fused HIP PixelCNN engine implements PixelSynth's OurPixelCNN configuration only")
        sd = self.state_dict()
        sig = (H, W, tuple((sd[k].data_ptr(), sd[k]._version) for k in PARAM_KEYS))
        if slot == 0:
            if self._engine is None or self._engine_sig != sig or self._engine.max_frames < max_frames:
                if self._engine is not None:
                    self._engine.close()
                self._engine = PixelCNNEngine(sd, H, W, max(max_frames, getattr(self._engine, "max_frames", 1)))
                self._engine_sig = sig
            return self._engine
        extra = self.__dict__.setdefault("_engine_slots", {})
        eng, esig = extra.get(slot, (None, None))
        if eng is None or esig != sig or eng.max_frames < max_frames:
            if eng is not None:
                eng.close()
            eng = PixelCNNEngine(sd, H, W, max(max_frames, getattr(eng, "max_frames", 1)))
            extra[slot] = (eng, sig)
        return eng

    @staticmethod
    def onehot_to_codes(x):
        """(B,C,H,W) one-hot / all-zero float input -> (codes (B,H,W) int32 with -1 for zero, is_valid bool tensor)."""
        mx, am = x.max(dim=1)
        mn = x.min(dim=1)[0]
        sm = x.sum(dim=1)
        onehot = (mx == 1) & (sm == 1) & (mn == 0)
        zero = (mx == 0) & (mn == 0)
        codes = torch.where(onehot, am, torch.full_like(am, -1)).to(torch.int32)
        return codes, (onehot | zero).all()

    def forward(self, x, sample=False, mask_init=None, mask_undilated=None, mask_dilated=None):
        if isinstance(x, list):
            mask_init, mask_undilated, mask_dilated = x[1], x[2], x[3]
            x = x[0]
        B, C, H, W = x.shape
        if self._engine_ok and x.is_cuda:
            codes, ok = self.onehot_to_codes(x)
            if bool(ok):  # one device->host sync; callers that hold codes use engine() directly
                eng = self.engine(H, W, B)
                return eng.forward(codes, compact_mask(mask_init, B, C + 1), compact_mask(mask_undilated, B, 160),
                                   compact_mask(mask_dilated, B, 80))
        return self._forward_layers(x, sample, mask_init, mask_undilated, mask_dilated)

    def _forward_layers(self, x, sample, mask_init, mask_undilated, mask_dilated):
        """Layer-by-layer forward with the reference's dataflow (model.py:118-155); every lmconv is the HIP kernel.
        Up pass: u0 = norm(u_init([x, 1])), three groups of gated blocks separated by dilated convs, every u kept;
        down pass: the same in reverse, each gated block consuming the latest unused u as its skip input."""
        ones = x.new_ones(x.size(0), 1, x.size(2), x.size(3))
        stack = [self.norm_init(self.u_init(torch.cat((x, ones), 1), mask=mask_init), mask=mask_undilated)]
        for g in range(3):
            stack.extend(self.up_layers[g](stack[-1], mask=mask_undilated))
            if g < 2:
                d = self.downsize_u_stream[g](stack[-1], mask=mask_dilated)
                stack.append(self.norm_ds[g](d, mask=mask_dilated) if self.norm_ds else d)
        u = stack.pop()
        for g in range(3):
            u = self.down_layers[g](u, stack, mask=mask_undilated)
            if g < 2:
                u = self.upsize_u_stream[g](u, mask=mask_dilated)
                if self.norm_us:
                    u = self.norm_us[g](u, mask=mask_dilated)
        return self.nin_out(F.elu(u))


import os

# Columns one launch takes (csrc/lmconv.hip): k_column, the latency form -- one CU per column -- takes COL_CAP = 128; waves with
# more columns run as k_column_tp, the throughput form -- 16-column MFMA chain tiles -- which takes TP_COL_CAP = 1024.  Few
# frames never fill more than the latency form holds, so their schedule is capped at its capacity.  The crossover, measured
# on the bench's sweeps (ms per step, latency form / throughput form): 24 views 7.8 / 11.0, 48: 14.4 / 15.8, 56: 16.5 / 16.6,
# 64: 19.0 / 17.5, 96: 28.2 / 20.6, 128: 38 / 24.4.
COLUMNS_PER_LAUNCH = int(os.environ.get("PS_WAVE_COLS", "128"))
COLUMNS_PER_LAUNCH_TP = int(os.environ.get("PS_WAVE_COLS_TP", "1024"))
# frames from which a batch's column launches take the throughput form.  60 until the launches were packed (round 6: up to four batches share a
# launch, so 24 views already fill it: 56 views 9.6 -> 6.9 ms per step, 40 views 7.6 -> 6.2, 24 views 5.16 -> 5.08; 16 views 3.54 -> 3.70, so not lower)
TP_MIN_FRAMES = int(os.environ.get("PS_TP_MIN_FRAMES", "24"))


import threading

_COLS_STAGE = {}            # page-locked staging for the schedule upload, one buffer per power-of-two capacity, the two largest kept
_COLS_LOCK = threading.Lock()


def launch_capacity(frames):
    return COLUMNS_PER_LAUNCH_TP if frames >= TP_MIN_FRAMES else COLUMNS_PER_LAUNCH


def wavefronts(order_host, H, W, first_step, device=None, max_cols=None, keep_host=False, first_steps=None):
    """Wavefront schedule of an AR run (ps_ar_wavefronts_capped): order_host (F,L) int32 numpy array ->
    (cols int32 (n,2) tensor on `device`, wave_start int32 numpy array of n_waves + 1 entries).
    max_cols: columns per wave (0 = the pure dependency levels; None = what a launch takes for this many frames).
    keep_host: a third value, the (n,2) columns as a numpy array of the caller's own (schedule surgery on the host).
    first_steps: (F,) int32 numpy array, a first walked position PER FRAME (ps_ar_wavefronts_frames; first_step = their minimum)."""
    import ctypes
    order_host = np.ascontiguousarray(order_host, np.int32)
    F_, L = order_host.shape
    if max_cols is None:
        max_cols = launch_capacity(F_)
    with _COLS_LOCK:     # the staging buffer is shared state: one schedule is staged at a time
        return _wavefronts(order_host, F_, L, H, W, first_step, device, max_cols, keep_host, first_steps)


def pack_launches(batches, cap, until_oldest_done=True, budget=None):
    """Launch after launch out of the batches in flight (round 6; z_buffermodel.outpaint_pipelined): `batches`, OLDEST FIRST, are dicts
    with 'ws' (a schedule's wave boundaries, numpy), 'w' (the wave the batch stands at) and 'off' (columns of that wave already
    launched) -- advanced in place.  A launch takes, from the oldest batch on, what is left of each batch's CURRENT wave while there is
    room under `cap` (the columns of a wave are independent: a wave may be dealt to several launches); a batch moves on to its next wave
    only in the launch AFTER the one that took the last of the current one, so every batch's waves keep their order and every
    dependency is met -- and a launch is as full as the batches in flight can make it (the older scheme cut every schedule into
    `depth` parts and merged part p of the batch p calls ago: 44-50 launches per step where 33 hold the columns).
    Stops when the batch that was oldest at the start is complete (until_oldest_done), after `budget` launches, or when nothing is left.
    -> (slices, starts): slices = [(index into `batches`, begin, end)] column ranges in launch order, starts = launch boundaries."""
    slices, starts, total, n = [], [0], 0, 0
    live = [k for k, b in enumerate(batches) if b["w"] < len(b["ws"]) - 1]
    oldest = live[0] if live else None
    while live and (budget is None or n < budget) and not (until_oldest_done and budget is None and oldest not in live):
        room, adv = cap, []
        for k in live:
            b = batches[k]
            if room == 0:
                break          # (the younger batches wait)
            a0 = int(b["ws"][b["w"]]) + b["off"]
            take = min(room, int(b["ws"][b["w"] + 1]) - a0)
            if take > 0:
                slices.append((k, a0, a0 + take))
                total += take
                room -= take
                b["off"] += take
            if a0 + take == int(b["ws"][b["w"] + 1]):
                adv.append(k)
        for k in adv:
            batches[k]["w"] += 1
            batches[k]["off"] = 0
        live = [k for k in live if batches[k]["w"] < len(batches[k]["ws"]) - 1]
        starts.append(total)
        n += 1
    return slices, np.asarray(starts, np.int32)


def _wavefronts(order_host, F_, L, H, W, first_step, device, max_cols, keep_host=False, first_steps=None):
    import ctypes
    nsteps = L - first_step
    n = F_ * nsteps
    if first_steps is not None:
        first_steps = np.ascontiguousarray(first_steps, np.int32)
        assert first_steps.shape == (F_,) and int(first_steps.min()) >= first_step
        n = int((L - first_steps.astype(np.int64)).sum())
    cap = 1 << max(10, int(max(n, 1) - 1).bit_length())          # page-locked staging, kept per power-of-two capacity
    stage = _COLS_STAGE.get(cap) if device is not None else None
    if device is not None and stage is None:
        stage = _COLS_STAGE[cap] = torch.empty((cap, 2), dtype=torch.int32, pin_memory=torch.cuda.is_available())
        for old in sorted(_COLS_STAGE)[:-2]:                     # (bounded: callers that vary their batch do not pile up pinned memory)
            if old != cap:
                del _COLS_STAGE[old]
    cols = stage.numpy() if stage is not None else np.empty((max(n, 1), 2), np.int32)
    wave_start = np.zeros(nsteps + (n + max_cols - 1) // max_cols + 2 if max_cols else nsteps + 1, np.int32)
    nw = ctypes.c_int32(0)
    if first_steps is not None:
        rc = _lib.lib().ps_ar_wavefronts_frames(_lib.ptr(order_host), F_, H, W, _lib.ptr(first_steps), int(max_cols), _lib.ptr(cols),
                                                _lib.ptr(wave_start), ctypes.cast(ctypes.byref(nw), ctypes.c_void_p))
    else:
        rc = _lib.lib().ps_ar_wavefronts_capped(_lib.ptr(order_host), F_, H, W, int(first_step), int(max_cols), _lib.ptr(cols),
                                                _lib.ptr(wave_start), ctypes.cast(ctypes.byref(nw), ctypes.c_void_p))
    _lib.check(rc, "ps_ar_wavefronts_capped")
    host = cols[:n].copy() if keep_host else None
    if device is not None:
        cols_t = stage[:n].to(device, non_blocking=True)
        if cols_t.is_cuda:
            torch.cuda.current_stream().synchronize()             # the staging buffer is free again
    else:
        cols_t = torch.from_numpy(cols[:n].copy() if n else cols[:0].copy())
    ws = np.ascontiguousarray(wave_start[:nw.value + 1])
    return (cols_t, ws, host) if keep_host else (cols_t, ws)
