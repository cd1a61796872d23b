"""Generation orders and kernel masks -- drop-in for the parts of the reference's
models/lmconv/masking.py that PixelSynth uses ('custom' order, kernel_masks, get_unfolded_masks,
get_masks).  The integer work runs in the C++ host code of libpixelsynth_hip.so
(csrc/host_order.cpp); plotting helpers and the undefined hilbert/gilbert orders are out of scope."""
import numpy as np
import torch

from .. import _lib


def raster_scan_idx(rows, cols):
    return np.array([(r, c) for r in range(rows) for c in range(cols)])


def s_curve_idx(rows, cols):
    idx = []
    for r in range(rows):
        col_idx = range(cols) if r % 2 == 0 else range(cols - 1, -1, -1)
        idx += [(r, c) for c in col_idx]
    return np.array(idx)


def custom_idx(rows, cols, distances, mass_center=None):
    """get_custom_order.custom_idx (models/lmconv/get_custom_order.pyx:4-124).  `distances` must be an
    int64 numpy array and is multiplied by 10000 in place, exactly like the reference (:26);
    mass_center is unused there too."""
    assert rows == cols
    if not (isinstance(distances, np.ndarray) and distances.dtype == np.int64 and distances.flags.c_contiguous):
        raise TypeError("custom_idx expects a C-contiguous int64 numpy array (it is modified in place)")
    order = np.empty((rows * cols, 2), np.int32)
    _lib.check(_lib.lib().ps_custom_order(rows, cols, _lib.ptr(distances), _lib.ptr(order)), "ps_custom_order")
    return order.astype(np.int64)


def get_generation_order_idx(order, rows, cols, distances=None, mass_center=None):
    """Get (rows*cols) x 2 np array given order that pixels are generated (masking.py:113-119)."""
    assert order in ["raster_scan", "s_curve", "custom"], f"order {order!r} is not available"
    if order == "custom":
        return custom_idx(rows, cols, distances, mass_center)
    return {"raster_scan": raster_scan_idx, "s_curve": s_curve_idx}[order](rows, cols)


def kernel_masks(generation_order_idx, nrows, ncols, k=3, dilation=1, mask_type='B', set_padding=0,
                 observed_idx=None):
    """masking.py:287-341 -> (nrows*ncols, k, k) float64 numpy, row-major location index."""
    if observed_idx is not None or set_padding != 0:
        raise NotImplementedError("observed_idx / set_padding are unused by PixelSynth")
    assert mask_type in ['A', 'B']
    order = np.ascontiguousarray(generation_order_idx, dtype=np.int32)
    m = np.empty((k * k, nrows * ncols), np.float32)
    _lib.check(_lib.lib().ps_kernel_masks_f32(_lib.ptr(order), order.shape[0], nrows, ncols, k, dilation,
                                              int(mask_type == 'B'), _lib.ptr(m)), "ps_kernel_masks_f32")
    return m.T.reshape(nrows * ncols, k, k).astype(np.float64)


def get_unfolded_masks(generation_order_idx, nrows, ncols, k=3, dilation=1, mask_type='B', observed_idx=None):
    """masking.py:343-349 -> (1, k*k, nrows*ncols) float tensor (CPU)."""
    if observed_idx is not None:
        raise NotImplementedError("observed_idx is unused by PixelSynth")
    assert mask_type in ['A', 'B']
    order = np.ascontiguousarray(generation_order_idx, dtype=np.int32)
    m = np.empty((k * k, nrows * ncols), np.float32)
    _lib.check(_lib.lib().ps_kernel_masks_f32(_lib.ptr(order), order.shape[0], nrows, ncols, k, dilation,
                                              int(mask_type == 'B'), _lib.ptr(m)), "ps_kernel_masks_f32")
    return torch.from_numpy(m)[None]


def get_masks(generation_idx, nrows, ncols, k=3, max_dilation=1, observed_idx=None, out_dir="runs",
              plot_suffix="", plot=True):
    """masking.py:351-370: type-A, type-B and dilated type-B masks, moved to the GPU and repeated
    torch.cuda.device_count() times along dim 0 (DataParallel support in the reference)."""
    n = max(1, torch.cuda.device_count())
    mask_init = get_unfolded_masks(generation_idx, nrows, ncols, k=k, dilation=1, mask_type='A')
    mask_undilated = get_unfolded_masks(generation_idx, nrows, ncols, k=k, dilation=1, mask_type='B')
    mask_init = mask_init.cuda(non_blocking=True).repeat(n, 1, 1)
    mask_undilated = mask_undilated.cuda(non_blocking=True).repeat(n, 1, 1)
    if max_dilation == 1:
        mask_dilated = mask_undilated
    else:
        mask_dilated = get_unfolded_masks(generation_idx, nrows, ncols, k=k, dilation=max_dilation, mask_type='B')
        mask_dilated = mask_dilated.cuda(non_blocking=True).repeat(n, 1, 1)
    return mask_init, mask_undilated, mask_dilated
