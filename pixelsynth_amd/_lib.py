"""ctypes binding of libpixelsynth_hip.so (the C ABI declared in include/pixelsynth_hip.h; the measurement / tuning / debugging
entry points tests, bench.py and tools use are declared in include/pixelsynth_hip_debug.h).

The product path has NO fallback: if the shared library is missing or a call fails, a RuntimeError
is raised.  Nothing here (or anywhere under pixelsynth_amd/) imports oracle/.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PS_HIP_LIB") or os.path.join(_HERE, "libpixelsynth_hip.so")   # (PS_HIP_LIB: tuning builds)
_lib = None

c_void_p, c_int, c_float, c_double, c_size_t = (ctypes.c_void_p, ctypes.c_int, ctypes.c_float,
                                                 ctypes.c_double, ctypes.c_size_t)

_PROTOS = {
    "ps_abi_version": (c_int, []),
    "ps_last_error": (ctypes.c_char_p, []),
    "ps_build_info": (ctypes.c_char_p, []),
    "ps_pixelcnn_launch_kinds": (c_int, []),
    "ps_pixelcnn_launch_kind_name": (ctypes.c_char_p, [c_int]),
    "ps_pixelcnn_launch_counts": (c_int, [c_void_p, c_void_p, c_int]),
    "ps_pixelcnn_profile_begin": (c_int, [c_void_p]),
    "ps_pixelcnn_profile_end": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "ps_project_pts_f32": (c_int, [c_void_p] * 5 + [c_int, c_int, c_void_p, c_void_p]),
    "ps_project_pts_cumulative_f32": (c_int, [c_void_p] * 8 + [c_int] * 4 + [c_void_p] * 3),
    "ps_splat_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_double]),
    "ps_splat_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_double, c_int, c_float, c_int,
                             c_int, c_int] + [c_void_p] * 6 + [c_size_t, c_void_p]),
    "ps_project_splat_f32": (c_int, [c_void_p] * 6 + [c_int, c_int, c_int, c_double, c_int, c_float, c_int, c_int,
                                                      c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "ps_generation_order": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "ps_ar_plan": (c_int, [c_void_p, c_int, c_int, c_int] + [c_void_p] * 6),
    "ps_order_masks_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ps_read_status": (c_int, [c_void_p, c_void_p]),
    "ps_custom_order": (c_int, [c_int, c_int, c_void_p, c_void_p]),
    "ps_kernel_masks_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ps_lmconv_workspace_bytes": (c_size_t, [c_int] * 5),
    "ps_lmconv_forward_f32": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                      c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "ps_pixelcnn_create": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "ps_pixelcnn_destroy": (None, [c_void_p]),
    "ps_pixelcnn_forward_f32": (c_int, [c_void_p] * 5 + [c_int, c_void_p, c_void_p]),
    "ps_pixelcnn_ar_run": (c_int, [c_void_p] * 9 + [c_float, c_int, c_int, c_void_p, c_void_p]),
    "ps_pixelcnn_ar_run_waves": (c_int, [c_void_p] * 9 + [c_float, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "ps_pixelcnn_ar_prefix": (c_int, [c_void_p] * 7 + [c_int, c_int, c_int, c_int, c_void_p]),
    "ps_pixelcnn_ar_prefix_frames": (c_int, [c_void_p] * 7 + [c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "ps_pixelcnn_ar_columns": (c_int, [c_void_p] * 9 + [c_float, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "ps_pixelcnn_set_compute_units": (c_int, [c_void_p, c_int]),
    "ps_stream_create_cu_range": (c_int, [c_int, c_int, c_void_p]),
    "ps_stream_destroy": (c_int, [c_void_p]),
    "ps_pixelcnn_time_ar_run_waves": (c_int, [c_void_p] * 8 + [c_float, c_int, c_int, c_void_p, c_void_p, c_int] + [c_void_p] * 4),
    "ps_pixelcnn_time_ar_run_waves_range": (c_int, [c_void_p] * 8 + [c_float, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int] + [c_void_p] * 4),
    "ps_ar_wavefronts": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "ps_ar_wavefronts_capped": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "ps_ar_wavefronts_frames": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "ps_pixelcnn_status": (c_int, [c_void_p, c_void_p]),
    "ps_pixelcnn_debug_cache": (c_void_p, [c_void_p, c_int, c_int]),
    "ps_pixelcnn_set_tuning": (c_int, [c_void_p, ctypes.c_char_p, c_int]),
    "ps_pixelcnn_get_tuning": (c_int, [c_void_p, ctypes.c_char_p, c_void_p]),
    "ps_zbuffer_scatter_f32": (c_int, [c_void_p] * 4 + [c_int] * 4 + [c_void_p] * 4),
    "ps_zbuffer_project_f32": (c_int, [c_void_p] * 6 + [c_int, c_int] + [c_void_p] * 5),
    "ps_zbuffer_scatter_sorted_f32": (c_int, [c_void_p] * 5 + [c_int] * 4 + [c_void_p] * 4),
    "ps_vq_nearest_f32": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "ps_vq_embed_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ps_affine_relu_nhwc_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ps_pool_add_nhwc_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ps_pool_add_post_nhwc_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ps_upsample_add_nhwc_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ps_add_bias_nhwc_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ps_cat_mask_nhwc_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ps_noise_affine_f32": (c_int, [c_void_p] * 6 + [ctypes.c_float, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "ps_conv3x3_thin_in_nhwc_f32": (c_int, [c_void_p] * 4 + [c_int] * 4 + [c_void_p, c_void_p]),
    "ps_conv3x3_thin_in_f16x3_nhwc": (c_int, [c_void_p] * 4 + [c_int] * 4 + [c_void_p, c_void_p, c_void_p]),
    "ps_conv3x3_thin_out_nhwc_f32": (c_int, [c_void_p] * 4 + [c_int] * 5 + [c_void_p, c_void_p]),
    "ps_vq_stem_s2d_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ps_vq_head_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ps_conv1x1_takes": (c_int, [c_int, c_int]),
    "ps_conv1x1_nhwc_f32": (c_int, [c_void_p, c_void_p, ctypes.c_size_t, c_int, c_int, c_void_p, c_void_p]),
    "ps_conv1x1_ex_nhwc_f32": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, ctypes.c_size_t, c_int, c_int, c_void_p, c_void_p]),
    "ps_conv3x3_f16x3_packed_bytes": (ctypes.c_size_t, [c_int, c_int]),
    "ps_conv3x3_f16x3_pack": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "ps_conv3x3_f16x3_nhwc": (c_int, [c_void_p] * 6 + [c_int] * 5 + [c_void_p, c_void_p, c_void_p]),
    "ps_conv3x3_f16x3_ex_nhwc": (c_int, [c_void_p] * 6 + [c_int] * 8 + [c_void_p, c_void_p, c_void_p]),
    "ps_pixelcnn_time_column_step": (c_int, [c_void_p] * 6 + [c_int, c_int, c_int] + [c_void_p] * 5),
    "ps_pixelcnn_ar_step": (c_int, [c_void_p] * 6 + [c_int, c_int, c_int, c_void_p, c_void_p]),
}


def exported_symbols():
    """Names every entry point include/pixelsynth_hip.h and include/pixelsynth_hip_debug.h declare (used by the CPU load test)."""
    return sorted(_PROTOS)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m pixelsynth_amd.build` "
                "(there is no CPU/PyTorch fallback for the HIP path)")
        # torch must be imported first: it brings its own libamdhip64/libhsa-runtime64, and this library has
        # to bind to THAT runtime instance (same streams, same allocations) instead of loading a second one.
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            if not hasattr(L, name):
                continue  # optional symbols are checked by tests/test_abi.py
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().ps_last_error()
        raise RuntimeError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")


def ptr(t):
    """Device (or host) pointer of a contiguous torch tensor / numpy array, or None."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return ctypes.c_void_p(t.data_ptr())
    return ctypes.c_void_p(t.ctypes.data)


def current_stream():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_STATUS = {}


def status_word(device=None):
    """The caller-owned status word (int32, zero) of a device that asynchronous entry points raise bits in
    (include/pixelsynth_hip.h: PS_STATUS_*); one per device, owned by this binding -- the library keeps none."""
    import torch
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    t = _STATUS.get(idx)
    if t is None:
        t = _STATUS[idx] = torch.zeros(1, dtype=torch.int32, device=torch.device("cuda", idx))
    return t


def read_status(what, device=None):
    """Synchronise the current stream OF THE STATUS WORD'S DEVICE and raise if an asynchronous call raised a bit in it."""
    import torch
    word = status_word(device)
    with torch.cuda.device(word.device):
        check(lib().ps_read_status(ptr(word), ctypes.c_void_p(torch.cuda.current_stream(word.device).cuda_stream)), what)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("pixelsynth_amd HIP path needs tensors on the ROCm device "
                               "(got a CPU tensor; there is no CPU fallback)")
