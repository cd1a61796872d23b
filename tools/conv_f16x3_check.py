"""csrc/conv_f16x3.hip against torch's fp32 convolution (MIOpen) and an fp64 reference, and timed beside it on the decoder's shapes.
usage: python tools/conv_f16x3_check.py [views]"""
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from pixelsynth_amd import _lib

dev = torch.device("cuda:0")
L = _lib.lib()
st = lambda: torch.cuda.current_stream().cuda_stream
flag = torch.zeros(1, dtype=torch.int32, device=dev)


def pack(w):   # (Co, Ci, 3, 3) -> packed bytes
    Co, Ci = w.shape[:2]
    wl = w.permute(0, 2, 3, 1).contiguous()
    out = torch.empty(L.ps_conv3x3_f16x3_packed_bytes(Co, Ci), dtype=torch.uint8, device=dev)
    _lib.check(L.ps_conv3x3_f16x3_pack(wl.data_ptr(), Co, Ci, out.data_ptr(), st()), "pack")
    return out


def conv(x_nhwc, wp, Co, scale=None, shift=None):   # x (B, H, W, Ci) contiguous
    B, H, W, Ci = x_nhwc.shape
    y = torch.empty(B, H, W, Co, dtype=torch.float32, device=dev)
    _lib.check(L.ps_conv3x3_f16x3_nhwc(x_nhwc.data_ptr(), None if scale is None else scale.data_ptr(), None if shift is None else shift.data_ptr(),
                                       wp.data_ptr(), None, None, B, H, W, Ci, Co, y.data_ptr(), flag.data_ptr(), st()), "conv")
    return y


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


torch.manual_seed(0)
# ---- accuracy: small case, fp64 reference
for (B, H, Ci, Co, fuse) in [(2, 32, 64, 128, False), (2, 32, 128, 128, True), (1, 16, 256, 256, True), (3, 48, 32, 128, False), (40, 32, 64, 64, True)]:
    x = torch.randn(B, Ci, H, H, device=dev) * 1.5
    w = torch.randn(Co, Ci, 3, 3, device=dev) * (1.0 / (3 * Ci ** 0.5))
    sc = (torch.rand(B, Ci, device=dev) + 0.5) if fuse else None
    sh = (torch.randn(B, Ci, device=dev) * 0.3) if fuse else None
    xa = torch.clamp_min(x * sc.view(B, Ci, 1, 1) - sh.view(B, Ci, 1, 1), 0) if fuse else x
    ref64 = F.conv2d(xa.double(), w.double(), None, 1, 1)
    y32 = F.conv2d(xa.contiguous(memory_format=torch.channels_last), w.contiguous(memory_format=torch.channels_last), None, 1, 1)
    y = conv(x.permute(0, 2, 3, 1).contiguous(), pack(w), Co, sc, sh).permute(0, 3, 1, 2)
    torch.cuda.synchronize()
    scale = ref64.abs().max().item()
    e16 = (y.double() - ref64).abs().max().item() / scale
    e32 = (y32.double() - ref64).abs().max().item() / scale
    print(f"B {B} H {H} Ci {Ci} Co {Co} fuse {fuse}: max |err| / max |ref|: f16x3 {e16:.3e}, torch fp32 {e32:.3e}; rms f16x3 "
          f"{((y.double() - ref64).pow(2).mean().sqrt() / ref64.pow(2).mean().sqrt()).item():.3e}, torch fp32 "
          f"{((y32.double() - ref64).pow(2).mean().sqrt() / ref64.pow(2).mean().sqrt()).item():.3e}; overflow flag {flag.item()}")
# ---- speed: the decoder's layers at V views
V = int(sys.argv[1]) if len(sys.argv) > 1 else 16
tot16 = tot32 = 0.0
for (H, Ci, Co, n) in [(256, 64, 64, 1), (256, 64, 128, 1), (256, 128, 128, 3), (128, 128, 256, 1), (128, 256, 256, 1), (64, 256, 256, 2), (64, 256, 128, 1),
                       (64, 128, 128, 1), (128, 128, 128, 2)]:
    x = torch.randn(V, H, H, Ci, device=dev)
    w = torch.randn(Co, Ci, 3, 3, device=dev) * 0.05
    wp = pack(w)
    xcl = x.permute(0, 3, 1, 2)   # channels_last view
    wcl = w.contiguous(memory_format=torch.channels_last)
    t16 = timeit(lambda: conv(x, wp, Co))
    t32 = timeit(lambda: F.conv2d(xcl, wcl, None, 1, 1))
    fl = 2 * 9 * Ci * Co * H * H * V
    tot16 += n * t16
    tot32 += n * t32
    print(f"{H:4d}^2 {Ci:3d} -> {Co:3d} x{n}: f16x3 {t16:7.3f} ms = {fl / t16 / 1e9:6.1f} TFLOP/s (fp32-equivalent; {3 * fl / t16 / 1e9:6.0f} on the fp16 pipe),"
          f"  MIOpen fp32 {t32:7.3f} ms = {fl / t32 / 1e9:6.1f} TFLOP/s   x{t32 / t16:.2f}")
print(f"the decoder's wide 3x3 layers at {V} views: f16x3 {tot16:.2f} ms, MIOpen fp32 {tot32:.2f} ms")
