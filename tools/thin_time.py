"""The decoder's 4 -> 64 layer at 16 x 256 x 256: fp32 FMA kernel against the split-fp16 MFMA kernel (csrc/conv_thin.hip)."""
import sys, torch
sys.path.insert(0, ".")
from pixelsynth_amd import _lib
dev = torch.device("cuda:0"); L = _lib.lib(); st = lambda: torch.cuda.current_stream().cuda_stream
V = int(sys.argv[1]) if len(sys.argv) > 1 else 16
x = torch.randn(V, 256, 256, 4, device=dev); w = torch.randn(3, 3, 4, 64, device=dev) * 0.1
sc = torch.rand(V, 4, device=dev) + 0.5; sh = torch.randn(V, 4, device=dev) * 0.1
y = torch.empty(V, 256, 256, 64, device=dev); flag = torch.zeros(1, dtype=torch.int32, device=dev)
def t(fn):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / 20 * 1e3
a = t(lambda: _lib.check(L.ps_conv3x3_thin_in_nhwc_f32(x.data_ptr(), sc.data_ptr(), sh.data_ptr(), w.data_ptr(), V, 256, 256, 64, y.data_ptr(), st()), "a"))
y1 = y.clone()
b = t(lambda: _lib.check(L.ps_conv3x3_thin_in_f16x3_nhwc(x.data_ptr(), sc.data_ptr(), sh.data_ptr(), w.data_ptr(), V, 256, 256, 64, y.data_ptr(), flag.data_ptr(), st()), "b"))
print(f"thin_in {V} views: fp32 FMA {a:.1f} us, split-fp16 MFMA {b:.1f} us; max |difference| {(y - y1).abs().max().item():.2e} of {y1.abs().max().item():.2f}")
