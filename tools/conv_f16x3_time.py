"""Time csrc/conv_f16x3.hip alone on some of the decoder's shapes (tuning builds: PS_HIP_LIB).  usage: python tools/conv_f16x3_time.py [views]"""
import sys
import torch
sys.path.insert(0, ".")
from pixelsynth_amd import _lib
dev = torch.device("cuda:0")
L = _lib.lib()
st = lambda: torch.cuda.current_stream().cuda_stream
flag = torch.zeros(1, dtype=torch.int32, device=dev)
V = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 16
out = []
for (H, Ci, Co) in [(256, 64, 128), (256, 128, 128), (128, 256, 256), (64, 256, 256)]:
    x = torch.randn(V, H, H, Ci, device=dev)
    w = torch.randn(Co, 3, 3, Ci, device=dev) * 0.05
    wp = torch.empty(L.ps_conv3x3_f16x3_packed_bytes(Co, Ci), dtype=torch.uint8, device=dev)
    _lib.check(L.ps_conv3x3_f16x3_pack(w.data_ptr(), Co, Ci, wp.data_ptr(), st()), "pack")
    y = torch.empty(V, H, H, Co, device=dev)
    res = torch.randn(V, H, H, Co, device=dev) if "--res" in sys.argv else None
    fn = lambda: _lib.check(L.ps_conv3x3_f16x3_nhwc(x.data_ptr(), None, None, wp.data_ptr(), None, None if res is None else res.data_ptr(), V, H, H, Ci, Co,
                                                    y.data_ptr(), flag.data_ptr(), st()), "conv")
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    clk = ""
    if "--clock" in sys.argv:   # a -DPS_CONV_EXP=16 build: the shader clock and the wave-time of the launches just timed
        import ctypes
        raw = ctypes.CDLL(_lib.LIB_PATH)
        buf = (ctypes.c_ulonglong * 2)()
        raw.ps_conv_debug_clock(buf)          # (13 launches: 3 warm-up + 10 timed; every workgroup's first wave)
        clk = f", clock {buf[0] / (buf[1] * 10.0):.3f} GHz"
    out.append(f"{H}^2 {Ci}->{Co}: {t:.3f} ms ({3 * 2 * 9 * Ci * Co * H * H * V / t / 1e9:.0f} TF fp16{clk})")
print(" | ".join(out))
