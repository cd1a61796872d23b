import sys, torch, torch.nn.functional as F
sys.path.insert(0, ".")
from pixelsynth_amd import _lib
dev = torch.device("cuda:0"); L = _lib.lib(); st = lambda: torch.cuda.current_stream().cuda_stream
flag = torch.zeros(1, dtype=torch.int32, device=dev)
torch.manual_seed(1)
for (V, H, Ci, Co) in [(16, 256, 128, 128), (64, 256, 128, 128), (128, 256, 128, 128), (128, 256, 64, 128), (128, 128, 256, 256), (128, 64, 256, 256)]:
    x = torch.randn(V, H, H, Ci, device=dev)
    w = torch.randn(Co, Ci, 3, 3, device=dev) * 0.03
    wl = w.permute(0, 2, 3, 1).contiguous()
    wp = torch.empty(L.ps_conv3x3_f16x3_packed_bytes(Co, Ci), dtype=torch.uint8, device=dev)
    _lib.check(L.ps_conv3x3_f16x3_pack(wl.data_ptr(), Co, Ci, wp.data_ptr(), st()), "pack")
    y = torch.empty(V, H, H, Co, device=dev)
    _lib.check(L.ps_conv3x3_f16x3_nhwc(x.data_ptr(), None, None, wp.data_ptr(), None, None, V, H, H, Ci, Co, y.data_ptr(), flag.data_ptr(), st()), "conv")
    y32 = F.conv2d(x.permute(0, 3, 1, 2), w.contiguous(memory_format=torch.channels_last), None, 1, 1)
    worst16 = worst32 = 0.0
    bad = []
    for v in range(0, V, 8):
        ref = F.conv2d(x[v:v + 8].permute(0, 3, 1, 2).double(), w.double(), None, 1, 1)
        top = ref.abs().max().item()
        e16 = (y[v:v + 8].permute(0, 3, 1, 2).double() - ref).abs().amax(dim=(1, 2, 3)) / top
        e32 = (y32[v:v + 8].double() - ref).abs().amax(dim=(1, 2, 3)) / top
        worst16, worst32 = max(worst16, e16.max().item()), max(worst32, e32.max().item())
        bad += [v + i for i in range(e16.numel()) if e16[i] > 1e-5]
    print(f"V {V} H {H} {Ci}->{Co}: f16x3 {worst16:.3e}  torch fp32 {worst32:.3e}  frames beyond 1e-5: {bad[:20]} ({len(bad)})", flush=True)
