"""Records per 8x8 tile of the splat for bench.py's inputs (how uneven the per-tile lists of k_composite are)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DEBUG", "False")
import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
model = bench.build_model(dev)
d, _ = bench.make_inputs(1, B, dev)
pm = model.pts_transformer
pts = pm.project_pts(d["depth"], d["K"], d["Kinv"], d["P"], d["Pinv"], d["RT2"], d["RT2inv"])  # (B,3,N) sampler
x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
S, hw = 256, 4.0 * 1.0001 + 0.01
# the splat negates x,y, then pixel (reversed index) c = ((p+1)S-1)/2
cx = ((-x + 1) * S - 1) * 0.5
cy = ((-y + 1) * S - 1) * 0.5
ok = (z >= 0)
cnt = torch.zeros(B, 32, 32, device=dev)
for b in range(B):
    x0 = ((S - 1 - (cx[b] + hw).floor().clamp(0, S - 1)) // 8).long(); x1 = ((S - 1 - (cx[b] - hw).ceil().clamp(0, S - 1)) // 8).long()
    y0 = ((S - 1 - (cy[b] + hw).floor().clamp(0, S - 1)) // 8).long(); y1 = ((S - 1 - (cy[b] - hw).ceil().clamp(0, S - 1)) // 8).long()
    inb = ok[b] & (cx[b] + hw >= 0) & (cx[b] - hw <= S - 1) & (cy[b] + hw >= 0) & (cy[b] - hw <= S - 1)
    for ty in range(3):
        for tx in range(3):
            yy = y0 + ty; xx = x0 + tx
            m = inb & (yy <= y1) & (xx <= x1)
            cnt[b].view(-1).index_add_(0, (yy * 32 + xx)[m], torch.ones(int(m.sum()), device=dev))
c = cnt.flatten()
print("tiles", c.numel(), "mean", c.mean().item(), "max", c.max().item(), "empty", (c == 0).float().mean().item())
for q in (0.5, 0.9, 0.99, 0.999):
    print("quantile", q, torch.quantile(c, q).item())
print("sum over tiles of ceil(n/256) rounds", (c / 256).ceil().sum().item(), " per-frame max tile:", cnt.flatten(1).max(1).values[:8].tolist())
