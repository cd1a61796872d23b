"""How many walked columns of a bench step does nobody need?  (observed locations after a frame's last sampled position)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
V = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda", 0)
model = bench.build_model(dev)
d, _ = bench.make_inputs(0, V, dev)
out = bench.run_step(model, d, 1)
plan = out["plan"]
order = plan._order_host            # (F, L) location by rank
region = plan.region.cpu().numpy() # (F, L) by location
first = plan.first_step
F, L = order.shape
walked = L - first
tail = 0; samp = 0; lastr = []
for f in range(F):
    r = region[f][order[f]]        # by rank
    s = np.nonzero(r)[0]
    samp += len(s)
    last = s.max() if len(s) else first - 1
    lastr.append(int(last))
    tail += L - 1 - last
print(f"V={V} first={first} walked/frame={walked} sampled mean={samp/F:.1f}; columns after the last sampled rank: {tail/F:.1f} per frame ({100*tail/(F*walked):.1f} % of walked)")
print("last sampled rank per frame (first 16):", lastr[:16])
fs = [int(np.nonzero(region[f][order[f]])[0].min()) if region[f].any() else L for f in range(F)]
print("first sampled rank per frame (first 16):", fs[:16])
