"""Tuning aid: print the per-stage shader-clock stamps PS_CHAIN_TRACE=<file> makes ps_pixelcnn_time_column_step dump."""
import sys
import numpy as np

a = np.loadtxt(sys.argv[1], dtype=np.int64, comments='#')
# file columns: s, slot0, slot5, slot6, slot7, slot8, slot1, slot2, slot3, slot4
slot = {0: 1, 5: 2, 6: 3, 7: 4, 8: 5, 1: 6, 2: 7, 3: 8, 4: 9}
g = lambda k: a[:, slot[k]]
print("stage | post wave: wait_chains  operands+sum  post  barrier2 | chain wave0: chain  barrier1  barrier2 | stage total")
tot = np.zeros(8, dtype=np.int64)
for k in range(len(a)):
    row = [g(1)[k] - g(0)[k], g(2)[k] - g(1)[k], g(3)[k] - g(2)[k], g(4)[k] - g(3)[k],
           g(6)[k] - g(5)[k], g(7)[k] - g(6)[k], g(8)[k] - g(7)[k], g(4)[k] - g(0)[k]]
    tot += np.array(row)
    print(k, *row)
print("sum", *tot)

for line in open(sys.argv[1]):
    if line.startswith("# marks"):
        v = [int(x) for x in line.split(":")[1].split()]
        print("post wave of workgroup 0, cycles: u0", v[1] - v[0], "| stage loop", v[2] - v[1], "| nin_out + draw", v[3] - v[2])
