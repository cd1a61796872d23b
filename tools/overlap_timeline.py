"""Summarise a rocprofv3 kernel trace CSV of an overlapped bench run: per queue, busy time and kernel mix over the last step."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
t_end = max(r["e"] for r in rows)
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 60e6   # last N ms
sel = [r for r in rows if r["s"] > t_end - win]
t0 = min(r["s"] for r in sel)
byq = defaultdict(list)
for r in sel:
    byq[r["Queue_Id"]].append(r)
print(f"window {(t_end - t0) / 1e6:.2f} ms, {len(sel)} kernels")
for q, rs in byq.items():
    busy = sum(r["e"] - r["s"] for r in rs)
    mix = defaultdict(lambda: [0, 0])
    for r in rs:
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:28]
        mix[n][0] += 1
        mix[n][1] += r["e"] - r["s"]
    top = sorted(mix.items(), key=lambda kv: -kv[1][1])[:5]
    print(f"queue {q}: {len(rs)} kernels, busy {busy / 1e6:.2f} ms; " + "; ".join(f"{k} x{v[0]} avg {v[1] / v[0] / 1e3:.1f} us" for k, v in top))
