"""Tuning aid: from a rocprofv3 --kernel-trace CSV of a bench run, the main queue's last steps: busy time by kernel, idle gaps.
usage: python tools/step_gaps.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    r["n"] = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("pslm::", "").replace("void ", "").split("(")[0][:30]
rows.sort(key=lambda r: r["s"])
mainq = max(set(r["Queue_Id"] for r in rows), key=lambda q: sum(1 for r in rows if r["Queue_Id"] == q and r["n"].startswith("k_column")))
main = [r for r in rows if r["Queue_Id"] == mainq]
# a step starts with its k_mask_codes (ar_run's first kernel)
starts = [i for i, r in enumerate(main) if r["n"].startswith("k_mask_codes")]
mid = max(1, min(4, len(starts) - 4))   # (the last AR runs of a bench run are the roofline timing runs: no splat beside them)
for a, b in zip(starts[mid:mid + 3], starts[mid + 1:mid + 4]):
    seg = main[a:b]
    t0, t1 = seg[0]["s"], main[b]["s"]
    busy = defaultdict(lambda: [0, 0])
    gaps = 0
    big = []
    for k, r in enumerate(seg):
        busy[r["n"]][0] += 1
        busy[r["n"]][1] += r["e"] - r["s"]
        nxt = seg[k + 1]["s"] if k + 1 < len(seg) else t1
        g = nxt - r["e"]
        if g > 0:
            gaps += g
            if g > 20000:
                big.append((round(g / 1e3, 1), r["n"], (seg[k + 1]["n"] if k + 1 < len(seg) else "next step")))
    tot = sum(v[1] for v in busy.values())
    print(f"step {(t1 - t0) / 1e6:.3f} ms: kernels {tot / 1e6:.3f} ms, gaps {gaps / 1e6:.3f} ms; big gaps (us, after, before): {big[:8]}")
    for n, (c, t) in sorted(busy.items(), key=lambda kv: -kv[1][1])[:8]:
        print(f"    {n:32s} x{c:4d}  {t / 1e6:7.3f} ms  avg {t / c / 1e3:7.1f} us")
    other = [r for r in rows if r["Queue_Id"] != mainq and r["s"] < t1 and r["e"] > t0]
    ob = defaultdict(int)
    for r in other:
        ob[r["n"]] += r["e"] - r["s"]
    print("    other queues in this window: " + "; ".join(f"{n} {t / 1e6:.2f} ms" for n, t in sorted(ob.items(), key=lambda kv: -kv[1])[:6]))
