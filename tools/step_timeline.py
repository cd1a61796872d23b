"""Tuning aid: the timeline of ONE steady-state step from a rocprofv3 --kernel-trace CSV of a bench run (PS_BENCH_PMC_CHILD=1: only the
timed steps in it): every queue's kernels between the first column launch of one step and the first column launch of the next, in
start order, runs of one kernel collapsed -- start offset, span, kernel time, idle time inside the run, gap in front of it.
usage: python tools/step_timeline.py <kernel_trace.csv> [step index from the end, default 3]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("pslm::", "").replace("void ", "")
    r["n"] = (n.split("(")[0] if not n.startswith("at::") else "torch:" + n.split("<")[0].split("::")[-1] + ":" + (n.split("Functor")[0].split("::")[-1] if "Functor" in n else ""))[:44]
rows.sort(key=lambda r: r["s"])
queues = sorted(set(r["Queue_Id"] for r in rows))
mainq = max(queues, key=lambda q: sum(1 for r in rows if r["Queue_Id"] == q and r["n"].startswith("k_column")))
main = [r for r in rows if r["Queue_Id"] == mainq]
# a step's column phase starts with the first k_column* launch behind a k_gemm_ws launch on the main queue
firsts = [i for i, r in enumerate(main) if r["n"].startswith("k_column") and i > 0 and not main[i - 1]["n"].startswith("k_column")]
firsts = [i for i in firsts if any(m["n"].startswith("k_gemm") for m in main[max(0, i - 40):i])]
a, b = firsts[-back - 1], firsts[-back]
t0, t1 = main[a]["s"], main[b]["s"]
print(f"step {(t1 - t0) / 1e6:.3f} ms (main queue {mainq}; {len(firsts)} steps in the trace)")
for q in queues:
    seg = [r for r in rows if r["Queue_Id"] == q and r["s"] >= t0 and r["s"] < t1]
    if not seg:
        continue
    print(f"-- queue {q}{' (main)' if q == mainq else ''}: {len(seg)} kernels, {sum(r['e'] - r['s'] for r in seg) / 1e6:.3f} ms of kernel time")
    i = 0
    prev_end = t0
    while i < len(seg):
        j = i
        while j + 1 < len(seg) and seg[j + 1]["n"] == seg[i]["n"]:
            j += 1
        run = seg[i:j + 1]
        ktime = sum(r["e"] - r["s"] for r in run)
        span = run[-1]["e"] - run[0]["s"]
        print(f"   +{(run[0]['s'] - t0) / 1e3:9.1f} us  gap {max(0, run[0]['s'] - prev_end) / 1e3:7.1f}  {run[0]['n']:44s} x{len(run):3d}  kernels {ktime / 1e3:8.1f} us  span {span / 1e3:8.1f} us")
        prev_end = run[-1]["e"]
        i = j + 1
