"""Builds profiles/<tag>_k_column_pmc.json -- the record bench.py's roofline.traffic / mfma_counters come from -- out of what
tools/collect_profiles.sh <tag> left under gpurun_out/<tag>/ (pmc_summary.txt: rocprofv3 --pmc, one pass per group; kernel_stats.csv:
the kernel-trace pass).   usage: python tools/pmc_record.py <tag> [views]"""
import csv
import json
import os
import re
import sys
from collections import defaultdict

tag = sys.argv[1]
views = int(sys.argv[2]) if len(sys.argv) > 2 else 128
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", tag)
pmc = defaultdict(dict)      # kernel -> counter -> (dispatches, mean)
for line in open(os.path.join(root, "pmc_summary.txt")):
    m = re.match(r"\s+(\S.*?)\s+(\w+)\s+dispatches\s+(\d+)\s+mean\s+([\d.]+)\s+total", line)
    if m:
        pmc[m.group(1).replace("pslm::", "")][m.group(2)] = (int(m.group(3)), float(m.group(4)))
avg_us, calls = {}, {}
for r in csv.DictReader(open(os.path.join(root, "kernel_stats.csv"))):
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("pslm::", "")
    avg_us[n.split("(")[0]] = float(r["AverageNs"]) / 1e3
    calls[n.split("(")[0]] = int(r["Calls"])
# the throughput form is two kernels since round 5 (chain tiles of 16 columns: k_column_tp, of 8: k_column_tp8): one row, weighted by dispatches
forms = {k: dict(pmc.pop(k)) for k in ("k_column_tp", "k_column_tp8") if k in pmc}
if forms:
    merged = {}
    for c in set().union(*[set(v) for v in forms.values()]):
        n = sum(v[c][0] for v in forms.values() if c in v)
        merged[c] = (n, sum(v[c][0] * v[c][1] for v in forms.values() if c in v) / n)
    pmc["k_column_tp"] = merged
    n = sum(calls.get(k, 0) for k in forms)
    tp_forms_us = {k: round(avg_us[k], 1) for k in forms if k in avg_us}
    avg_us["k_column_tp"] = sum(avg_us[k] * calls[k] for k in forms if k in avg_us) / max(1, n)
CLK = 2.4e3   # shader cycles per us
SIMDS = 1024


def busy(kernel):   # v_mfma_f32_16x16x4_f32: 32 cycles each
    n = pmc[kernel]["SQ_INSTS_MFMA"][1]
    return round(n * 32 / SIMDS / (avg_us[kernel] * CLK), 4)


def traffic(kernel):   # gfx950: FETCH_SIZE (KB) reports half of the bytes of wide coalesced reads -> doubled; WRITE_SIZE as is
    return int(round((2 * pmc[kernel]["FETCH_SIZE"][1] + pmc[kernel]["WRITE_SIZE"][1]) * 1024))


tp, la = "k_column_tp", "k_column_la"
ntp, nla = pmc[tp]["FETCH_SIZE"][0], pmc[la]["FETCH_SIZE"][0]
rec = {
    "kernel": f"k_column_tp ({ntp} dispatches) + k_column_la ({nla}: wavefronts of up to 256 columns as two latency-form launches)",
    "views": views,
    "workload": "PS_BENCH_PMC_CHILD=1 bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-extra (C5: 8 sources x 16 views = 128 views per GPU; the 7 "
                "pipelined steps and nothing else -- the pipeline's fill and flush are in it: launches with fewer than four batches in flight are "
                "narrower, which is where this record's k_column_la / k_column_tp8 dispatches come from)",
    "dispatches": ntp + nla,
    "FETCH_SIZE_KB_mean": round((pmc[tp]["FETCH_SIZE"][1] * ntp + pmc[la]["FETCH_SIZE"][1] * nla) / (ntp + nla), 3),
    "WRITE_SIZE_KB_mean": round((pmc[tp]["WRITE_SIZE"][1] * ntp + pmc[la]["WRITE_SIZE"][1] * nla) / (ntp + nla), 3),
    "correction": "MI355X_MICROARCH.md / HBM: on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) coalesced streaming reads -> "
                  "doubled; WRITE_SIZE uncalibrated, taken as is; both count the L2's memory-side requests (Infinity-Cache hits included)",
    "traffic_bytes_per_launch": int(round((traffic(tp) * ntp + traffic(la) * nla) / (ntp + nla))),
    "mfma": {
        "SQ_INSTS_MFMA_per_launch": int(round((pmc[tp]["SQ_INSTS_MFMA"][1] * ntp + pmc[la]["SQ_INSTS_MFMA"][1] * nla) / (ntp + nla))),
        "mfma_busy_fraction_of_all_simd_cycles": round((busy(tp) * avg_us[tp] * ntp + busy(la) * avg_us[la] * nla)
                                                       / (avg_us[tp] * ntp + avg_us[la] * nla), 4),
        tp: {"SQ_INSTS_MFMA_per_launch": int(pmc[tp]["SQ_INSTS_MFMA"][1]), "mfma_busy_fraction_of_all_simd_cycles": busy(tp),
             "avg_us_under_trace": round(avg_us[tp], 1)},
        la: {"SQ_INSTS_MFMA_per_launch": int(pmc[la]["SQ_INSTS_MFMA"][1]), "mfma_busy_fraction_of_all_simd_cycles": busy(la),
             "avg_us_under_trace": round(avg_us[la], 1),
             "note": "the latency form's chains are fp32 FMA on the vector ALU; MFMA only in the neighbour role"},
        "note": "v_mfma_f32_16x16x4_f32 only (32 cycles each), all 1024 SIMDs at 2.4 GHz, whole launches (durations of the kernel-trace pass)",
    },
    "per_kernel": {k: {"dispatches": pmc[k]["FETCH_SIZE"][0], "FETCH_SIZE_KB_mean": round(pmc[k]["FETCH_SIZE"][1]),
                       "WRITE_SIZE_KB_mean": round(pmc[k]["WRITE_SIZE"][1]), "traffic_bytes_per_launch": traffic(k)} for k in (tp, la)},
    "k_gemm_wg": {k: {"dispatches": pmc[k]["FETCH_SIZE"][0], "FETCH_SIZE_KB_mean": round(pmc[k]["FETCH_SIZE"][1]),
                      "WRITE_SIZE_KB_mean": round(pmc[k]["WRITE_SIZE"][1]), "SQ_INSTS_MFMA_per_launch": int(pmc[k]["SQ_INSTS_MFMA"][1]),
                      "avg_us_under_trace": round(avg_us[k], 1), "mfma_busy_fraction_of_all_simd_cycles": busy(k)}
                  for k in sorted(pmc) if k.startswith("k_gemm_wg")},
    "kernel_table": None,   # filled below
    "source": f"profiles/{tag}_bench_v{views}_pmc_summary.txt + profiles/{tag}_bench_v{views}_kernel_stats.csv (tools/collect_profiles.sh {tag}; "
              "tools/pmc_record.py); traffic_bytes_per_launch is the average over ALL column launches of an AR run, as bench.py's avg_launch_us is",
}
# where a step's time is: every kernel of the profiled command that takes more than 0.5 % of it.  The command runs 3 pipelined steps
# and the 3 extra AR runs of measure_roofline: the AR kernels (namespace pslm) are dispatched 6 times per "step's worth", the splat
# kernels 3 times -- 9 times since the AR runs of consecutive steps overlap (measure_roofline then times three TWO-batch runs).
try:
    with open(os.path.join(root, "bench_under_trace.json")) as fh:
        AR_RUNS = 9 if json.loads(fh.read().strip().splitlines()[-1])["roofline"].get("ar_runs_overlapped") else 6
except Exception:
    AR_RUNS = 6
if forms:   # the table shows the two throughput kernels apart
    rec["mfma"][tp]["forms_avg_us_under_trace"] = tp_forms_us
    for k, v in forms.items():
        pmc[k] = v
        avg_us[k] = tp_forms_us.get(k, avg_us[tp])
stats = {}
for r in csv.DictReader(open(os.path.join(root, "kernel_stats.csv"))):
    raw = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    stats[raw.replace("pslm::", "").split("(")[0]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, "pslm::" in raw)
total_ms = sum(v[2] for v in stats.values())
table = []
for k, (calls, us, ms, ar) in sorted(stats.items(), key=lambda kv: -kv[1][2]):
    if ms < 0.005 * total_ms:
        continue
    row = {"kernel": k, "launches_per_step": round(calls / (AR_RUNS if ar else 3), 1), "avg_us_under_trace": round(us, 1)}
    c = pmc.get(k, {})
    if "SQ_INSTS_MFMA" in c and c["SQ_INSTS_MFMA"][1] > 0:
        row["mfma_busy"] = busy(k)
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        row["traffic_MB_per_launch"] = round(traffic(k) / 1e6, 1)
    row["ms_per_step"] = round(row["launches_per_step"] * us / 1e3, 3)
    table.append(row)
rec["kernel_table"] = table
out = os.path.join(os.path.dirname(root), "..", "profiles")
for src, dst in (("kernel_stats.csv", f"{tag}_bench_v{views}_kernel_stats.csv"), ("pmc_summary.txt", f"{tag}_bench_v{views}_pmc_summary.txt")):
    with open(os.path.join(root, src)) as fi, open(os.path.join(out, dst), "w") as fo:
        fo.write(fi.read())
with open(os.path.join(out, f"{tag}_k_column_pmc.json"), "w") as fh:
    json.dump(rec, fh, indent=1)
print(json.dumps(rec, indent=1))
