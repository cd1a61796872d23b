"""Print the numbers of interest from bench.py's JSON line (a file argument, else stdin)."""
import sys, json
text = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
d = json.loads(text.strip().splitlines()[-1])
r = d["roofline"]
print("value", d["value"], "ms/step", d["ms_per_step"], "frac", round(r["frac"], 4), "launch_us", r.get("avg_launch_us"))
for k, v in d.get("other_single_gpu_configs", {}).items():
    print(k, {a: b for a, b in v.items() if not isinstance(b, (str, dict))})
