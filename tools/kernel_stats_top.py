"""The first N rows of a rocprofv3 kernel_stats.csv (find-mode reference kernels left out) with their share and total time.
usage: python tools/kernel_stats_top.py <kernel_stats.csv> <N>"""
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows=[r for r in rows if 'naive' not in r['Name']]
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(sys.argv[2])]:
    print("%-100s %5s %9.1f us avg  %5.1f%%  %8.2f ms" % (r["Name"][:100], r["Calls"], float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot, float(r["TotalDurationNs"])/1e6))
print("total ms", tot/1e6)
