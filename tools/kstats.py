"""Per-step breakdown of a rocprofv3 kernel_stats.csv: calls per step x average duration, normalised by the
number of calls of a kernel that runs once per step (first argument substring)."""
import csv
import sys

path, once = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(path)))
steps = [int(r["Calls"]) for r in rows if once in r["Name"]][0]
tot = 0.0
for r in rows:
    c = int(r["Calls"])
    us = float(r["TotalDurationNs"]) / steps / 1e3
    tot += us
    print("%6.2f x %7.2f us = %7.2f  %s" % (c / steps, float(r["AverageNs"]) / 1e3, us, r["Name"][:84]))
print("sum %.1f us/step over %d steps" % (tot, steps))
