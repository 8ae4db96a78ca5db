"""Idle gaps between consecutive kernels of the last `n` kernels of a rocprofv3 kernel trace (the bench's timed region);
an optional third argument names a kernel: the window then ends at its last launch."""
import csv
import glob
import sys

d, n = sys.argv[1], int(sys.argv[2])
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
if len(sys.argv) > 3:
    last = max(i for i, r in enumerate(rows) if sys.argv[3] in r["Kernel_Name"])
    rows = rows[:last + 1]
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = None
busy = 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) if prev_end is not None else 0
    name = r["Kernel_Name"].replace("gcrl::", "").replace("(anonymous namespace)::", "").replace("void ", "")[:34]
    print("%9.1f us  gap %7.1f  dur %6.1f  %s" % ((s - t0) / 1e3, gap / 1e3, (e - s) / 1e3, name))
    prev_end = max(prev_end or 0, e)
    busy += e - s
print("span %.1f us, kernel time %.1f us" % ((prev_end - t0) / 1e3, busy / 1e3))
