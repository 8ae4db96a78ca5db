#!/bin/bash
# tools/gather_micro under rocprofv3 (kernel trace + stats) for several launch sizes; per-size summaries into gpurun_out/<tag>/
set -e
tag=${1:-gm}; shift || true
sizes=${*:-"10240 77824 81920 327680 1000000"}
thrash=${THRASH_MB:-0}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for n in $sizes; do
  $root/tools/gather_micro $n 40 $thrash > $out/event_$n.txt
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$n -o p -- $root/tools/gather_micro $n 40 $thrash > /dev/null 2>&1
  f=$(find $out/prof_$n -name "*kernel_stats.csv" | head -1)
  python3 - "$f" $n > $out/rocprof_$n.txt <<'PY'
import csv, sys
n = int(sys.argv[2])
for r in csv.DictReader(open(sys.argv[1])):
    us = float(r["AverageNs"]) / 1e3
    print("%-70s calls %4s avg %7.2f us min %7.2f us  alg %5.2f TB/s = %.3f of 8" % (r["Name"][:70], r["Calls"], us, float(r["MinNs"]) / 1e3, 416.0 * n / us / 1e6, 416.0 * n / us / 1e6 / 8))
PY
  echo "== rows $n"; cat $out/rocprof_$n.txt
  find $out/prof_$n -name "*kernel_trace.csv" -delete
done
