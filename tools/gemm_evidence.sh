#!/bin/bash
# Evidence for the LDS-tiled GEMM at TQC's launch sizes, everything into gpurun_out/<tag>/ (copy what is to be judged into profiles/):
# event-clock rates (tools/gemm_micro), the profiler's kernel durations of the same launches (rocprofv3 --kernel-trace --stats),
# the residency probe and the MFMA issue ceiling.
set -e
tag=${1:-gemm_ev}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/$tag
mkdir -p $out
{
  echo "== tools/occupancy_probe"; $root/tools/occupancy_probe
  echo "== tools/gemm_micro (HIP-event clock, 30 back-to-back launches)"
  $root/tools/gemm_micro 10240 512 512 fwd; $root/tools/gemm_micro 10240 512 512 dx; $root/tools/gemm_micro 512 512 2048 dw 10 1; $root/tools/gemm_micro 512 512 2048 dw 10 2; $root/tools/gemm_micro 256 256 2048 dw 4 8
  $root/tools/gemm_micro 40960 512 512 fwd; $root/tools/gemm_micro 2048 512 512 fwd
} > $out/gemm_micro.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for mode in "10240 512 512 fwd" "10240 512 512 dx" "512 512 2048 dw 10 1" "512 512 2048 dw 10 2" "256 256 2048 dw 4 8"; do
  n=$(echo $mode | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$n -o p -- $root/tools/gemm_micro $mode > /dev/null 2>&1
  f=$(find $out/prof_$n -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$mode" >> $out/gemm_micro.txt <<'PY'
import csv, sys
mode = sys.argv[2].split()
M, N, K = int(mode[0]), int(mode[1]), int(mode[2])
flop = 2.0 * M * N * K if mode[3] != "dw" else 2.0 * int(mode[4]) * M * (N + 1) * K
for r in csv.DictReader(open(sys.argv[1])):
    if "gemm_tiled" in r["Name"]:
        us = float(r["AverageNs"]) / 1e3
        print("rocprofv3 %-22s gemm_tiled_kernel: %s calls, avg %.2f us (min %.2f) -> %.1f TFLOP/s by the profiler's clock" % (" ".join(mode), r["Calls"], us, float(r["MinNs"]) / 1e3, flop / us / 1e6))
PY
  find $out/prof_$n -name "*kernel_trace.csv" -delete
done
cat $out/gemm_micro.txt
