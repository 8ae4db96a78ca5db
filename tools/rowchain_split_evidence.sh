#!/bin/bash
# the two-CU k-split of a row-chain pass, measured (tools/microbench_rowchain_split.hip).  hipEvent
# timings of back-to-back launches for three batch sizes, then rocprofv3 kernel durations of the B = 256 case.
set -e
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-rowchain_split}
mkdir -p $out
cd $GRAFT_REPO_ROOT
for B in 16 256 512; do
  echo "== B=$B H=256 NL=10" >> $out/microbench.txt
  timeout -k 10 120 ./tools/microbench_rowchain_split $B 256 10 >> $out/microbench.txt
done
cat $out/microbench.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o p -- $GRAFT_REPO_ROOT/tools/microbench_rowchain_split 256 256 10 > $out/prof.log 2>&1
cd $GRAFT_REPO_ROOT
find $out/prof -name "*kernel_trace.csv" -delete
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
cut -c1-200 "$f" | head -8 | tee $out/kernel_stats_head.txt
