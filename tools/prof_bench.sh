#!/bin/bash
# rocprofv3 kernel trace of the default bench (short run), summary into gpurun_out/prof_<tag>/
set -e
tag=${1:-x}; shift || true
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-profiler --steps 2000 --warmup 200 "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_$tag -name "*kernel_trace.csv" -delete
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
cut -c1-170 "$f" | head -14
