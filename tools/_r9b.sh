set -e
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_tqc9 -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-profiler --workload tqc_push_b2048 --steps 400 --warmup 100 > $GRAFT_REPO_ROOT/gpurun_out/prof_tqc9.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/trace_gaps.py gpurun_out/prof_tqc9 48 adam_kernel > gpurun_out/tqc9_map.txt
find gpurun_out/prof_tqc9 -name "*kernel_trace.csv" -delete
tail -50 gpurun_out/tqc9_map.txt
