cd $GRAFT_REPO_ROOT
for r in 1 2; do
for e in "X=1" "GCRL_DDPG_KSPLIT_PER_CU=2"; do
echo "cfg2 $e: $(env $e timeout -k 10 100 python bench.py --workload ddpg_reach_b1024 --no-cpu-baseline --no-profiler --steps 2000 --warmup 200 2>&1 | tail -1 | cut -c1-200 | grep -o '"ms_per_step": [0-9.]*\|Error.*')"
done; done
