cd $GRAFT_REPO_ROOT
for e in "GCRL_DW_SLEEP=1" "GCRL_DW_SLEEP=4" "GCRL_DW_SLEEP=16" "GCRL_NO_DW_INLINE=1"; do
echo "$e: $(env $e timeout -k 10 100 python bench.py --no-cpu-baseline --no-profiler --steps 3000 --warmup 300 2>&1 | tail -1 | cut -c1-200 | grep -o '"ms_per_step": [0-9.]*\|Error.*')"
done
bash tools/prof_bench.sh dwinline 2>&1 | grep -E "rowchain|adam_pair|gemm_batch" | cut -c1-150
