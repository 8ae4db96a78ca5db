cd $GRAFT_REPO_ROOT
bash tools/round_profile.sh r04b > gpurun_out/r04b_round_profile.log 2>&1
tail -12 gpurun_out/r04b_round_profile.log | cut -c1-200
echo "== A/B (same box): row-chain launch with the two-role critic phase | without it | weight-slice launch" > gpurun_out/r04b/ab_ddpg_launch_forms.txt
for r in 1 2 3; do
for e in "X=1" "GCRL_NO_DDPG_KSPLIT=1" "GCRL_ROWTILE=1" "GCRL_ROWTILE=1 GCRL_ROWTILE_SC1=1"; do
echo "round $r $e: $(env $e timeout -k 10 200 python bench.py --no-cpu-baseline --no-profiler --steps 3000 --warmup 300 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*')" >> gpurun_out/r04b/ab_ddpg_launch_forms.txt
done; done
cat gpurun_out/r04b/ab_ddpg_launch_forms.txt
