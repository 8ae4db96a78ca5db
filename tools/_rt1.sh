cd $GRAFT_REPO_ROOT
export GCRL_ROWTILE=1
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "ddpg or DDPG" 2>&1 | tail -5 | cut -c1-200
for e in "X=1" "GCRL_NO_ROWTILE=1" "X=1" "GCRL_NO_ROWTILE=1"; do
echo "$e: $(env $e timeout -k 10 200 python bench.py --no-cpu-baseline --no-profiler --steps 3000 --warmup 300 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*')"
done
GCRL_HIP_LIB=$GRAFT_REPO_ROOT/tools/abl/libgcrl_rtstamps0.so timeout -k 10 120 python tools/rt_stamps.py 2>&1 | tail -100 > gpurun_out/stamps6.txt
