set -e
mkdir -p gpurun_out/r9
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r9/tests.txt 2>&1 || { tail -30 gpurun_out/r9/tests.txt; exit 1; }
tail -3 gpurun_out/r9/tests.txt
for w in tqc_push_b2048 sac_slide_b512; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-profiler --workload $w --steps 2000 --warmup 200 2>&1 | tail -1 | cut -c1-200
done
bash tools/_r9b.sh > /dev/null 2>&1 || true
