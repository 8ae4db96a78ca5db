cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/full2
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/full2/default.txt 2>&1; echo "rc=$?"; tail -6 gpurun_out/full2/default.txt | cut -c1-250
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
