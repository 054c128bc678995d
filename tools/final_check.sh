mkdir -p gpurun_out/final
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/final/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > gpurun_out/final/smoke.txt
python bench.py 2>&1 | tail -1 > gpurun_out/final/bench.json
cat gpurun_out/final/gpu_suite.txt gpurun_out/final/smoke.txt; cut -c1-400 gpurun_out/final/bench.json
