set -o pipefail
mkdir -p gpurun_out/final
python -m pytest tests -x -q -m gpu > gpurun_out/final/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 gpurun_out/final/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err; echo "bench rc=$?"; tail -c 1500 gpurun_out/final/bench_default.json
