set -x
mkdir -p gpurun_out/rc15
timeout 3400 python -m pytest tests -m gpu -x -q > gpurun_out/rc15/pytest_gpu.log 2>&1; tail -4 gpurun_out/rc15/pytest_gpu.log
python bench.py > gpurun_out/rc15/bench_default.json 2> gpurun_out/rc15/bench_default.err; cut -c1-300 gpurun_out/rc15/bench_default.json
