set -x
mkdir -p gpurun_out/rc4
timeout 3400 python -m pytest tests -m gpu -x -q > gpurun_out/rc4/pytest_gpu.log 2>&1; tail -8 gpurun_out/rc4/pytest_gpu.log
python bench.py > gpurun_out/rc4/bench_default.json 2> gpurun_out/rc4/bench_default.err; cut -c1-600 gpurun_out/rc4/bench_default.json
