set -x
mkdir -p gpurun_out/rc14
timeout 3400 python -m pytest tests -m gpu -x -q > gpurun_out/rc14/pytest_gpu.log 2>&1; tail -4 gpurun_out/rc14/pytest_gpu.log
bash tools/profile_session.sh > gpurun_out/rc14/profile_session.log 2>&1; tail -3 gpurun_out/rc14/profile_session.log | cut -c1-200
cp gpurun_out/sess/bench.json gpurun_out/rc14/bench_default.json
