set -x
mkdir -p gpurun_out/rc9
timeout 3400 python -m pytest tests -m gpu -x -q > gpurun_out/rc9/pytest_gpu.log 2>&1; tail -4 gpurun_out/rc9/pytest_gpu.log
timeout 600 python tools/soak_phase_a.py > gpurun_out/rc9/soak_a.log 2>&1; tail -3 gpurun_out/rc9/soak_a.log | cut -c1-300
timeout 900 python tools/stress_phase_b_graph.py > gpurun_out/rc9/soak_b.log 2>&1; tail -3 gpurun_out/rc9/soak_b.log | cut -c1-300
python bench.py > gpurun_out/rc9/bench_default.json 2> gpurun_out/rc9/bench_default.err; cut -c1-400 gpurun_out/rc9/bench_default.json
