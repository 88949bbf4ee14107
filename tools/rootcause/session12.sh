set -x
mkdir -p gpurun_out/rc12
timeout 900 python -m pytest -m gpu -x -q tests/test_gpu_lossnets.py tests/test_gpu_loops.py > gpurun_out/rc12/pytest_ln.log 2>&1; tail -6 gpurun_out/rc12/pytest_ln.log | cut -c1-250
ENVS="EG3D_LOSS_NET_PRESPLIT=0;EG3D_LOSS_NET_PRESPLIT=1" BENCH_ARGS="--loss-net vgg16" bash tools/ab_bench.sh 2>&1 | grep "==" | tee gpurun_out/rc12/ab_vgg.log
