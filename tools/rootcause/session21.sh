set -x
mkdir -p gpurun_out/rc21
ENVS="EG3D_GATHER_NT=0;EG3D_GATHER_NT=1" BENCH_ARGS="--images-per-gpu 8 --steps 40" bash tools/ab_bench.sh 2>&1 | grep "==" | tee gpurun_out/rc21/ab_n8.log
ENVS="EG3D_GATHER_NT=0;EG3D_GATHER_NT=1" bash tools/ab_bench.sh 2>&1 | grep "==" | tee gpurun_out/rc21/ab_n1.log
timeout 600 python -m pytest -m gpu -x -q tests/test_gpu_ops.py -k "render" 2>&1 | tail -2
