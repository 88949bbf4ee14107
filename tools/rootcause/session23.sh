set -x
mkdir -p gpurun_out/rc23
ENVS="EG3D_LIBNAME=libeg3d_hip_prev.so;EG3D_LIBNAME=libeg3d_hip.so" BENCH_ARGS="--images-per-gpu 8 --steps 40" bash tools/ab_bench.sh 2>&1 | grep "==" | tee gpurun_out/rc23/ab_n8.log
ENVS="EG3D_LIBNAME=libeg3d_hip_prev.so;EG3D_LIBNAME=libeg3d_hip.so" bash tools/ab_bench.sh 2>&1 | grep "==" | tee gpurun_out/rc23/ab_n1.log
for i in 1 2 3; do timeout 300 python -m pytest -m gpu -x -q tests/test_gpu_lossnets.py -k four_channel 2>&1 | tail -1; done
