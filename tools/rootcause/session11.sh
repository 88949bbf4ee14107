set -x
mkdir -p gpurun_out/rc11
timeout 900 python -m pytest -m gpu -x -q tests/test_gpu_lossnets.py tests/test_gpu_posenet.py > gpurun_out/rc11/pytest_ln.log 2>&1; tail -15 gpurun_out/rc11/pytest_ln.log | cut -c1-250
ENVS="EG3D_LOSS_NET_PRESPLIT=0;EG3D_LOSS_NET_PRESPLIT=1" BENCH_ARGS="--loss-net vgg16" bash tools/ab_bench.sh 2>&1 | grep "==" | tee gpurun_out/rc11/ab_vgg.log
python tools/time_loss_nets.py 2>&1 | grep -v amdgpu | tail -8 | tee gpurun_out/rc11/time_loss_nets.log
EG3D_LOSS_NET_PRESPLIT=0 python tools/time_loss_nets.py 2>&1 | grep -v amdgpu | tail -8 | tee gpurun_out/rc11/time_loss_nets_off.log
