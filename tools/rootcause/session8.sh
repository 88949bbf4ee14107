set -x
mkdir -p gpurun_out/rc8
timeout 900 python -m pytest -m gpu -x -q tests/test_gpu_ops.py -k "up2" > gpurun_out/rc8/pytest_up2.log 2>&1; tail -3 gpurun_out/rc8/pytest_up2.log
timeout 600 python tools/rootcause/stress_v2.py --launches 3000 --cases up2_256x256to128,up2r4_256x256to128,up2r4_128x256to128,up2r4_128x32to256 2>&1 | grep -v amdgpu.ids | cut -c1-220 | tee gpurun_out/rc8/stress_up2.log
ENVS="EG3D_UP2_ROWS4=0;EG3D_UP2_ROWS4=1" bash tools/ab_bench.sh 2>&1 | grep "==" | tee gpurun_out/rc8/ab_n1.log
ENVS="EG3D_UP2_ROWS4=0;EG3D_UP2_ROWS4=1" BENCH_ARGS="--images-per-gpu 8 --steps 40" bash tools/ab_bench.sh 2>&1 | grep "==" | tee gpurun_out/rc8/ab_n8.log
ENVS="EG3D_UP2_ROWS4=0;EG3D_UP2_ROWS4=1" PAT="up2|conv_igemm_kernel<128, 128|Li128ELi128|conv_igemm_kernel<32, 128|split_act" bash tools/ab_step_kernels.sh 2>&1 | grep -v amdgpu.ids | tee gpurun_out/rc8/ab_step.log
timeout 900 python -m pytest -m gpu -x -q tests/test_gpu_generator.py tests/test_gpu_graphed.py > gpurun_out/rc8/pytest_gen.log 2>&1; tail -3 gpurun_out/rc8/pytest_gen.log
