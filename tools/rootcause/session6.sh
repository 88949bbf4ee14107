set -x
mkdir -p gpurun_out/rc6
timeout 1200 python -m pytest -m gpu -x -q tests/test_gpu_ops.py tests/test_gpu_fixtures.py > gpurun_out/rc6/pytest_ops.log 2>&1; tail -3 gpurun_out/rc6/pytest_ops.log
timeout 600 python -m pytest -m gpu -q -s "tests/test_gpu_ops.py::test_sampler_indices_exact" 2>&1 | grep "mismatches away\|passed\|failed" | tee gpurun_out/rc6/sampler.log
EG3D_DETERMINISTIC=1 timeout 600 python -m pytest -m gpu -q -s "tests/test_gpu_ops.py::test_sampler_indices_exact" 2>&1 | grep "mismatches away\|passed\|failed" | tee gpurun_out/rc6/sampler_det.log
ENVS="EG3D_LIBNAME=libeg3d_hip_prev.so;EG3D_LIBNAME=libeg3d_hip.so" bash tools/ab_bench.sh 2>&1 | grep "==" | tee gpurun_out/rc6/ab_bench.log
LIBS="libeg3d_hip_prev.so libeg3d_hip.so" KERN="split_act_lds|upconv_epilogue" bash tools/ab_kernel_libs.sh 2>&1 | grep -v amdgpu.ids | tee gpurun_out/rc6/ab_kernels.log
timeout 600 python -m pytest -m gpu -x -q tests/test_gpu_generator.py tests/test_gpu_loops.py > gpurun_out/rc6/pytest_gen.log 2>&1; tail -3 gpurun_out/rc6/pytest_gen.log
