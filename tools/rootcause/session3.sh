set -x
mkdir -p gpurun_out/rc3
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "conv_v2 or conv_up2 or conv_v3 or wgrad_v2 or fused_activation" > gpurun_out/rc3/pytest_conv.log 2>&1; tail -3 gpurun_out/rc3/pytest_conv.log
timeout 900 python tools/rootcause/stress_v2.py --launches 5000 --json gpurun_out/rc3/stress_new.json > gpurun_out/rc3/stress_new.log 2>&1; tail -1 gpurun_out/rc3/stress_new.log
ENVS="EG3D_LIBNAME=libeg3d_hip_old.so;EG3D_LIBNAME=libeg3d_hip.so;EG3D_LIBNAME=libeg3d_hip_fix1.so" bash tools/ab_bench.sh > gpurun_out/rc3/ab_bench.log 2>&1; cat gpurun_out/rc3/ab_bench.log
LIBS="libeg3d_hip_old.so libeg3d_hip.so" KERN="conv_v2|conv_up2|s2adj|wgrad_v2" bash tools/ab_kernel_libs.sh > gpurun_out/rc3/ab_kernels.log 2>&1; grep -v amdgpu.ids gpurun_out/rc3/ab_kernels.log | head -40
timeout 600 python tools/rootcause/slp_isa_patch.py run > gpurun_out/rc3/slp_isa_patch.log 2>&1; cat gpurun_out/rc3/slp_isa_patch.log
timeout 3400 python -m pytest tests -m gpu -x -q > gpurun_out/rc3/pytest_gpu.log 2>&1; tail -5 gpurun_out/rc3/pytest_gpu.log
