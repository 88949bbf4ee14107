set -x
mkdir -p gpurun_out/rc16
ENVS="EG3D_LIBNAME=libeg3d_hip.so;EG3D_LIBNAME=libeg3d_hip_prio1.so;EG3D_LIBNAME=libeg3d_hip_prio3.so" bash tools/ab_bench.sh 2>&1 | grep "==" | tee gpurun_out/rc16/ab.log
LIBS="libeg3d_hip.so libeg3d_hip_prio1.so libeg3d_hip_prio3.so" KERN="conv_v2_kernel" bash tools/ab_kernel_libs.sh 2>&1 | grep -v amdgpu.ids | tee gpurun_out/rc16/ab_kernels.log
