set -x
mkdir -p gpurun_out/rc2
# 1. the new library (explicit step boundary): stress, both builds
timeout 900 python tools/rootcause/stress_v2.py --launches 5000 --json gpurun_out/rc2/stress_new.json > gpurun_out/rc2/stress_new.log 2>&1; tail -1 gpurun_out/rc2/stress_new.log
# 2. A/B: step time old vs new library
ENVS="EG3D_LIBNAME=libeg3d_hip_old.so;EG3D_LIBNAME=libeg3d_hip.so;EG3D_LIBNAME=libeg3d_hip_fix1.so" bash tools/ab_bench.sh > gpurun_out/rc2/ab_bench.log 2>&1; cat gpurun_out/rc2/ab_bench.log
# 3. per-kernel A/B
LIBS="libeg3d_hip_old.so libeg3d_hip.so" KERN="conv_v2|conv_up2|s2adj|wgrad_v2" bash tools/ab_kernel_libs.sh > gpurun_out/rc2/ab_kernels.log 2>&1; tail -60 gpurun_out/rc2/ab_kernels.log
# 5. SLP probes
for v in slp slpwz slpnz; do
  EG3D_LIBNAME=libeg3d_hip_$v.so timeout 300 python tools/rootcause/slp_probe.py > gpurun_out/rc2/slp_probe_$v.log 2>&1; cat gpurun_out/rc2/slp_probe_$v.log | cut -c1-200
  EG3D_LIBNAME=libeg3d_hip_$v.so timeout 900 python tools/rootcause/stress_v2.py --launches 2000 --json gpurun_out/rc2/stress_$v.json > gpurun_out/rc2/stress_$v.log 2>&1; tail -1 gpurun_out/rc2/stress_$v.log
done
# 4. full GPU suite (last: longest)
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/rc2/pytest_gpu.log 2>&1; tail -5 gpurun_out/rc2/pytest_gpu.log
