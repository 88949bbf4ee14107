set -x
mkdir -p gpurun_out/rc1
for v in oldkh old oldslp fix1kh fix1 fix1slp; do
  EG3D_LIBNAME=libeg3d_hip_$v.so timeout 900 python tools/rootcause/stress_v2.py --launches 5000 --json gpurun_out/rc1/stress_$v.json > gpurun_out/rc1/stress_$v.log 2>&1
  tail -1 gpurun_out/rc1/stress_$v.log
done
EG3D_LIBNAME=libeg3d_hip_oldslp.so timeout 300 python tools/debug_half.py > gpurun_out/rc1/debug_half_oldslp.log 2>&1
EG3D_LIBNAME=libeg3d_hip_fix1slp.so timeout 300 python tools/debug_half.py > gpurun_out/rc1/debug_half_fix1slp.log 2>&1
grep -h "differing_launches\": [1-9]" gpurun_out/rc1/*.log | cut -c1-300
