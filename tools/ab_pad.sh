cd /root/repo
for lib in libeg3d_hip.so libpad_16.so libpad_32.so libpad_48.so libpad_96.so libpad_160.so libpad_224.so; do
  echo "== $lib"
  EG3D_LIBNAME=$lib python tools/conv_launch_table.py 2>/dev/null | awk '$2 != 5 && NR>1' | awk '{s+=$(NF-1)} END {print "igemm total us", s}'
  EG3D_LIBNAME=$lib python tools/conv_launch_table.py 2>/dev/null | sed -n '10,20p;38,50p' > gpurun_out/pad_$lib.txt
  EG3D_LIBNAME=$lib python bench.py --no-side-configs --no-cpu-baseline --no-final-psnr --no-roofline | cut -c90-140
done
