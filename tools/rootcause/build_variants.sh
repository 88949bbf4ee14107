#!/bin/bash
# Variant libraries for the conv_v2 root-cause session: the same objects as libeg3d_hip.so except the conv_v2-family translation units,
# which are rebuilt from (a) the first KH spelling (git e820cd6^), (b) HEAD with the SLP vectoriser on, each with and without the one-line protocol fix
# (LDS reads returned before the step barrier: `s_waitcnt vmcnt(N) lgkmcnt(0)`).  Outputs: inv3d_amd/libeg3d_hip_<variant>.so (EG3D_LIBNAME selects one).
set -e
cd "$(dirname "$0")/../../3dgan-inversion_amd"
HIPCC=/opt/rocm/bin/hipcc
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -Icsrc -I../include"
W=build/variants; mkdir -p $W
FAM="conv_v2 conv_v2_s2adj conv_v2_up conv_wgrad_v2"
others() { for o in build/*.o; do b=$(basename $o .o); case " $FAM " in *" $b "*) ;; *) echo $o;; esac; done; }
variant() {   # name, srcdir, extra flags
  local name=$1 src=$2; shift 2
  mkdir -p $W/$name
  for f in $FAM; do $HIPCC $BASE "$@" -c $src/$f.hip -o $W/$name/$f.o & done; wait
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o inv3d_amd/libeg3d_hip_$name.so $(others) $(for f in $FAM; do echo $W/$name/$f.o; done)
  echo built $name
}
# sources: HEAD copy, HEAD copy + the old wait (for variants built AFTER the fix landed in csrc/), first-KH-spelling copy
rm -rf $W/src_*; mkdir -p $W/src_old $W/src_oldkh $W/src_fix $W/src_fixkh
OLD=${OLD_REV:-943c759}
for f in $FAM; do git show $OLD:3dgan-inversion_amd/csrc/$f.hip > $W/src_old/$f.hip; done
for h in common.h det.h render_common.h conv_v2_common.h; do git show $OLD:3dgan-inversion_amd/csrc/$h > $W/src_old/$h; done
cp $W/src_old/* $W/src_oldkh/; git show e820cd6^:3dgan-inversion_amd/csrc/conv_v2.hip > $W/src_oldkh/conv_v2.hip
# the one-line fix applied to the OLD sources: every counted wait also drains this wave's LDS reads
for d in old oldkh; do
  t=$W/src_fix${d#old}; cp $W/src_$d/* $t/
  sed -i 's/asm volatile("s_waitcnt vmcnt(\([0-9]\))" ::: "memory")/asm volatile("s_waitcnt vmcnt(\1) lgkmcnt(0)" ::: "memory")/; s/asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")/asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory")/' $t/conv_v2_common.h $t/conv_v2.hip $t/conv_v2_s2adj.hip $t/conv_v2_up.hip $t/conv_wgrad_v2.hip
done
sed -i 's#"../../include/eg3d_hip.h"#"eg3d_hip.h"#' $W/src_*/common.h
variant old     $W/src_old    -fno-slp-vectorize
variant oldkh   $W/src_oldkh  -fno-slp-vectorize
variant oldslp  $W/src_old
variant fix1    $W/src_fix    -fno-slp-vectorize
variant fix1kh  $W/src_fixkh  -fno-slp-vectorize
variant fix1slp $W/src_fix
ls -la inv3d_amd/*.so
