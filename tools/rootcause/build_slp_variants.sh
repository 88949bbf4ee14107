#!/bin/bash
# SLP-vectoriser probes: the conv_v2-family translation units of HEAD rebuilt WITHOUT -fno-slp-vectorize, plus discriminating variants:
#   slp        SLP on
#   slpwz      SLP on, every s_waitcnt the compiler inserts forced to zero (-mllvm -amdgpu-waitcnt-forcezero): a missing / short wait disappears, an ALU-level fault stays
#   slpnz      SLP on, the forward epilogue's noise term made opaque to the vectoriser (asm "+v" on nzs): the packed {v.x * scl.x, nz * strength} product and the
#              op_sel broadcast of its high half cannot form
set -e
cd "$(dirname "$0")/../../3dgan-inversion_amd"
HIPCC=/opt/rocm/bin/hipcc
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -I../include"
W=build/variants; mkdir -p $W/src_head $W/src_nz
FAM="conv_v2 conv_v2_s2adj conv_v2_up conv_wgrad_v2"
cp csrc/*.h $W/src_head/; for f in $FAM; do cp csrc/$f.hip $W/src_head/; done
sed -i 's#"../../include/eg3d_hip.h"#"eg3d_hip.h"#' $W/src_head/common.h
cp $W/src_head/* $W/src_nz/
python3 - <<PY
p='$W/src_nz/conv_v2_common.h'; s=open(p).read()
old='const float nzs = nz[u] * strength;'
assert old in s
s=s.replace(old,'float nzs = nz[u] * strength; asm volatile("" : "+v"(nzs));')
open(p,'w').write(s)
PY
others() { for o in build/*.o; do b=$(basename $o .o); case " $FAM " in *" $b "*) ;; *) echo $o;; esac; done; }
variant() { local name=$1 src=$2; shift 2; mkdir -p $W/$name
  for f in $FAM; do $HIPCC $BASE -I$src "$@" -c $src/$f.hip -o $W/$name/$f.o & done; wait
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o inv3d_amd/libeg3d_hip_$name.so $(others) $(for f in $FAM; do echo $W/$name/$f.o; done); echo built $name; }
variant slp   $W/src_head
variant slpwz $W/src_head -mllvm -amdgpu-waitcnt-forcezero
variant slpnz $W/src_nz
