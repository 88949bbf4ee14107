#!/bin/bash
# A/B library for kernel changes not yet committed: libeg3d_hip_prev.so = the current objects, except the listed translation units, which are rebuilt from the
# sources (and headers) of git revision REV (default HEAD).   usage: REV=HEAD tools/ab_build_prev.sh conv_v2 epilogue   -> EG3D_LIBNAME=libeg3d_hip_prev.so
set -e
cd "$(dirname "$0")/../3dgan-inversion_amd"
REV=${REV:-HEAD}; W=build/prev; rm -rf $W; mkdir -p $W
for h in common.h det.h render_common.h conv_v2_common.h; do git show $REV:3dgan-inversion_amd/csrc/$h > $W/$h; done
sed -i 's#"../../include/eg3d_hip.h"#"eg3d_hip.h"#' $W/common.h
git show $REV:include/eg3d_hip.h > $W/eg3d_hip.h
for f in "$@"; do git show $REV:3dgan-inversion_amd/csrc/$f.hip > $W/$f.hip; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -I$W -c $W/$f.hip -o $W/$f.o & done; wait
OBJS=$(for o in build/*.o; do b=$(basename $o .o); case " $* " in *" $b "*) echo $W/$b.o;; *) echo $o;; esac; done)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o inv3d_amd/libeg3d_hip_prev.so $OBJS
ls -la inv3d_amd/libeg3d_hip_prev.so
