#!/bin/bash
# developer tool: build csrc/libsdtrain_hip_alt.so = the same sources with extra hipcc flags on gemm.hip (e.g. -DCV_READS_FIRST)
# for a same-box A/B against the default build: SDT_LIB=stable_diffusion_training_amd/csrc/libsdtrain_hip_alt.so (OUT=<name>.so for several)
set -e
cd "$(dirname "$0")/../stable_diffusion_training_amd/csrc"
src=${SRC:-gemm}
out=${OUT:-libsdtrain_hip_alt.so}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -Wno-unused-value "$@" -c $src.hip -o /tmp/${src}_$out.o
objs=""
for f in *.hip; do b=${f%.hip}; [ "$b" = "$src" ] && objs="$objs /tmp/${src}_$out.o" || { [ -f $b.o ] && objs="$objs $b.o"; }; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out $objs
ls -la $out
