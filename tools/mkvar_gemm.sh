#!/bin/bash
# usage: mkvar_gemm.sh name "-DFLAGS" : builds orbit-2_amd/lib/alt/name.so with gemm.hip recompiled under FLAGS (other objects reused)
set -e
name=$1; flags=$2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p /tmp/var_$name $R/orbit-2_amd/lib/alt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -I$R/include $flags -c $R/orbit-2_amd/csrc/gemm.hip -o /tmp/var_$name/gemm.o
objs=""
for f in $R/orbit-2_amd/build/*.o; do b=$(basename $f); [ "$b" = gemm.o ] && continue; objs="$objs $f"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/orbit-2_amd/lib/alt/$name.so /tmp/var_$name/gemm.o $objs
echo built $name
