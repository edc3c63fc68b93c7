#!/bin/bash
# usage: mkvar.sh name "-DFLAGS"   : builds orbit-2_amd/lib/alt/name.so with attn.hip recompiled under FLAGS (other objects reused)
set -e
name=$1; flags=$2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p /tmp/var_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result $flags -c $R/orbit-2_amd/csrc/attn.hip -o /tmp/var_$name/attn.o
objs=""
for f in $R/orbit-2_amd/build/*.o; do b=$(basename $f); [ "$b" = attn.o ] && continue; objs="$objs $f"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/orbit-2_amd/lib/alt/$name.so /tmp/var_$name/attn.o $objs
echo built $name
