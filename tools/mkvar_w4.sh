#!/bin/bash
# usage: mkvar_w4.sh name "lstride=5,bar2=88" : builds orbit-2_amd/lib/alt/name.so with the 4-wave loop regenerated under that schedule
set -e
name=$1; cfg=$2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
rm -rf /tmp/var_$name; mkdir -p /tmp/var_$name/a/csrc /tmp/var_$name/include $R/orbit-2_amd/lib/alt; cp $R/include/*.h /tmp/var_$name/include/
cp $R/orbit-2_amd/csrc/*.h $R/orbit-2_amd/csrc/gemm.hip /tmp/var_$name/a/csrc/
python3 $R/tools/gen_gemm_w4.py --cfg "$cfg" --out /tmp/var_$name/a/csrc/gemm_w4_asm.h
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -I$R/include -c /tmp/var_$name/a/csrc/gemm.hip -o /tmp/var_$name/gemm.o
objs=""
for f in $R/orbit-2_amd/build/*.o; do b=$(basename $f); [ "$b" = gemm.o ] && continue; objs="$objs $f"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/orbit-2_amd/lib/alt/$name.so /tmp/var_$name/gemm.o $objs
echo built $name
