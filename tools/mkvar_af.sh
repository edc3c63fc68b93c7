#!/bin/bash
# usage: mkvar_af.sh name "bar=8,dstride=2" [fwd|dq|dkv] : builds orbit-2_amd/lib/alt/name.so with one generated attention kernel
# (tools/gen_attn_{fwd,dq,dkv}.py) regenerated under that schedule / ablation; every other object comes from orbit-2_amd/build/
set -e
name=$1; cfg=$2; which=${3:-fwd}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
rm -rf /tmp/var_$name; mkdir -p /tmp/var_$name/a/csrc /tmp/var_$name/include $R/orbit-2_amd/lib/alt; cp $R/include/*.h /tmp/var_$name/include/
cp $R/orbit-2_amd/csrc/*.h $R/orbit-2_amd/csrc/attn.hip /tmp/var_$name/a/csrc/
if [ -n "$cfg" ]; then python3 $R/tools/gen_attn_$which.py --cfg "$cfg" --out /tmp/var_$name/a/csrc/attn_${which}_asm.h; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -I$R/include -c /tmp/var_$name/a/csrc/attn.hip -o /tmp/var_$name/attn.o
objs=""
for f in $R/orbit-2_amd/build/*.o; do b=$(basename $f); [ "$b" = attn.o ] && continue; objs="$objs $f"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/orbit-2_amd/lib/alt/$name.so /tmp/var_$name/attn.o $objs
echo built $name
