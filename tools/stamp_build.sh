#!/bin/bash
# Diagnostic library with in-kernel cycle stamps (-DO2_STAMP): orbit-2_amd/lib/alt/stamp.so.  Never shipped / loaded by
# default; use with ORBIT2_HIP_LIB=orbit-2_amd/lib/alt/stamp.so python tools/gemm_stamp.py
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
mkdir -p "$root/orbit-2_amd/lib/alt"
objs=""
for f in "$root"/orbit-2_amd/csrc/*.hip; do
  o="$tmp/$(basename "$f" .hip).o"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -DO2_STAMP ${STAMP_DEFS} -c "$f" -o "$o" &
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/orbit-2_amd/lib/alt/${1:-stamp}.so" $objs
rm -rf "$tmp"
echo "built orbit-2_amd/lib/alt/${1:-stamp}.so"
