#!/bin/bash
# Builds liborbit2_hip.so of another git revision into orbit-2_amd/lib/alt/<name>.so for same-box A/B timing:
#   tools/ab_build.sh <git-rev> <name>;  then on the GPU box:  ORBIT2_HIP_LIB=orbit-2_amd/lib/alt/<name>.so python tools/...
set -e
rev=$1; name=$2
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
git -C "$root" archive "$rev" orbit-2_amd/csrc include | tar -x -C "$tmp"
mkdir -p "$root/orbit-2_amd/lib/alt"
objs=""
for f in "$tmp"/orbit-2_amd/csrc/*.hip; do
  o="$tmp/$(basename "$f" .hip).o"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -c "$f" -o "$o" &
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/orbit-2_amd/lib/alt/$name.so" $objs
rm -rf "$tmp"
echo "built orbit-2_amd/lib/alt/$name.so from $rev"
