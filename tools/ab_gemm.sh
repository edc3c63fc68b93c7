#!/bin/bash
# same-box A/B of GEMM variants: tools/ab_gemm.sh <alt-lib-name>  (runs tools/dw_ab.py with both builds, twice)
R=$GRAFT_REPO_ROOT
for tag in base alt base alt; do
  lib=""; [ "$tag" = alt ] && lib="$R/orbit-2_amd/lib/alt/$1.so"
  echo "== $tag"; ORBIT2_HIP_LIB=$lib python3 $R/tools/${AB_SCRIPT:-dw_ab.py} $AB_ARGS
done
