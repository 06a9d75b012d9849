#!/bin/bash
# Build a diagnostic copy of the library with wave begin / end stamps in k_knn2_u8_direct (HERE, before gpurun: the .so travels)
#   bash tools/exp_matcher_lifetimes.sh build
# and run it on the GPU box:
#   gpurun -- 'bash tools/exp_matcher_lifetimes.sh run'
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/sfm_amd/lib/libsfm_amd_stamps.so
if [ "${1:-run}" = build ]; then
  python3 -m sfm_amd.build > /dev/null
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form -I$R/include ${EXTRA_FLAGS:-} -DSFM_MATCH_STAMPS=1 \
    -c $R/sfm_amd/csrc/match.hip -o /tmp/match_stamps.o
  objs=$(ls $R/sfm_amd/lib/obj/*.o | grep -v /match.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $objs /tmp/match_stamps.o -ldl
  echo built $OUT
else
  SFM_AMD_LIB=$OUT python3 $R/tools/exp_matcher_lifetimes.py "${@:2}"
fi
