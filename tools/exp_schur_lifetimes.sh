#!/bin/bash
# Diagnostic copy of the library with wave begin / end stamps in k_schur_items:
#   bash tools/exp_schur_lifetimes.sh build            (here: the .so travels with the snapshot)
#   gpurun -- 'bash tools/exp_schur_lifetimes.sh run [cams pts [visibility]]'
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/sfm_amd/lib/libsfm_amd_stamps.so
if [ "${1:-run}" = build ]; then
  python3 -m sfm_amd.build > /dev/null
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form -I$R/include ${EXTRA_FLAGS:-} -DSFM_SCHUR_STAMPS=1 \
    -c $R/sfm_amd/csrc/ba.hip -o /tmp/ba_stamps.o
  objs=$(ls $R/sfm_amd/lib/obj/*.o | grep -v /ba.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $objs /tmp/ba_stamps.o -ldl
  echo built $OUT
else
  SFM_AMD_LIB=$OUT python3 $R/tools/exp_schur_lifetimes.py "${@:2}"
fi
