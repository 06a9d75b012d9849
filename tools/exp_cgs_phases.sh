#!/bin/bash
# Diagnostic copy of the library with per-iteration phase stamps in k_cgs_persist (workgroup 0, thread 0):
#   bash tools/exp_cgs_phases.sh build          (here: the .so travels with the snapshot)
#   gpurun -- 'bash tools/exp_cgs_phases.sh run [cams pts]'
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/sfm_amd/lib/libsfm_amd_cgstamps.so
if [ "${1:-run}" = build ]; then
  python3 -m sfm_amd.build > /dev/null
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form -I$R/include -DSFM_CGS_STAMPS=1 \
    -c $R/sfm_amd/csrc/ba.hip -o /tmp/ba_cgstamps.o 2> /dev/null
  objs=$(ls $R/sfm_amd/lib/obj/*.o | grep -v /ba.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $objs /tmp/ba_cgstamps.o -ldl
  echo built $OUT
else
  SFM_AMD_LIB=$OUT python3 $R/tools/exp_cgs_phases.py "${@:2}"
fi
