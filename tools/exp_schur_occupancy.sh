#!/bin/bash
# k_schur_items: waves per SIMD x chunk loads in flight (build-time knobs SFM_SCHUR_WAVES / SFM_SCHUR_U), d = 10 and d = 6
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-exp_occ}
mkdir -p $OUT
BA="--no-cpu-baseline --no-matcher --no-d6 --no-mixed --no-pcg --no-dropin --no-driver-rows --no-alt-camera-solver --no-coherent"
for cfg in "5 8" "8 4" "6 6" "7 5" "8 3" "8 2"; do
  set -- $cfg
  (cd $R/sfm_amd/csrc && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form -I../../include -DSFM_SCHUR_WAVES=$1 -DSFM_SCHUR_U=$2 -c ba.hip -o ../lib/obj/ba.o && cd ../lib && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libsfm_amd.so obj/ctx.o obj/ba.o obj/problem.o obj/trf.o obj/dense.o obj/match.o obj/driver.o obj/comm_rccl.o -ldl) || exit 1
  timeout -k 10 200 python3 $R/bench.py $BA > $OUT/d10_w$1_u$2.json 2> $OUT/d10_w$1_u$2.err
  timeout -k 10 200 python3 $R/bench.py --cam-dim 6 $BA > $OUT/d6_w$1_u$2.json 2> $OUT/d6_w$1_u$2.err
  echo "done W=$1 U=$2"
done
