#!/bin/bash
# k_schur_items: waves per SIMD x chunk loads in flight (build-time knobs SFM_SCHUR_WAVES / SFM_SCHUR_U), d = 10 and d = 6 on the
# random scene, d = 10 on the spatially coherent one (with contiguous XCD row groups too).
#   bash tools/exp_schur_occupancy.sh build            (here: the variant libraries travel with the snapshot)
#   gpurun -- 'bash tools/exp_schur_occupancy.sh run [outdir]'
# Every variant is its OWN library (sfm_amd/lib/libsfm_amd_w<W>_u<U>.so, objects under /tmp), selected with SFM_AMD_LIB: the
# shipped libsfm_amd.so and lib/obj/ are never touched (an earlier form of this script relinked over them and left the last
# variant installed, with build.py's time stamps none the wiser).
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
VARIANTS="5:8 6:6 7:5 8:4 8:3 8:2"
if [ "${1:-run}" = build ]; then
  python3 -m sfm_amd.build > /dev/null
  for v in $VARIANTS; do
    W=${v%%:*}; U=${v##*:}
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form -I$R/include -DSFM_SCHUR_WAVES=$W -DSFM_SCHUR_U=$U \
      -c $R/sfm_amd/csrc/ba.hip -o /tmp/ba_w${W}_u${U}.o
    objs=$(ls $R/sfm_amd/lib/obj/*.o | grep -v /ba.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/sfm_amd/lib/libsfm_amd_w${W}_u${U}.so $objs /tmp/ba_w${W}_u${U}.o -ldl
    echo built $R/sfm_amd/lib/libsfm_amd_w${W}_u${U}.so
  done
  exit 0
fi
OUT=$R/gpurun_out/${2:-exp_occ}
mkdir -p $OUT
BA="--no-cpu-baseline --no-matcher --no-d6 --no-mixed --no-pcg --no-dropin --no-driver-rows --no-alt-camera-solver --no-coherent --no-reference-order"
for v in $VARIANTS; do
  W=${v%%:*}; U=${v##*:}
  export SFM_AMD_LIB=$R/sfm_amd/lib/libsfm_amd_w${W}_u${U}.so
  [ -f $SFM_AMD_LIB ] || { echo "missing $SFM_AMD_LIB: run '$0 build' first"; exit 1; }
  timeout -k 10 200 python3 $R/bench.py $BA > $OUT/d10_w${W}_u${U}.json 2> $OUT/d10_w${W}_u${U}.err
  timeout -k 10 200 python3 $R/bench.py --cam-dim 6 $BA > $OUT/d6_w${W}_u${U}.json 2> $OUT/d6_w${W}_u${U}.err
  timeout -k 10 200 python3 $R/bench.py --visibility nearest $BA > $OUT/coh_w${W}_u${U}.json 2> $OUT/coh_w${W}_u${U}.err
  SFM_XCD_GROUP=contig timeout -k 10 200 python3 $R/bench.py --visibility nearest $BA > $OUT/cohc_w${W}_u${U}.json 2> $OUT/cohc_w${W}_u${U}.err
  echo "done W=$W U=$U"
done
