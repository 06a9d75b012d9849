set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3u
mkdir -p $OUT
BA="--no-cpu-baseline --no-matcher --no-d6 --no-mixed --no-pcg --no-dropin --no-driver-rows --no-alt-camera-solver --no-coherent"
for cfg in "5 8" "6 6" "7 5" "8 4"; do
  set -- $cfg
  (cd $R/sfm_amd/csrc && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form -I../../include -DSFM_SCHUR_WAVES=$1 -DSFM_SCHUR_U=$2 -c ba.hip -o ../lib/obj/ba.o 2>/dev/null && cd ../lib && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libsfm_amd.so obj/ctx.o obj/ba.o obj/problem.o obj/trf.o obj/dense.o obj/match.o obj/driver.o obj/comm_rccl.o -ldl) || exit 1
  timeout -k 10 200 python3 $R/bench.py --visibility nearest $BA > $OUT/coh_w$1_u$2.json 2> $OUT/coh_w$1_u$2.err
  SFM_XCD_GROUP=contig timeout -k 10 200 python3 $R/bench.py --visibility nearest $BA > $OUT/cohc_w$1_u$2.json 2> $OUT/cohc_w$1_u$2.err
  echo "done W=$1 U=$2"
done
