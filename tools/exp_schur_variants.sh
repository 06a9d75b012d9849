#!/bin/bash
# Schur-gather experiments (round 3): XCD grouping x G block stride x scene, kernel times from bench.py's HIP events and
# fabric traffic of k_schur_items from a FETCH_SIZE pass.  Run through gpurun from the repo root.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-exp_schur}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BA="--no-cpu-baseline --no-matcher --no-d6 --no-mixed --no-pcg --no-dropin --no-driver-rows --no-alt-camera-solver --no-reference-order"
for grp in mod8 contig; do for pad in 0 1; do
  echo "== bench group=$grp pad=$pad"
  SFM_XCD_GROUP=$grp SFM_G_PAD=$pad timeout -k 10 200 python3 $R/bench.py $BA > $OUT/bench_${grp}_pad${pad}.json 2> $OUT/bench_${grp}_pad${pad}.err || exit 1
done; done
for vis in random nearest; do for pad in 0 1; do for grp in mod8 contig; do
  echo "== pmc vis=$vis pad=$pad group=$grp"
  SFM_XCD_GROUP=$grp SFM_G_PAD=$pad timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch_${vis}_${grp}_pad${pad} -- python3 $R/bench.py --steps 2 --warmup 1 --visibility $vis --no-coherent $BA > $OUT/fetch_${vis}_${grp}_pad${pad}.log 2>&1 || exit 1
  python3 $R/tools/pmc_summary.py k_schur_items $OUT/fetch_${vis}_${grp}_pad${pad} > $OUT/fetch_${vis}_${grp}_pad${pad}.txt
  find $OUT/fetch_${vis}_${grp}_pad${pad} -name "*kernel_trace.csv" -delete
done; done; done
echo done
