#!/bin/bash
# L2 hit / miss counters of k_schur_items on the random (BASELINE) and the spatially coherent scene.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-exp_l2}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BA="--no-cpu-baseline --no-matcher --no-d6 --no-mixed --no-pcg --no-dropin --no-driver-rows --no-alt-camera-solver --no-coherent --no-reference-order"
for vis in random nearest; do for grp in mod8 contig; do
  SFM_XCD_GROUP=$grp timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $OUT/l2_${vis}_${grp} -- python3 $R/bench.py --steps 2 --warmup 1 --visibility $vis $BA > $OUT/l2_${vis}_${grp}.log 2>&1 || exit 1
  python3 $R/tools/pmc_summary.py k_schur_items $OUT/l2_${vis}_${grp} > $OUT/l2_${vis}_${grp}.txt
  find $OUT/l2_${vis}_${grp} -name "*kernel_trace.csv" -delete
done; done
echo done
