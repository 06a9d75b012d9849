#!/bin/bash
# k-rank item order inside a block row (SFM_XCD_ORDER=krank) on both scenes: kernel time, fabric traffic, L2 hits
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-exp_order}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BA="--no-cpu-baseline --no-matcher --no-d6 --no-mixed --no-pcg --no-dropin --no-driver-rows --no-alt-camera-solver"
for ord in plain krank; do for grp in mod8 contig; do
  SFM_XCD_ORDER=$ord SFM_XCD_GROUP=$grp timeout -k 10 200 python3 $R/bench.py $BA > $OUT/bench_${ord}_${grp}.json 2> $OUT/bench_${ord}_${grp}.err || exit 1
done; done
for vis in random nearest; do for ord in plain krank; do
  SFM_XCD_ORDER=$ord timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $OUT/l2_${vis}_${ord} -- python3 $R/bench.py --steps 2 --warmup 1 --visibility $vis --no-coherent $BA > $OUT/l2_${vis}_${ord}.log 2>&1 || exit 1
  python3 $R/tools/pmc_summary.py k_schur_items $OUT/l2_${vis}_${ord} > $OUT/l2_${vis}_${ord}.txt
  find $OUT/l2_${vis}_${ord} -name "*kernel_trace.csv" -delete
done; done
echo done
