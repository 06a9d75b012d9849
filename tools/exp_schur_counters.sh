#!/bin/bash
# Counter evidence for k_schur_items on the random (BASELINE) and the spatially coherent scene: L2 requests / hits / misses and
# what goes out to the fabric, how much of that reaches DRAM (the rest is served by the Infinity Cache), request sizes, the L1's
# view (requests it passes on, their latency, stalls) and address translation.  Four --pmc passes per scene (separate runs with
# --kernel-trace only, MI355X_MICROARCH.md), summaries under gpurun_out/<dir>/.
#   gpurun -- 'bash tools/exp_schur_counters.sh [dir]'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-schur_counters}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BA="--no-cpu-baseline --no-matcher --no-d6 --no-mixed --no-pcg --no-dropin --no-driver-rows --no-alt-camera-solver --no-coherent --no-reference-order"
P1="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
P2="TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
P3="TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"
P4="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCC_TAG_STALL_sum"
for vis in random nearest; do
  i=0
  for P in "$P1" "$P2" "$P3" "$P4"; do
    i=$((i + 1))
    timeout -k 10 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/${vis}_p$i -- python3 $R/bench.py --steps 2 --warmup 1 --visibility $vis $BA > $OUT/${vis}_p$i.log 2>&1 || exit 1
    python3 $R/tools/pmc_summary.py k_schur_items $OUT/${vis}_p$i > $OUT/${vis}_p$i.txt
    find $OUT/${vis}_p$i -name "*.csv" -delete
    echo "$vis pass $i done"
  done
done
echo done
