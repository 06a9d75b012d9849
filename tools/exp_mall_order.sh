#!/bin/bash
# cfg5 (1000 cams / 500k pts): does the order in which the tile-streaming CG walks its 405-MB triangle matter to the memory-side
# cache (256 MB)?  A cyclic walk is the worst case for an LRU cache smaller than the walk; walking back and forth finds the tail
# of the previous pass still there.  Knobs: SFM_CGB_ZIGZAG (0: always ascending, 1: odd launches descending, 2: even launches
# descending), SFM_SCALE_REV (k_scale_system_lower: bit 0 column strips descending, bit 1 rows descending - the kernel of sharded
# problems since k_schur_assemble_scaled; the sweep therefore runs with SFM_SCHUR_FUSE_SCALE=0).  ZZ / REV: the values swept.
# (k_schur_assemble's order was swept the same way - no effect either way, knob removed.)
#   gpurun -- 'bash tools/exp_mall_order.sh [outdir]'
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${1:-exp_mall}
mkdir -p $OUT
BA="--cams 1000 --pts 500000 --steps 3 --warmup 1 --no-cpu-baseline --no-matcher --no-d6 --no-mixed --no-pcg --no-dropin --no-driver-rows --no-alt-camera-solver --no-coherent --no-reference-order"
for zz in ${ZZ:-0 1 2}; do for rev in ${REV:-0 1 3}; do
  SFM_SCHUR_FUSE_SCALE=0 SFM_CGB_ZIGZAG=$zz SFM_SCALE_REV=$rev timeout -k 10 300 python3 $R/bench.py $BA > $OUT/zz${zz}_rev${rev}.json 2> $OUT/zz${zz}_rev${rev}.err
  python3 - $OUT/zz${zz}_rev${rev}.json $zz $rev <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); k = d["ba"]["kernels"]
print("zigzag", sys.argv[2], "scale_rev", sys.argv[3], "LM-it/s %.2f" % d["value"], "chol %.0f trsv %.0f schur %.0f (after the gather: %.0f) us" % (k["chol"]["us_per_launch"], k["trsv"]["us_per_launch"], k["schur"]["us_per_launch"], k["schur"]["us_per_launch"] - k["schur_items"]["us_per_launch"]), flush=True)
PY
done; done
