#!/bin/bash
# Collect the rocprofv3 material behind profiles/rNN_* on an MI355X box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh r03
# Writes under gpurun_out/prof_<round>/; tools/summarise_profiles.py turns that into the files kept in profiles/.
# Counter passes are separate from each other and carry --kernel-trace only (MI355X_MICROARCH.md, rocprofv3 PMC slots).
set -u
ROUND=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
# every collection gets a directory of its own (gpurun MERGES gpurun_out/ back, it does not replace it: a fixed directory
# would accumulate one file set per collection and the summariser could pick a stale one - round 2 did)
STAMP=$(date -u +%Y%m%dT%H%M%SZ)
OUT=$R/gpurun_out/prof_$ROUND/$STAMP
mkdir -p "$OUT"
(cd $R && git rev-parse HEAD 2>/dev/null; sha256sum sfm_amd/lib/libsfm_amd.so bench.py) > $OUT/provenance.txt 2>&1
cd /tmp && export TMPDIR=/tmp
BA="--no-cpu-baseline --no-matcher --no-d6 --no-mixed --no-pcg --no-dropin --no-driver-rows --no-alt-camera-solver --no-coherent --no-reference-order"
echo "[1/7] bench line (defaults)";           python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
echo "[2/7] kernel trace + stats (BA + matcher)"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-d6 --no-mixed --no-pcg --no-dropin --no-driver-rows --no-alt-camera-solver --no-coherent > $OUT/kt.log 2>&1 || exit 1
echo "[2b/7] kernel trace + stats with the factorisation forced"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktc -- python3 $R/bench.py --steps 6 --warmup 3 --camera-solver cholesky $BA > $OUT/ktc.log 2>&1 || exit 1
echo "[3/7] PMC FETCH_SIZE";  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $R/bench.py --steps 2 --warmup 1 $BA > $OUT/fetch.log 2>&1 || exit 1
echo "[4/7] PMC WRITE_SIZE";  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $R/bench.py --steps 2 --warmup 1 $BA > $OUT/write.log 2>&1 || exit 1
echo "[4b/7] PMC FETCH_SIZE on the spatially coherent scene, L2 hit counters on both scenes"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch_coherent -- python3 $R/bench.py --steps 2 --warmup 1 --visibility nearest $BA > $OUT/fetch_coherent.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write_coherent -- python3 $R/bench.py --steps 2 --warmup 1 --visibility nearest $BA > $OUT/write_coherent.log 2>&1 || exit 1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $OUT/l2 -- python3 $R/bench.py --steps 2 --warmup 1 $BA > $OUT/l2.log 2>&1 || exit 1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $OUT/l2_coherent -- python3 $R/bench.py --steps 2 --warmup 1 --visibility nearest $BA > $OUT/l2_coherent.log 2>&1 || exit 1
echo "[5/7] matcher SQ counters (two passes)"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/msq1 -- python3 $R/tools/run_matcher.py > $OUT/msq1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/msq2 -- python3 $R/tools/run_matcher.py > $OUT/msq2.log 2>&1 || exit 1
echo "[6/7] cfg3 bench line (50 cams / 20k pts / 200k obs)"
python3 $R/bench.py --cams 50 --pts 20000 --no-matcher --no-driver-rows --no-dropin --no-coherent > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err || exit 1
echo "[7/7] cfg5 bench line (1000 cams / 500k pts / 5M obs) + kernel stats"
python3 $R/bench.py --cams 1000 --pts 500000 --steps 3 --warmup 1 --no-matcher --no-driver-rows --no-dropin --no-d6 --no-cpu-baseline --no-coherent --no-reference-order > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt5 -- python3 $R/bench.py --cams 1000 --pts 500000 --steps 2 --warmup 1 --no-matcher --no-driver-rows --no-dropin --no-d6 --no-mixed --no-pcg --no-cpu-baseline --no-alt-camera-solver --no-coherent --no-reference-order > $OUT/kt5.log 2>&1 || exit 1
# keep only what the summariser reads (the raw traces are large)
find $OUT -name "*kernel_trace.csv" -size +20M -delete
du -sh $OUT
date -u +%Y%m%dT%H%M%SZ > $OUT/done
echo done
