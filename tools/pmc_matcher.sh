#!/bin/bash
# SQ counters of the kNN kernel alone (two --pmc passes), printed as per-launch means.  Run on the GPU box:
#   bash tools/pmc_matcher.sh [tag]      -> gpurun_out/pmcm_<tag>/summary.txt
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-x}
OUT=$R/gpurun_out/pmcm_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/p1 -- python3 $R/tools/run_matcher.py > $OUT/p1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p2 -- python3 $R/tools/run_matcher.py > $OUT/p2.log 2>&1 || exit 1
python3 - "$OUT" <<'PY' > $OUT/summary.txt
import sys, glob, csv, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_knn2_u8<" in row["Kernel_Name"] or "k_knn2_u8_direct<" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
for k in sorted(m): print(f"{k:32s} n={len(acc[k]):3d} mean={m[k]:16.1f}")
if "GRBM_GUI_ACTIVE" in m:
    simd_cycles = m["GRBM_GUI_ACTIVE"] / 8 * 1024
    print("cycles per XCD", m["GRBM_GUI_ACTIVE"] / 8)
    for k in ("SQ_VALU_MFMA_BUSY_CYCLES",): print(k, "/ SIMD-cycles", m.get(k, 0) / simd_cycles)
    for k in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"):
        if k in m: print(k, "x4 / SIMD-cycles", 4 * m[k] / simd_cycles)
    if "SQ_INSTS_MFMA" in m: print("VALU per MFMA", m["SQ_INSTS_VALU"] / m["SQ_INSTS_MFMA"], "SALU per MFMA", m.get("SQ_INSTS_SALU", 0) / m["SQ_INSTS_MFMA"])
PY
cat $OUT/summary.txt
