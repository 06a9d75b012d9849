#!/usr/bin/env python3
"""gpurun_out/prof_<round>/ (tools/collect_profiles.sh) -> the files kept under profiles/:
  <round>_bench.json, <round>_bench_cfg3.json, <round>_bench_cfg5.json      the bench lines
  <round>_kernel_stats.csv, <round>_kernel_stats_cfg5.csv                   rocprofv3 --kernel-trace --stats summaries
  <round>_pmc_per_kernel.csv, <round>_pmc_summary.json                      HBM bytes per launch per kernel
  <round>_pmc_matcher.txt                                                   SQ counters of the matcher kernel
gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes, so read bytes = 2 * FETCH_SIZE KiB * 1024
(MI355X_MICROARCH.md).    <round>_pmc_summary_coherent.json, <round>_l2_hits.json                   the same kernel on the spatially coherent scene; L2 hit counters
usage: tools/summarise_profiles.py r03"""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
# tools/collect_profiles.sh writes every collection into a time-stamped directory of its own and drops a `done` marker
# last: take the NEWEST complete one (gpurun merges gpurun_out/ back, so older collections stay next to it)
base = os.path.join(ROOT, "gpurun_out", "prof_" + rnd)
runs = sorted(d for d in glob.glob(os.path.join(base, "*")) if os.path.isdir(d) and os.path.exists(os.path.join(d, "done")))
if not runs:
    raise SystemExit(f"no complete collection under {base} (tools/collect_profiles.sh {rnd})")
src = runs[-1]
dst = os.path.join(ROOT, "profiles")
print("summarising", src)


def short(name):
    return name.split("(")[0].replace("void ", "")


def counters(d, wanted=None):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if wanted is None or r["Counter_Name"] in wanted:
                agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def copy_stats(sub, out_name):
    fs = glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True)
    if len(fs) > 1:
        raise SystemExit(f"{len(fs)} kernel_stats files under {src}/{sub}: one collection must hold one")
    if fs:
        shutil.copy(fs[0], os.path.join(dst, out_name))
        return fs[0]


for a, b in (("bench.json", f"{rnd}_bench.json"), ("bench_cfg3.json", f"{rnd}_bench_cfg3.json"), ("bench_cfg5.json", f"{rnd}_bench_cfg5.json")):
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dst, b))
copy_stats("kt", f"{rnd}_kernel_stats.csv")
copy_stats("ktc", f"{rnd}_kernel_stats_cholesky.csv")
copy_stats("kt5", f"{rnd}_kernel_stats_cfg5.csv")

fe, wr = counters(os.path.join(src, "fetch"), {"FETCH_SIZE"}), counters(os.path.join(src, "write"), {"WRITE_SIZE"})
rows = []
for k in sorted(set(fe) | set(wr)):
    f = fe.get(k, {}).get("FETCH_SIZE", []); w = wr.get(k, {}).get("WRITE_SIZE", [])
    fm = sum(f) / len(f) if f else 0.0; wm = sum(w) / len(w) if w else 0.0
    rows.append((k, max(len(f), len(w)), fm, wm, int((2 * fm + wm) * 1024)))
if rows:
    with open(os.path.join(dst, f"{rnd}_pmc_per_kernel.csv"), "w") as fh:
        fh.write("kernel,launches,FETCH_SIZE_mean_KiB,WRITE_SIZE_mean_KiB,hbm_bytes_corrected = (2*FETCH + WRITE)*1024\n")
        for r in rows:
            fh.write(f"\"{r[0]}\",{r[1]},{r[2]:.1f},{r[3]:.1f},{r[4]}\n")
    bench = json.load(open(os.path.join(src, "bench.json")))
    wl = bench["config"]["workload"]
    cams, pts, obs = int(wl.split(" cams")[0].split()[-1]), int(wl.split(" pts")[0].split()[-1]), int(wl.split(" obs")[0].split()[-1])
    d = 10 if "cam block 10" in wl else 6
    pick = lambda prefix: next((r for r in rows if r[0].startswith(prefix)), None)
    summary = {"provenance": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes, --kernel-trace only) -- python3 bench.py --steps 2 "
                             "--warmup 1 --no-cpu-baseline --no-matcher --no-d6 --no-mixed --no-pcg --no-dropin --no-driver-rows, MI355X, " + rnd,
               "correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B -> read bytes = 2 * FETCH_SIZE * 1024 (MI355X_MICROARCH.md)",
               "workload": {"cams": cams, "pts": pts, "obs": obs, "cam_dim": d}}
    for key, prefix in (("k_lin_obs", "k_lin_obs"), ("k_schur_items", "k_schur_items"), ("k_build_G", "k_build_G"),
                        ("k_chol_step", "k_chol_step"), ("k_axpy_step", "k_axpy_step")):
        r = pick(prefix)
        if r:
            summary[key] = {"kernel": r[0], "FETCH_SIZE_KiB": r[2], "WRITE_SIZE_KiB": r[3], "hbm_bytes_per_launch": r[4]}
    if "k_lin_obs" in summary:
        summary["k_lin_obs"]["algorithmic_bytes_per_launch"] = (8 + 16 + 16 + 2 * d * 8 + 48) * obs
    if "k_axpy_step" in summary:
        summary["k_axpy_step"]["note"] = f"calibration: reads 2 x {8 * (cams * d + 3 * pts) / 1e6:.2f} MB, writes 1 x"
    json.dump(summary, open(os.path.join(dst, f"{rnd}_pmc_summary.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in summary.items() if k.startswith("k_")}, indent=1))

# the Schur gather on the spatially coherent scene (bench.py's ba_coherent_scene row reads this file)
fe2, wr2 = counters(os.path.join(src, "fetch_coherent"), {"FETCH_SIZE"}), counters(os.path.join(src, "write_coherent"), {"WRITE_SIZE"})
k2 = next((k for k in fe2 if k.startswith("k_schur_items")), None)
if k2:
    bench = json.load(open(os.path.join(src, "bench.json")))
    wl = bench["config"]["workload"]
    cams, pts, obs = int(wl.split(" cams")[0].split()[-1]), int(wl.split(" pts")[0].split()[-1]), int(wl.split(" obs")[0].split()[-1])
    d = 10 if "cam block 10" in wl else 6
    f = fe2[k2]["FETCH_SIZE"]; w = wr2.get(k2, {}).get("WRITE_SIZE", [0.0])
    fm, wm = sum(f) / len(f), sum(w) / len(w)
    json.dump({"provenance": "as <round>_pmc_summary.json, bench.py --visibility nearest (synth.make_scene(visibility='nearest'))",
               "workload": {"cams": cams, "pts": pts, "obs": obs, "cam_dim": d, "visibility": "nearest"},
               "k_schur_items": {"kernel": k2, "FETCH_SIZE_KiB": fm, "WRITE_SIZE_KiB": wm, "hbm_bytes_per_launch": int((2 * fm + wm) * 1024)}},
              open(os.path.join(dst, f"{rnd}_pmc_summary_coherent.json"), "w"), indent=1)
l2 = {}
for tag, sub in (("random", "l2"), ("coherent", "l2_coherent")):
    cs = counters(os.path.join(src, sub))
    kk = next((k for k in cs if k.startswith("k_schur_items")), None)
    if kk:
        mean = {c: sum(v) / len(v) for c, v in cs[kk].items()}
        mean["hit_rate"] = mean.get("TCC_HIT_sum", 0.0) / max(mean.get("TCC_REQ_sum", 1.0), 1.0)
        l2[tag] = {"kernel": kk, **mean}
if l2:
    l2["note"] = "rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum, mean per launch of k_schur_items (128-byte line requests at the 8 L2s)"
    json.dump(l2, open(os.path.join(dst, f"{rnd}_l2_hits.json"), "w"), indent=1)

m = collections.defaultdict(list)
for sub in ("msq1", "msq2"):
    for k, cs in counters(os.path.join(src, sub)).items():
        if k.startswith("k_knn2_u8<") or k.startswith("k_knn2_u8_direct<"):      # the distance kernel itself, not k_knn2_u8_rerank
            for c, v in cs.items():
                m[c] += v
if m:
    with open(os.path.join(dst, f"{rnd}_pmc_matcher.txt"), "w") as fh:
        fh.write("# k_knn2_u8_direct<4>, 50k x 50k x 128 uint8, library built with -mllvm -amdgpu-mfma-vgpr-form (the shipped flags);\n"
                 "# rocprofv3 --pmc <8 SQ counters> --kernel-trace -- python3 tools/run_matcher.py, two passes, mean per launch\n")
        for c in sorted(m):
            fh.write(f"{c:32s} n={len(m[c]):3d} mean={sum(m[c]) / len(m[c]):16.1f}\n")
        g = lambda c: sum(m[c]) / len(m[c]) if m.get(c) else float("nan")
        # SQ counters are chip-wide sums (SQ_WAVES = every wave of the launch); GRBM_GUI_ACTIVE is summed over the 8 XCDs, so
        # the SIMD-cycles of the launch are (GRBM_GUI_ACTIVE / 8) * 1024.  SQ_ACTIVE_INST_* count quad-cycles,
        # SQ_VALU_MFMA_BUSY_CYCLES cycles (MI355X_MICROARCH.md)
        simd_cycles = g('GRBM_GUI_ACTIVE') / 8 * 1024
        fh.write(f"# VALU instructions per MFMA: {g('SQ_INSTS_VALU') / g('SQ_INSTS_MFMA'):.2f}\n")
        fh.write(f"# MFMA pipe busy: {g('SQ_VALU_MFMA_BUSY_CYCLES') / simd_cycles:.3f} of the launch's SIMD-cycles\n")
        fh.write(f"# VALU busy: {4 * g('SQ_ACTIVE_INST_VALU') / simd_cycles:.3f}; LDS bank conflicts: {g('SQ_LDS_BANK_CONFLICT'):.0f}\n")
    print(open(os.path.join(dst, f"{rnd}_pmc_matcher.txt")).read())

# The bench line of a collection is measured BEFORE its counter passes, so the `traffic` fields bench.py filled in come from
# the newest summary that existed then (the previous round's).  Point them at THIS collection's counters, the file kept beside it.
def _retarget(bench_name, pmc_name, key_path):
    bp, pp = os.path.join(dst, bench_name), os.path.join(dst, pmc_name)
    if not (os.path.exists(bp) and os.path.exists(pp)):
        return
    b, pmc = json.load(open(bp)), json.load(open(pp))
    changed = False
    for r in [b.get("roofline")] + list(b.get("rooflines") or []):
        if not r:
            continue
        for kk in ("k_schur_items", "k_lin_obs"):
            if r["kernel"].startswith(kk) and kk in pmc and r.get("traffic") is not None:
                r["traffic"] = pmc[kk]["hbm_bytes_per_launch"]
                r["traffic_source"] = "profiles/" + pmc_name + " (FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes of the same collection)"
                if "unique_bytes" in r:
                    r["wasted_traffic"] = round(r["traffic"] / r["unique_bytes"], 2)
                changed = True
    if changed:
        json.dump(b, open(bp, "w"))


_retarget(f"{rnd}_bench.json", f"{rnd}_pmc_summary.json", None)
coh_b, coh_p = os.path.join(dst, f"{rnd}_bench.json"), os.path.join(dst, f"{rnd}_pmc_summary_coherent.json")
if os.path.exists(coh_b) and os.path.exists(coh_p):
    b, pmc = json.load(open(coh_b)), json.load(open(coh_p))
    k = (b.get("ba_coherent_scene") or {}).get("k_schur_items")
    if k and "k_schur_items" in pmc:
        k["traffic"] = pmc["k_schur_items"]["hbm_bytes_per_launch"]
        k["wasted_traffic"] = round(k["traffic"] / k["unique_bytes"], 2)
        k["traffic_source"] = "profiles/" + os.path.basename(coh_p)
        json.dump(b, open(coh_b, "w"))
