#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE; each its own run, as
MI355X_MICROARCH.md prescribes) -> profiles/r01_pmc_per_kernel.csv + r01_pmc_summary.json.
gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes, so read bytes = 2 * FETCH_SIZE KiB * 1024.
usage: tools/pmc_per_kernel.py <fetch_dir> <write_dir> <out_csv> <out_json> cams pts obs cam_dim"""
import collections, csv, glob, json, sys


def mean_per_kernel(d, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def main():
    fd, wd, out_csv, out_json = sys.argv[1:5]
    cams, pts, obs, d = map(int, sys.argv[5:9])
    fe, wr = mean_per_kernel(fd, "FETCH_SIZE"), mean_per_kernel(wd, "WRITE_SIZE")
    rows = []
    for k in sorted(set(fe) | set(wr)):
        f, n = fe.get(k, (0.0, 0)); w, _ = wr.get(k, (0.0, 0))
        rows.append((k, n, f, w, int((2 * f + w) * 1024)))
    with open(out_csv, "w") as fh:
        fh.write("kernel,launches,FETCH_SIZE_mean_KiB,WRITE_SIZE_mean_KiB,hbm_bytes_corrected = (2*FETCH + WRITE)*1024\n")
        for r in rows:
            fh.write(f"{r[0]},{r[1]},{r[2]:.1f},{r[3]:.1f},{r[4]}\n")
    lin = [r for r in rows if r[0].startswith("k_lin_obs")][0]
    axpy = [r for r in rows if r[0].startswith("k_axpy_step")]
    summary = {
        "provenance": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --kernel-trace -- python3 bench.py --steps 2 "
                      "--warmup 1 --no-cpu-baseline --no-matcher --no-d6 --no-driver-rows, MI355X, round 1 (final kernels)",
        "correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B -> read bytes = 2 * FETCH_SIZE * 1024 (MI355X_MICROARCH.md, HBM)"
                      + (f"; check on k_axpy_step (reads 2 x {8 * (cams * d + 3 * pts) / 1e6:.2f} MB, writes 1 x): FETCH_SIZE "
                         f"{axpy[0][2]:.1f} KiB, WRITE_SIZE {axpy[0][3]:.1f} KiB" if axpy else ""),
        "workload": {"cams": cams, "pts": pts, "obs": obs, "cam_dim": d},
        "k_lin_obs": {"FETCH_SIZE_KiB": lin[2], "WRITE_SIZE_KiB": lin[3], "hbm_bytes_per_launch": lin[4],
                      "algorithmic_bytes_per_launch": (8 + 16 + 16 + 2 * d * 8 + 48) * obs},
    }
    json.dump(summary, open(out_json, "w"), indent=1)
    print(json.dumps(summary["k_lin_obs"]))


if __name__ == "__main__":
    main()
