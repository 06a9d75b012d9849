#!/usr/bin/env python3
"""Print a per-kernel table (VGPR/AGPR/SGPR, spills, LDS, occupancy) from hipcc's
-Rpass-analysis=kernel-resource-usage remarks.  usage: tools/kernel_resources.py file.hip"""
import re, subprocess, sys
src = sys.argv[1]
out = subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-Iinclude", "-mllvm", "-amdgpu-mfma-vgpr-form", "-c", src,
                      "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
cur = None; rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z /\[\]]+): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
print(f"{'kernel':44s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'vspill':>6s} {'LDS':>7s} {'occ':>4s}")
for k, r in rows.items():
    print(f"{k[:44]:44s} {r.get('VGPRs',0):5d} {r.get('AGPRs',0):5d} {r.get('SGPRs',0):5d} {r.get('VGPRs Spill',0):6d} "
          f"{r.get('LDS Size [bytes/block]',0):7d} {r.get('Occupancy [waves/SIMD]',0):4d}")
