#!/usr/bin/env python3
"""Plan, check and time the camera-tile prototype (tile_route.hip) on the bench's scenes.
usage: run.py [visibility=nearest] [cams=200] [pts=100000] [chunk=512]
Builds libtile_route.so next to itself when it is missing (hipcc, gfx950)."""
import ctypes as C, os, subprocess, sys, time
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import torch
from sfm_amd import synth

T, D, GS, MAXL = 8, 10, 32, 10
vis_kind = sys.argv[1] if len(sys.argv) > 1 else "nearest"
n_cams = int(sys.argv[2]) if len(sys.argv) > 2 else 200
n_pts = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
CH = int(sys.argv[4]) if len(sys.argv) > 4 else 512
lib_path = os.path.join(HERE, "libtile_route.so")
if not os.path.exists(lib_path):
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-o", lib_path,
                    os.path.join(HERE, "tile_route.hip")], check=True)
lib = C.CDLL(lib_path)

L = 10
sc = synth.make_scene(n_cams, n_pts, obs_per_point=L, seed=1004, noise_px=0.5, pt_sigma=0.02, cam_sigma=0.002, visibility=vis_kind)
cam = sc.cam_idx.reshape(n_pts, L)                       # ascending inside a track
grp = cam // T
n_grp = (n_cams + T - 1) // T

# ---- plan: one visit per (track, pair of its groups gr >= gq)
t0 = time.time()
vis_dtype = np.dtype([("o0", "<i4"), ("maskR", "u1"), ("maskQ", "u1"), ("len", "u1"), ("pad", "u1"), ("slotR", "i1", (8,)), ("slotQ", "i1", (8,))])
assert vis_dtype.itemsize == 24
recs, tiles = [], []
present = np.zeros((n_pts, n_grp), dtype=bool)
present[np.arange(n_pts)[:, None], grp] = True
for gr in range(n_grp):
    for gq in range(gr + 1):
        tr = np.nonzero(present[:, gr] & present[:, gq])[0]
        if tr.size == 0:
            continue
        r = np.zeros(tr.size, dtype=vis_dtype)
        r["o0"] = tr * L; r["len"] = L
        r["slotR"] = -1; r["slotQ"] = -1
        c = cam[tr]                                        # [n, L]
        for pos in range(L):
            inR = grp[tr, pos] == gr; inQ = grp[tr, pos] == gq
            sR = c[:, pos] - gr * T; sQ = c[:, pos] - gq * T
            r["slotR"][np.nonzero(inR)[0], sR[inR]] = pos
            r["slotQ"][np.nonzero(inQ)[0], sQ[inQ]] = pos
        r["maskR"] = ((r["slotR"] >= 0) * (1 << np.arange(8))).sum(axis=1).astype(np.uint8)
        r["maskQ"] = ((r["slotQ"] >= 0) * (1 << np.arange(8))).sum(axis=1).astype(np.uint8)
        recs.append(r); tiles.append((gr, gq, tr.size))
vis = np.concatenate(recs)
item_gr, item_gq, item_v0, item_v1, tile_i0, tile_i1 = [], [], [], [], [], []
v = 0
for gr, gq, n in tiles:
    tile_i0.append(len(item_gr))
    for a in range(0, n, CH):
        item_gr.append(gr); item_gq.append(gq); item_v0.append(v + a); item_v1.append(v + min(n, a + CH))
    tile_i1.append(len(item_gr))
    v += n
n_items, n_tiles = len(item_gr), len(tiles)
pairs = 0
for (gr, gq, n), r in zip(tiles, recs):
    nR = (r["slotR"] >= 0).sum(axis=1); nQ = (r["slotQ"] >= 0).sum(axis=1)
    pairs += int((nR * (nR + 1) // 2).sum()) if gr == gq else int((nR * nQ).sum())
staged = int((vis["slotR"] >= 0).sum() + (vis["slotQ"] >= 0).sum())
print(f"{vis_kind}: {n_tiles} tiles of {n_grp * (n_grp + 1) // 2}, {len(vis)} visits, {n_items} items of <= {CH} tracks, {pairs} block pairs "
      f"(the pair list of the library: {n_pts * L * (L + 1) // 2}), {staged} G blocks staged = {staged * 256 / 1e9:.2f} GB, planning {time.time() - t0:.1f} s on the host (numpy)")

dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
Gh = np.zeros((n_pts * L, GS)); Gh[:, : 3 * D] = rng.normal(size=(n_pts * L, 3 * D))
G = torch.from_numpy(Gh).to(dev)
tg = lambda a, dt: torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=dt))).to(dev)
d_gr, d_gq, d_v0, d_v1 = tg(item_gr, np.int32), tg(item_gq, np.int32), tg(item_v0, np.int32), tg(item_v1, np.int32)
d_vis = torch.from_numpy(vis.view(np.uint8).reshape(-1)).to(dev)
d_i0, d_i1 = tg(tile_i0, np.int32), tg(tile_i1, np.int32)
part = torch.empty((n_items, T, T, D * D), dtype=torch.float64, device=dev)
out = torch.empty((n_tiles, T, T, D * D), dtype=torch.float64, device=dev)
p = lambda t: C.c_void_p(t.data_ptr())


def run():
    rc = lib.tile_route_run(C.c_void_p(torch.cuda.current_stream().cuda_stream), n_items, n_tiles, p(d_gr), p(d_gq), p(d_v0), p(d_v1), p(d_vis),
                            p(G), p(part), p(d_i0), p(d_i1), p(out))
    assert rc == 0, rc


run(); torch.cuda.synchronize()
# ---- check a few tiles against numpy
res = out.cpu().numpy()
Gb = Gh[:, : 3 * D].reshape(n_pts, L, 3, D)
worst = 0.0
for ti in sorted(set([0, n_tiles // 2, n_tiles - 1, min(3, n_tiles - 1)])):
    gr, gq, n = tiles[ti]
    ref = np.zeros((T, T, D, D))
    r = recs[ti]
    tr = r["o0"] // L
    for a in range(T):
        for b in range(T):
            if gr == gq and b > a:
                continue
            both = np.nonzero((r["slotR"][:, a] >= 0) & (r["slotQ"][:, b] >= 0))[0]
            if both.size:
                A = Gb[tr[both], r["slotR"][both, a]]          # [m, 3, D]
                B = Gb[tr[both], r["slotQ"][both, b]]
                ref[a, b] = np.einsum("nki,nkj->ij", A, B)
    got = res[ti].reshape(T, T, D, D)
    worst = max(worst, float(np.max(np.abs(got - ref)) / max(1.0, np.max(np.abs(ref)))))
print(f"check against numpy on 4 tiles: worst relative difference {worst:.1e}")
assert worst < 1e-12
# ---- time
for _ in range(3):
    run()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
reps = 20
ev[0].record()
for _ in range(reps):
    run()
ev[1].record(); torch.cuda.synchronize()
print(f"tile kernel + reduction: {ev[0].elapsed_time(ev[1]) / reps * 1e3:.1f} us per build of S "
      f"(the library's pair-list gather + assemble on this scene: profiles/r04_bench.json, kernels_us.schur)")
