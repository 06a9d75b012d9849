"""GPU parity tests of the matcher (through the C-ABI) against the CPU oracle: bit-exact indices,
float32-exact distances."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def gpu_knn2(d1, d2, metric="auto"):
    from sfm_amd import matcher
    i1, i2, a, b = matcher.knn2(d1, d2, metric)
    return i1.cpu().numpy(), i2.cpu().numpy(), a.cpu().numpy(), b.cpu().numpy()


def assert_knn_equal(got, ref):
    for g, r, name in zip(got, ref, ("idx1", "idx2", "d1", "d2")):
        assert np.array_equal(g, r), f"{name}: {np.sum(g != r)} of {len(r)} differ"


@pytest.mark.parametrize("nq,nt,dim", [(64, 64, 128), (1, 2, 128), (257, 129, 128), (1000, 1000, 128),
                                       (777, 3001, 128), (500, 2500, 64), (300, 700, 32)])
def test_l2_u8_matches_oracle(gpu_ready, nq, nt, dim):
    from oracle import matcher_oracle as mo
    from sfm_amd import synth
    d1, d2 = synth.make_descriptors(nq, nt, seed=nq + nt, dim=dim)
    u1, u2 = d1.astype(np.uint8), d2.astype(np.uint8)
    ref = mo.knn2(u1, u2, "l2")
    assert_knn_equal(gpu_knn2(u1, u2, "l2"), ref)
    # integer-valued float32 (what SIFT emits) takes the same exact path
    assert_knn_equal(gpu_knn2(d1, d2, "l2"), ref)


def test_l2_u8_uniform_random_bytes(gpu_ready):
    """Uniform bytes: large d^2 (up to ~2.2e6), every int8 value incl. -128 / 127 on both operands."""
    from oracle import matcher_oracle as mo
    rng = np.random.default_rng(7)
    u1 = rng.integers(0, 256, size=(900, 128), dtype=np.uint8)
    u2 = rng.integers(0, 256, size=(1500, 128), dtype=np.uint8)
    u1[0] = 0; u1[1] = 255; u2[0] = 255; u2[1] = 0; u2[2] = 128; u1[2] = 127
    assert_knn_equal(gpu_knn2(u1, u2, "l2"), mo.knn2(u1, u2, "l2"))


def test_l2_ties_and_duplicates(gpu_ready):
    """Exact duplicates in the train set: the lower train index must come first; d1 == d2 == 0."""
    from oracle import matcher_oracle as mo
    from sfm_amd import synth
    d1, d2 = synth.make_descriptors(300, 600, seed=3)
    u1, u2 = d1.astype(np.uint8), d2.astype(np.uint8)
    u2[500] = u2[10]; u2[599] = u2[10]; u2[130] = u2[129]       # duplicates across chunks and splits
    u1[5] = u2[10]; u1[6] = u2[129]
    ref = mo.knn2(u1, u2, "l2")
    got = gpu_knn2(u1, u2, "l2")
    assert_knn_equal(got, ref)
    assert (got[0][5], got[1][5], got[2][5], got[3][5]) == (10, 500, 0.0, 0.0)
    assert (got[0][6], got[1][6]) == (129, 130)


def test_l2_f32_general_floats(gpu_ready):
    from oracle import matcher_oracle as mo
    from sfm_amd import synth
    d1, d2 = synth.make_descriptors(400, 900, seed=11, kind="uniform")
    assert_knn_equal(gpu_knn2(d1, d2, "l2"), mo.knn2(d1, d2, "l2"))


@pytest.mark.parametrize("nq,nt", [(64, 64), (1000, 1000), (333, 2049)])
def test_hamming_matches_oracle(gpu_ready, nq, nt):
    """The in-tree reference path: ORB 32-byte descriptors, NORM_HAMMING (find_matches.py:144)."""
    from oracle import matcher_oracle as mo
    rng = np.random.default_rng(nq)
    b1 = rng.integers(0, 256, size=(nq, 32), dtype=np.uint8)
    b2 = rng.integers(0, 256, size=(nt, 32), dtype=np.uint8)
    b2[: min(nq, nt) // 2] = b1[: min(nq, nt) // 2] ^ rng.integers(0, 2, size=(min(nq, nt) // 2, 32), dtype=np.uint8)
    assert_knn_equal(gpu_knn2(b1, b2, "auto"), mo.knn2(b1, b2, "hamming"))


@pytest.mark.parametrize("nq,nt,nbytes", [(700, 3000, 16), (4001, 6100, 32), (13000, 2500, 16), (257, 5000, 64), (2, 2, 32),
                                          (2000, 20001, 32), (300, 30017, 32)])
def test_hamming_sizes_and_widths(gpu_ready, nq, nt, nbytes):
    """Hamming on 128 / 256 bits runs as exact uint8 L2 over the unpacked bits on the int8 kernels (k_unpack_bits; 128 bits can
    take the LDS-free kernel, 256 bits from 8e6 distances on its dim-256 instantiation - k_knn2_u8_direct<2, 8>: odd tile counts,
    several train splits, a last tile of one row - and the LDS kernel below, both with the candidate filter from 2,048 train rows on);
    512 bits stay on the popcount kernel.  All against the NumPy oracle: many exact ties (distances are small integers), so
    the lowest-train-index rule decides most second neighbours."""
    from oracle import matcher_oracle as mo
    rng = np.random.default_rng(nq + nbytes)
    b1 = rng.integers(0, 256, size=(nq, nbytes), dtype=np.uint8)
    b2 = rng.integers(0, 256, size=(nt, nbytes), dtype=np.uint8)
    k = min(nq, nt) // 2
    b2[:k] = b1[:k] ^ (rng.integers(0, 256, size=(k, nbytes), dtype=np.uint8) & rng.integers(0, 256, size=(k, nbytes), dtype=np.uint8)
                       & rng.integers(0, 256, size=(k, nbytes), dtype=np.uint8))
    b2[k: k + k // 4] = b2[:k // 4]                                  # exact duplicates among the train rows
    assert_knn_equal(gpu_knn2(b1, b2, "hamming"), mo.knn2(b1, b2, "hamming"))


def test_match_features_contract(gpu_ready):
    """List of DMatch-like objects, query-ordered, one per query, strict ratio test in double."""
    from oracle import matcher_oracle as mo
    from sfm_amd import synth
    from sfm_amd.matcher import ImageMatcher
    d1, d2 = synth.make_descriptors(2000, 2300, seed=21)
    ms = ImageMatcher().match_features(d1, d2)
    q, t, d = mo.match_features(d1, d2, 0.75, "l2")
    assert [m.queryIdx for m in ms] == q.tolist()
    assert [m.trainIdx for m in ms] == t.tolist()
    assert np.array_equal(np.array([m.distance for m in ms], dtype=np.float32), d)
    assert all(m.imgIdx == 0 for m in ms)
    assert np.all(np.diff(q) > 0) and 0 < len(ms) < 2000
    # distances are sqrtf of an integer, like the reference's shipped match files
    d2i = np.rint(d.astype(np.float64) ** 2)
    assert np.array_equal(np.sqrt(d2i.astype(np.float32)), d)
    # degenerate sizes as the reference: one train row -> ValueError from the unpacking at find_matches.py:151,
    # no train rows / no query rows -> []
    with pytest.raises(ValueError, match="not enough values to unpack"):
        ImageMatcher().match_features(d1, d2[:1])
    assert ImageMatcher().match_features(d1, d2[:0]) == []
    assert ImageMatcher().match_features(d1[:0], d2) == []


def test_ratio_boundary_is_strict(gpu_ready):
    """d1 == 0.75 * d2 exactly must be rejected (find_matches.py:152 uses '<')."""
    from sfm_amd.matcher import match_arrays
    q = np.zeros((1, 128), np.uint8)
    t = np.zeros((3, 128), np.uint8)
    t[0, 0] = 3; t[1, 0] = 4; t[2, 0] = 200                       # d = 3, 4, 200 -> 3 < 0.75*4 is False
    assert len(match_arrays(q, t, 0.75, "l2")[0]) == 0
    t[0, 0] = 2
    qi, ti, d = match_arrays(q, t, 0.75, "l2")
    assert (qi.tolist(), ti.tolist(), d.tolist()) == ([0], [0], [2.0])


def far_apart_sets(nq, nt, dim, seed):
    """Every pair far apart AND closely spaced: three quarters of the bytes are 0 (query) against 255 (train),
    the rest differ by 0..2, so all d^2 sit within ~100 of 0.75 * dim * 255^2 (6.2e6 at dim 128, above 2^22),
    where about one in five neighbouring integers shares its float32 root with the next."""
    rng = np.random.default_rng(seed)
    near = dim // 4
    q = np.zeros((nq, dim), np.uint8)
    t = np.full((nt, dim), 255, np.uint8)
    q[:, dim - near:] = 128 + rng.integers(0, 2, size=(nq, near))
    t[:, dim - near:] = 128 + rng.integers(0, 3, size=(nt, near))
    return q, t


@pytest.mark.parametrize("nq,nt,dim", [(300, 5000, 128), (70, 700, 128), (129, 2100, 64)])
def test_l2_u8_sqrtf_collisions_follow_the_float_ranking(gpu_ready, nq, nt, dim):
    """d^2 in [2^22, 8.3e6]: distinct integer d^2 round to the same float32 distance and OpenCV's rule (compare
    the float32 values, lower train index first) picks a different neighbour than ranking on the integers.  The
    kernel ranks on integers and re-ranks exactly these queries on the float32 value (k_knn2_u8_rerank); both
    oracles (NumPy, C) rank on the float32 value.  Train sets span several LDS chunks and splits."""
    from oracle import matcher_oracle as mo, ba_c
    q, t = far_apart_sets(nq, nt, dim, seed=nt)
    ref = mo.knn2(q, t, "l2")
    assert (ref[2].astype(np.float64) ** 2).min() >= 2 ** 22 or dim < 128     # dim <= 64: d^2 <= 64 * 255^2 < 2^22, no collisions exist
    got = gpu_knn2(q, t, "l2")
    assert_knn_equal(got, ref)
    assert_knn_equal(got, ba_c.knn2_u8(q, t))
    # the case is adversarial for an integer ranking: it would pick other neighbours for some queries
    ai, bi = q.astype(np.int64), t.astype(np.int64)
    d2 = (ai * ai).sum(1)[:, None] + (bi * bi).sum(1)[None, :] - 2 * (ai @ bi.T)
    int_best = np.argmin(d2, axis=1)
    if dim == 128:
        assert np.any(int_best != ref[0]), "no float32 collision decided a neighbour: the case does not test the rule"


def test_l2_u8_mixed_near_and_far_rows(gpu_ready):
    """Ordinary SIFT-like queries and far-apart ones in one call: only the far ones take the re-rank path."""
    from oracle import matcher_oracle as mo
    from sfm_amd import synth
    d1, d2 = synth.make_descriptors(400, 3000, seed=77)
    q = d1.astype(np.uint8); t = d2.astype(np.uint8)
    q[::8] = 0                                     # all-zero queries: nearest neighbours are ordinary rows (d^2 ~ 2^18)
    q[5:55] = 255                                  # all-255 queries: every SIFT-like train row is > 2^22 away
    q[5:55, :8] = np.random.default_rng(9).integers(250, 256, size=(50, 8))
    assert_knn_equal(gpu_knn2(q, t, "l2"), mo.knn2(q, t, "l2"))


def test_full_size_50k_all_rows(gpu_ready):
    """BASELINE config 2 (50k x 50k x 128): EVERY query row bit-exact against the C oracle (same float32 ranking
    rule as the NumPy oracle, tests/test_matcher_oracle.py checks the two against each other), and the match
    list is query-sorted with in-range indices."""
    import torch
    from oracle import ba_c, matcher_oracle as mo
    from sfm_amd import synth, matcher
    d1, d2 = synth.make_descriptors(50000, 50000, seed=1002)
    u1, u2 = d1.astype(np.uint8), d2.astype(np.uint8)
    i1, i2, a, b = matcher.knn2(torch.from_numpy(u1).cuda(), torch.from_numpy(u2).cuda(), "l2")
    got = (i1.cpu().numpy(), i2.cpu().numpy(), a.cpu().numpy(), b.cpu().numpy())
    assert_knn_equal(got, ba_c.knn2_u8(u1, u2))
    rows = np.random.default_rng(0).choice(50000, size=256, replace=False)
    assert_knn_equal(tuple(g[rows] for g in got), mo.knn2(u1[rows], u2, "l2"))
    q, t, d = matcher.ratio_filter(i1, a, b, 0.75)
    q, t = q.cpu().numpy(), t.cpu().numpy()
    assert np.all(np.diff(q) > 0) and t.min() >= 0 and t.max() < 50000
    keep = got[2].astype(np.float64) < 0.75 * got[3].astype(np.float64)
    assert np.array_equal(q, np.nonzero(keep)[0])
    rq, rt, rd = mo.ratio_filter(got[0], got[2], got[3], 0.75)
    assert np.array_equal(t, rt) and np.array_equal(d.cpu().numpy(), rd)


@pytest.mark.gpu
@pytest.mark.parametrize("qb", [None, "4", "2"])
@pytest.mark.parametrize("nq,nt,kind", [(12288, 2049, "sift"), (16400, 1017, "uniform"), (20001, 3, "sift"),
                                        (16390, 4101, "far"), (17000, 6000, "dups"), (28672, 300, "sift"), (29001, 2049, "uniform")])
def test_l2_u8_many_queries_kernel_edges(gpu_ready, monkeypatch, nq, nt, kind, qb):
    """One dim-128 pair with many queries: the distances come from the LDS-free kernel (k_knn2_u8_direct: train set
    re-tiled into MFMA operand order, candidate filter, ranking pipelined across tiles) - with four query blocks per
    wavefront from 28,672 queries on, two below (the planner's measured crossover; SFM_MATCH_QB forces either so that both
    instantiations meet every edge).  Against the C
    oracle, every row: a query count that is not a multiple of the 512-query workgroup, train sets that end inside a
    32-row tile / a 256-row window / a split, three train rows only, rows far enough apart for the float32 re-ranking
    (d^2 >= 2^22), and many exact duplicates (ties on d^2: lowest train index first)."""
    import torch
    from oracle import ba_c
    from sfm_amd import synth, matcher
    if qb is not None:
        monkeypatch.setenv("SFM_MATCH_QB", qb)
    rng = np.random.default_rng(nq + nt)
    if kind == "sift":
        d1, d2 = synth.make_descriptors(nq, nt, seed=nq)
        u1, u2 = d1.astype(np.uint8), d2.astype(np.uint8)
    elif kind == "uniform":
        u1 = rng.integers(0, 256, size=(nq, 128), dtype=np.uint8)
        u2 = rng.integers(0, 256, size=(nt, 128), dtype=np.uint8)
    elif kind == "far":
        u1, u2 = far_apart_sets(nq, nt, 128, seed=5)
    else:
        base = rng.integers(0, 256, size=(40, 128), dtype=np.uint8)
        u2 = base[rng.integers(0, 40, size=nt)]                     # every train row exists ~150 times
        u1 = base[rng.integers(0, 40, size=nq)]
        u1[::3] = np.clip(u1[::3].astype(np.int32) + rng.integers(-1, 2, size=u1[::3].shape), 0, 255).astype(np.uint8)
    i1, i2, a, b = matcher.knn2(torch.from_numpy(u1).cuda(), torch.from_numpy(u2).cuda(), "l2")
    got = (i1.cpu().numpy(), i2.cpu().numpy(), a.cpu().numpy(), b.cpu().numpy())
    assert_knn_equal(got, ba_c.knn2_u8(u1, u2))


def test_l2_u8_randomised_sizes_and_data(gpu_ready):
    """A fixed-seed sweep over both uint8 distance kernels and both filter settings (tools/stress_matcher.py runs the
    long version): random query / train counts on either side of the kernel switches (12,288 and 28,672 queries) and of the filter
    switch (2,048 train rows), SIFT-like, uniform and duplicate-heavy rows, every row against the C oracle."""
    import torch
    from oracle import ba_c
    from sfm_amd import synth, matcher
    rng = np.random.default_rng(20260104)
    for case in range(10):
        nq = int(rng.integers(12288, 24000)) if case % 2 == 0 else int(rng.integers(1, 12288))
        if case in (4, 8):
            nq = int(rng.integers(28672, 33000))           # four query blocks per wavefront
        nt = int(rng.choice([rng.integers(2, 2048), rng.integers(2048, 12000)]))
        kind = case % 3
        if kind == 0:
            d1, d2 = synth.make_descriptors(nq, nt, seed=case)
            u1, u2 = d1.astype(np.uint8), d2.astype(np.uint8)
        elif kind == 1:
            u1 = rng.integers(0, 256, size=(nq, 128), dtype=np.uint8)
            u2 = rng.integers(0, 256, size=(nt, 128), dtype=np.uint8)
        else:
            base = rng.integers(0, 256, size=(30, 128), dtype=np.uint8)
            u2 = base[rng.integers(0, 30, size=nt)]
            u1 = base[rng.integers(0, 30, size=nq)]
        i1, i2, a, b = matcher.knn2(torch.from_numpy(u1).cuda(), torch.from_numpy(u2).cuda(), "l2")
        got = (i1.cpu().numpy(), i2.cpu().numpy(), a.cpu().numpy(), b.cpu().numpy())
        ref = ba_c.knn2_u8(u1, u2)
        for g, r in zip(got, ref):
            assert np.array_equal(g, r), (case, nq, nt, kind)


# ------------------------------------------------------------------ batched (segmented) matching
def _image_sets(sizes, seed, kind="sift", dim=128):
    from sfm_amd import synth
    rng = np.random.default_rng(seed)
    if kind == "orb":
        return [rng.integers(0, 256, size=(n, 32), dtype=np.uint8) for n in sizes]
    base, _ = synth.make_descriptors(max(sizes) + 50, 2, seed=seed, dim=dim)
    sets = []
    for n in sizes:      # images share descriptors (noisy copies) so that the ratio test keeps a good fraction
        rows = base[rng.permutation(base.shape[0])[:n]] + np.rint(rng.normal(0, 5.0, size=(n, dim))).astype(np.float32)
        sets.append(np.clip(rows, 0, 255).astype(np.float32))
    return sets


@pytest.mark.parametrize("kind,as_u8", [("sift", False), ("sift", True), ("orb", True)])
def test_batched_pairs_equal_per_pair_calls(gpu_ready, kind, as_u8):
    """All image pairs of a preprocessing step in one launch (find_matches.py:329-350 loops over them serially):
    bit-identical to one match_features call per pair - ragged image sizes incl. one above a split boundary,
    an empty image, the same image on both sides, SIFT-like (float32 and uint8) and ORB (Hamming) descriptors."""
    from sfm_amd.matcher import ImageMatcher, match_arrays, match_pairs
    sizes = [300, 511, 2, 257, 0, 4500, 128, 1000]
    sets = _image_sets(sizes, seed=5, kind=kind)
    if as_u8:
        sets = [s.astype(np.uint8) for s in sets]
    pairs = [(i, j) for i in range(len(sizes)) for j in range(i + 1, len(sizes))] + [(3, 3), (5, 0), (6, 1)]
    got = match_pairs(sets, pairs)
    assert len(got) == len(pairs)
    n_matches = 0
    for (i, j), (q, t, d) in zip(pairs, got):
        rq, rt, rd = match_arrays(sets[i], sets[j]) if sizes[i] and sizes[j] else (np.zeros(0, np.int32),) * 2 + (np.zeros(0, np.float32),)
        assert np.array_equal(q, rq) and np.array_equal(t, rt) and np.array_equal(d, rd), (i, j)
        n_matches += len(q)
    assert n_matches > 200
    ms = ImageMatcher().match_features_batched(sets, pairs[:3])
    assert [[(m.queryIdx, m.trainIdx) for m in l] for l in ms] == [list(zip(q.tolist(), t.tolist())) for q, t, _ in got[:3]]


def test_batched_pairs_vs_oracle_and_degenerate_sizes(gpu_ready):
    from oracle import matcher_oracle as mo
    from sfm_amd.matcher import match_pairs
    sets = _image_sets([200, 333, 1, 640], seed=11)
    got = match_pairs(sets, [(0, 1), (1, 3), (3, 0)])
    for (i, j), (q, t, d) in zip([(0, 1), (1, 3), (3, 0)], got):
        rq, rt, rd = mo.match_features(sets[i], sets[j], 0.75, "l2")
        assert np.array_equal(q, rq) and np.array_equal(t, rt) and np.array_equal(d, rd)
    # train image with ONE descriptor (find_matches.py:151): THAT pair carries the ValueError, the others their matches -
    # the reference's per-pair try / except (:344-350) skips only the failing pair
    got2 = match_pairs(sets, [(0, 1), (0, 2), (1, 3)])
    assert isinstance(got2[1], ValueError) and "not enough values to unpack" in str(got2[1])
    assert all(np.array_equal(a, b) for a, b in zip(got2[0], got[0])) and all(np.array_equal(a, b) for a, b in zip(got2[2], got[1]))
    from sfm_amd.matcher import ImageMatcher
    ms = ImageMatcher().match_features_batched(sets, [(0, 1), (0, 2)])
    assert ms[1] is None and len(ms[0]) == len(got[0][0])
    # an image without keypoints (cv2 returns None) is an empty set; an image no pair refers to may be anything
    got3 = match_pairs([sets[0], None, sets[1], np.zeros((5, 7), np.float32)], [(0, 1), (1, 0), (0, 2)])
    assert len(got3[0][0]) == 0 and len(got3[1][0]) == 0 and all(np.array_equal(a, b) for a, b in zip(got3[2], got[0]))
    # one image with non-integral floats: only the pairs touching it leave the exact uint8 path, every pair still equals
    # its own match_arrays call bit for bit
    from sfm_amd.matcher import match_arrays
    frac = sets[3].astype(np.float32) + np.float32(0.25)
    mixed = [sets[0], sets[1], frac]
    prs = [(0, 1), (1, 2), (2, 0), (1, 0)]
    for (i, j), (q, t, d) in zip(prs, match_pairs(mixed, prs)):
        rq, rt, rd = match_arrays(mixed[i], mixed[j])
        assert np.array_equal(q, rq) and np.array_equal(t, rt) and np.array_equal(d, rd), (i, j)
    far_q, far_t = far_apart_sets(40, 600, 128, seed=3)                        # re-rank path inside a batch
    ordinary = _image_sets([300], seed=2)[0].astype(np.uint8)
    got = match_pairs([far_q, far_t, ordinary], [(2, 2), (0, 1), (2, 1)], ratio=2.0)     # ratio 2: every query row is kept
    ref = mo.knn2(far_q, far_t, "l2")
    assert np.array_equal(got[1][1], ref[0]) and np.array_equal(got[1][2], ref[2]) and len(got[1][0]) == 40
