"""CPU tests of the matcher oracle: known answers and the output contract visible in the reference's
shipped match files (tests/golden/bunny_matches.npz, extracted from bunny_data/matches/*.npz)."""
import os

import numpy as np

from conftest import GOLDEN


def test_known_answers_ties_and_ratio():
    from oracle import matcher_oracle as mo
    q = np.zeros((2, 128), np.float32)
    t = np.zeros((4, 128), np.float32)
    t[0, 0] = 5; t[1, 0] = 3; t[2, 0] = 3; t[3, 0] = 4            # distances 5,3,3,4: tie -> lower index first
    i1, i2, d1, d2 = mo.knn2(q, t)
    assert (i1[0], i2[0], d1[0], d2[0]) == (1, 2, 3.0, 3.0)
    assert len(mo.match_features(q, t)[0]) == 0                    # d1 == d2 fails the strict ratio test
    t[2, 0] = 4                                                    # 3 vs 4: 3 < 0.75*4 is False (boundary)
    assert len(mo.match_features(q, t)[0]) == 0
    t[1, 0] = 2
    qi, ti, d = mo.match_features(q, t)
    assert qi.tolist() == [0, 1] and ti.tolist() == [1, 1] and d.tolist() == [2.0, 2.0]
    # degenerate sizes as the reference: one train row -> the unpacking at find_matches.py:151 raises; none -> []
    import pytest
    with pytest.raises(ValueError, match="not enough values to unpack"):
        mo.match_features(q, t[:1])
    assert len(mo.match_features(q, t[:0])[0]) == 0 and len(mo.match_features(q[:0], t)[0]) == 0


def test_c_and_numpy_oracles_agree_incl_sqrtf_collisions():
    """The two restatements rank on the float32 distance (lower train index on ties): ordinary SIFT-like sets and
    far-apart sets (d^2 >= 2^22, where distinct integers share a float32 root and the rule decides neighbours)."""
    from oracle import matcher_oracle as mo, ba_c
    from sfm_amd import synth
    d1, d2 = synth.make_descriptors(300, 900, seed=2)
    u1, u2 = d1.astype(np.uint8), d2.astype(np.uint8)
    for a, b in zip(mo.knn2(u1, u2), ba_c.knn2_u8(u1, u2)):
        assert np.array_equal(a, b)
    rng = np.random.default_rng(4)
    q = np.zeros((300, 128), np.uint8); t = np.full((5000, 128), 255, np.uint8)
    q[:, 96:] = 128 + rng.integers(0, 2, size=(300, 32))          # all d^2 within ~100 of 96 * 255^2 = 6.2e6
    t[:, 96:] = 128 + rng.integers(0, 3, size=(5000, 32))
    ref = mo.knn2(q, t)
    for a, b in zip(ref, ba_c.knn2_u8(q, t)):
        assert np.array_equal(a, b)
    ai, bi = q.astype(np.int64), t.astype(np.int64)
    d2i = (ai * ai).sum(1)[:, None] + (bi * bi).sum(1)[None, :] - 2 * (ai @ bi.T)
    assert d2i.min() >= 2 ** 22 and np.any(np.argmin(d2i, axis=1) != ref[0])      # the rule is exercised


def test_integer_path_equals_float_path():
    from oracle import matcher_oracle as mo
    from sfm_amd import synth
    d1, d2 = synth.make_descriptors(200, 300, seed=1)
    a = mo.sq_l2(d1, d2)
    acc = np.zeros_like(a)
    for k in range(128):
        diff = d1[:, k, None] - d2[None, :, k]
        acc = acc + diff * diff
    assert np.array_equal(a, acc)                                  # exact in float32 in any order


def test_hamming_popcount():
    from oracle import matcher_oracle as mo
    a = np.array([[0xFF, 0x00] + [0] * 30], np.uint8)
    b = np.array([[0x0F, 0x01] + [0] * 30, [0xFF, 0x00] + [0] * 30], np.uint8)
    assert mo.hamming(a, b).tolist() == [[5.0, 0.0]]


def test_output_contract_of_the_shipped_match_files():
    """Every one of the reference's 148 match files is query-sorted with one match per query and
    distances that are exactly sqrtf(integer) - what an L2 matcher over uint8-valued descriptors emits."""
    m = np.load(os.path.join(GOLDEN, "bunny_matches.npz"))
    off = m["offsets"]
    assert len(off) - 1 == 148 and off[-1] == 10907
    d = m["distance"]
    d2 = np.rint(d ** 2)
    assert np.array_equal(np.sqrt(d2.astype(np.float32)).astype(np.float64), d)
    assert d2.min() >= 775 and d2.max() <= 137175
    for i in range(148):
        q = m["queryIdx"][off[i]:off[i + 1]]
        assert np.all(np.diff(q) > 0)
    # the oracle's output has the same shape of contract
    from oracle import matcher_oracle as mo
    from sfm_amd import synth
    d1, d2s = synth.make_descriptors(500, 500, seed=3)
    q, t, dist = mo.match_features(d1, d2s)
    assert np.all(np.diff(q) > 0) and len(np.unique(t)) <= len(t)
    assert np.array_equal(np.sqrt(np.rint(dist.astype(np.float64) ** 2).astype(np.float32)), dist)
