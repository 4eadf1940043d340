"""Site-pattern compression on the device (mi_site_pattern_compress):
bit-exact against the fixtures dumped with the reference's own SitePattern code
(tests/golden/*.struct.json) and against the CPU oracle on large random alignments --
pattern matrix, weights AND the order (the iteration order of the reference's
unordered_map)."""
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

CODE = {c: i for i, c in enumerate("ACGT")}


def _codes(rows):
    return np.array([[CODE.get(ch.upper(), 4) for ch in row] for row in rows], dtype=np.int8)


@pytest.mark.parametrize("name", ["hello", "ds1_sub10", "flua"])
def test_reference_alignments_bit_exact(name):
    import libsbn_amd as L
    st = O.load_struct(name)
    seqs = O.read_fasta(os.path.join(O.DATA, st["source"]["fasta"]))
    rows = [seqs[t] for t in st["taxon_names"]]
    pats, w, _ = L.site_pattern_compress_device(_codes(rows))
    want_p = np.array(st["patterns"], dtype=np.int32)
    want_w = np.array(st["weights"], dtype=np.float64)
    assert pats.shape == want_p.shape
    assert np.array_equal(pats, want_p)  # same distinct columns in the same order
    assert np.array_equal(w, want_w)


@pytest.mark.parametrize("n,L,alphabet", [(1, 1, 4), (3, 7, 2), (40, 20000, 2), (64, 50000, 5),
                                          (5, 300000, 3)])
def test_random_alignments_match_the_oracle(n, L, alphabet):
    """Heavy duplication (small alphabets) and many distinct columns; the oracle's
    restatement of libstdc++'s unordered_map is itself pinned by the fixtures above."""
    import libsbn_amd as L_
    rng = np.random.default_rng(n * 1000 + L)
    codes = rng.integers(0, alphabet, size=(n, L)).astype(np.int8)
    if L > 100:  # blocks of repeated columns, so that weights > 1 occur at all sizes
        codes[:, L // 2:] = codes[:, :L - L // 2]
    rows = ["".join("ACGT-"[c] for c in row) for row in codes]
    want_p, want_w = O.site_pattern_compress(rows)
    pats, w, ms = L_.site_pattern_compress_device(codes)
    assert pats.shape == want_p.shape
    assert np.array_equal(pats, want_p)
    assert np.array_equal(w, want_w)
    assert w.sum() == L and ms >= 0.0
