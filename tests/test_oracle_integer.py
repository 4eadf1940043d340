"""Oracle, integer half: bit-exact against fixtures produced by the reference's
own structural code (oracle/_ref/ref_dump -> tests/golden/*.struct.json)."""
import os

import numpy as np
import pytest

import oracle_lib as O

CASES = ["hello", "hello_out", "five_taxon", "ds1_sub10", "ds1_top100", "flua"]


@pytest.mark.parametrize("name", CASES)
def test_site_pattern_compress_bit_exact(name):
    st = O.load_struct(name)
    seqs = O.read_fasta(os.path.join(O.DATA, st["source"]["fasta"]))
    rows = [seqs[nm] for nm in st["taxon_names"]]
    pats, w = O.site_pattern_compress(rows)
    assert pats.shape == (st["taxon_count"], st["pattern_count"])
    assert np.array_equal(pats, np.array(st["patterns"], dtype=np.int32))
    assert np.array_equal(w, np.array(st["weights"]))
    assert w.sum() == st["site_count"]


def test_site_pattern_unknown_symbol():
    with pytest.raises(RuntimeError, match="Symbol 'Z' not known"):
        O.site_pattern_compress(["ACGT", "ACZT"])


def test_site_pattern_symbol_table():
    # site_pattern.hpp:62-66: "-tgcaTGCA?" -> 4,3,2,1,0,3,2,1,0,4
    pats, w = O.site_pattern_compress(["-tgcaTGCA?"])
    # all columns distinct except the gap code which repeats
    assert sorted(pats[0].tolist()) == [0, 1, 2, 3, 4]
    assert w.sum() == 10


@pytest.mark.parametrize("name", ["hello", "five_taxon", "ds1_sub10", "ds1_top100"])
def test_detrifurcate_and_traversals(name):
    st = O.load_struct(name)
    n = st["taxon_count"]
    for tr in st["trees"]:
        c0, c1, bl = O.detrifurcate(n, tr["parent_ids"], tr["branch_lengths"])
        assert np.array_equal(bl, np.array(tr["bif_branch_lengths"]))
        # parent ids of the bifurcating tree
        par = np.full(2 * n - 2, -1)
        for i in range(n - 1):
            par[c0[i]] = n + i
            par[c1[i]] = n + i
        assert par.tolist() == tr["bif_parent_ids"]
        post, pre = O.traversal_triples(n, c0, c1)
        assert post.tolist() == tr["postorder_triples"]
        assert pre.tolist() == tr["preorder_triples"]


def test_rooted_traversals_flua():
    st = O.load_struct("flua")
    n = st["taxon_count"]
    tr = st["trees"][0]
    post = np.array(tr["postorder_triples"]).reshape(-1, 3)
    c0, c1 = post[:, 1].astype(np.int32), post[:, 2].astype(np.int32)
    assert post[:, 0].tolist() == list(range(n, 2 * n - 1))
    _, pre = O.traversal_triples(n, c0, c1)
    assert pre.tolist() == tr["preorder_triples"]
    # children order rule (node.cpp:32-59) reproduced from the parent-id vector alone
    h, bd, ra = O.time_tree_init(n, tr["parent_ids"], tr["branch_lengths"],
                                 O.parse_dates_from_names(st["taxon_names"]))
    assert h.shape == (2 * n - 1,)


def test_non_trifurcating_rejected():
    with pytest.raises(RuntimeError, match="non-trifurcating"):
        O.detrifurcate(4, [4, 4, 4, 5, 5], [0.1] * 6)
