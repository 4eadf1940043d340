"""Host-side formats of the product (libmi_phylo_host.so: FASTA, site patterns,
Newick/Nexus ingest, node ids, time trees) -- bit-exact against fixtures produced
by the reference's own structural code (tests/golden/*.struct.json) and against
the known-answer values of src/rooted_tree.hpp:124-159."""
import os

import numpy as np
import pytest

import oracle_lib as O
from libsbn_amd import _hostapi as H

CASES = ["hello", "hello_out", "five_taxon", "ds1_sub10", "ds1_top100", "flua"]


def _collection(st):
    path = os.path.join(O.DATA, st["source"]["trees"])
    if st["source"]["format"] == "nexus":
        return H.TreeCollection.of_nexus_file(path)
    return H.TreeCollection.of_newick_file(path)


@pytest.mark.parametrize("name", CASES)
def test_tree_ingest_matches_reference(name):
    st = O.load_struct(name)
    tc = _collection(st)
    assert tc.taxon_names == st["taxon_names"]
    assert tc.tree_count() == len(st["trees"])
    for t, ref in enumerate(st["trees"]):
        assert tc.parent_ids[t].tolist() == ref["parent_ids"]
        assert tc.branch_lengths[t].tolist() == ref["branch_lengths"]  # bit-exact doubles


@pytest.mark.parametrize("name", CASES)
def test_site_pattern_matches_reference(name):
    st = O.load_struct(name)
    tc = _collection(st)
    pats, w, sites = tc.site_pattern(os.path.join(O.DATA, st["source"]["fasta"]))
    assert sites == st["site_count"]
    assert np.array_equal(pats, np.array(st["patterns"], dtype=np.int32))
    assert np.array_equal(w, np.array(st["weights"]))


def test_newick_string_and_errors():
    tc = H.TreeCollection.of_newick_string("(x:0,(a:1.1,(b:2,('quack 1':0.1,duck:0):0):0):0,c:3):1.1;")
    assert tc.taxon_names == ["x", "a", "b", "quack 1", "duck", "c"]
    assert tc.branch_lengths[0][-1] == 1.1  # root branch length kept at the root id
    with pytest.raises(RuntimeError, match="not known in our taxon set"):
        H.TreeCollection.of_newick_string("(a:1,b:1,c:1);\n(a:1,b:1,d:1);")
    with pytest.raises(RuntimeError, match="Float conversion failed"):
        H.TreeCollection.of_newick_string("(a:xyz,b:1,c:1);")
    with pytest.raises(RuntimeError, match="Cannot open"):
        H.TreeCollection.of_newick_file("/nonexistent.nwk")
    with pytest.raises(RuntimeError, match="begin with #NEXUS"):
        H.TreeCollection.of_nexus_file(os.path.join(O.DATA, "hello.nwk"))
    # metadata comments as in BEAST trees: [&rate=1.0] after labels / before lengths
    tc = H.TreeCollection.of_newick_string("((a[&x=1]:[&r=2]1.5,b:2)[&p=.9]:0.5,c:3);")
    assert tc.branch_lengths[0].tolist() == [1.5, 2.0, 3.0, 0.5, 0.0]


def test_site_pattern_errors(tmp_path):
    tc = H.TreeCollection.of_newick_string("(a:1,b:1,c:1);")
    bad = tmp_path / "bad.fasta"
    bad.write_text(">a\nACGT\n>b\nACZT\n>c\nACGT\n")
    with pytest.raises(RuntimeError, match="Symbol 'Z' not known"):
        tc.site_pattern(str(bad))
    ragged = tmp_path / "ragged.fasta"
    ragged.write_text(">a\nACGT\n>b\nACG\n>c\nACGT\n")
    with pytest.raises(RuntimeError, match="not all the same length"):
        tc.site_pattern(str(ragged))
    missing = tmp_path / "missing.fasta"
    missing.write_text(">a\nACGT\n>b\nACGT\n>d\nACGT\n")
    with pytest.raises(RuntimeError, match="Taxon 'c' not found"):
        tc.site_pattern(str(missing))


def test_rooted_tree_example_exact():
    k = O.load_kats()["rooted_tree_example"]
    h, b, r = H.time_tree_from_branch_lengths(k["parent_ids"], k["branch_lengths"],
                                              k["tip_dates"])
    assert h.tolist() == k["node_heights"]
    assert b.tolist() == k["node_bounds"]
    assert r.tolist() == [1. / 3.5, 1.5 / 4., 7.]
    # rooted_tree.hpp:141-158: root height doubled
    bl, h2, _ = H.time_tree_from_height_ratios(k["parent_ids"], k["tip_dates"],
                                               [1. / 3.5, 1.5 / 4., 14.])
    assert h2.tolist() == [5., 3., 0., 1., 2.75, 7.125, 14.]
    assert bl[:6].tolist() == [9., 4.125, 2.75, 1.75, 4.375, 6.875]
    bad = list(k["branch_lengths"])
    bad[0] += 0.01
    with pytest.raises(RuntimeError, match="time-calibrated"):
        H.time_tree_from_branch_lengths(k["parent_ids"], bad, k["tip_dates"])


def test_flua_dates_and_time_tree_match_oracle():
    st = O.load_struct("flua")
    tc = _collection(st)
    dates = tc.dates_from_taxon_names()
    assert np.array_equal(dates, O.parse_dates_from_names(st["taxon_names"]))
    assert dates.min() == 0.0
    h, b, r = H.time_tree_from_branch_lengths(tc.parent_ids[0], tc.branch_lengths[0], dates)
    oh, ob, orr = O.time_tree_init(st["taxon_count"], tc.parent_ids[0], tc.branch_lengths[0],
                                   dates)
    assert np.array_equal(h, oh) and np.array_equal(b, ob) and np.array_equal(r, orr)


def test_instance_rejects_wrong_rooting():
    import libsbn_amd as L
    inst = L.rooted_instance("charlie")
    with pytest.raises(RuntimeError, match="bifurcating at the root"):
        inst.read_newick_file(os.path.join(O.DATA, "hello.nwk"))
    inst = L.unrooted_instance("charlie")
    with pytest.raises(RuntimeError, match="trifucation"):
        inst.read_newick_file(os.path.join(O.DATA, "fluA.tree"))


def test_protein_alphabet(tmp_path):
    """The 20-state engine's alphabet (not in the reference, whose SymbolVectorOf is
    DNA-only: site_pattern.cpp:16-46): ARNDCQEGHILKMFPSTWYV -> 0..19, gaps / ambiguity -> 20,
    anything else an error; patterns are compressed by the same code as DNA."""
    from libsbn_amd import _hostapi as H
    fa = tmp_path / "p.fasta"
    fa.write_text(">a\nARNDCQEGHILKMFPSTWYV-AA\n>b\narndcqeghilkmfpstwyvXAA\n>c\nAAAAAAAAAAAAAAAAAAAA?AA\n")
    tc = H.TreeCollection.of_newick_string("(a,b,c);")
    pats, w, sites = tc.site_pattern(str(fa), protein=True)
    assert sites == 23 and pats.shape[0] == 3
    cols = {tuple(pats[:, j]): w[j] for j in range(pats.shape[1])}
    assert cols[(0, 0, 0)] == 3.0                      # A A A three times
    assert cols[(20, 20, 20)] == 1.0                   # - X ?
    for k in range(1, 20):
        assert cols[(k, k, 0)] == 1.0
    assert w.sum() == 23
    bad = tmp_path / "bad.fasta"
    bad.write_text(">a\nA1\n>b\nAA\n>c\nAA\n")
    import pytest
    with pytest.raises(RuntimeError, match="not known"):
        tc.site_pattern(str(bad), protein=True)
