"""ctypes binding of libmi_phylo_host.so (FASTA / site patterns / Newick / Nexus /
time trees -- plain C++17, runs without a GPU)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmi_phylo_host.so")
_V = C.c_void_p
_lib = None

SYMBOLS = {
    "mih_last_error": (C.c_char_p, []),
    "mih_parse_newick_file": (_V, [C.c_char_p]),
    "mih_parse_nexus_file": (_V, [C.c_char_p]),
    "mih_parse_newick_string": (_V, [C.c_char_p]),
    "mih_trees_free": (None, [_V]),
    "mih_tree_count": (C.c_int32, [_V]),
    "mih_taxon_count": (C.c_int32, [_V]),
    "mih_taxon_name": (C.c_char_p, [_V, C.c_int32]),
    "mih_node_count": (C.c_int32, [_V, C.c_int32]),
    "mih_copy_tree": (C.c_int32, [_V, C.c_int32, _V, _V]),
    "mih_site_pattern_from_fasta": (_V, [C.c_char_p, _V]),
    "mih_site_pattern_from_fasta_protein": (_V, [C.c_char_p, _V]),
    "mih_site_pattern_free": (None, [_V]),
    "mih_pattern_count": (C.c_int32, [_V]),
    "mih_site_count": (C.c_int32, [_V]),
    "mih_sequence_count": (C.c_int32, [_V]),
    "mih_copy_site_pattern": (C.c_int32, [_V, _V, _V]),
    "mih_dates_from_taxon_names": (C.c_int32, [_V, _V]),
    "mih_time_tree_from_branch_lengths": (C.c_int32, [C.c_int32, _V, _V, _V, _V, _V, _V]),
    "mih_time_tree_from_height_ratios": (C.c_int32, [C.c_int32, _V, _V, _V, _V, _V, _V]),
}


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not found: run __graft_entry__.build()")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def _err():
    return RuntimeError(load().mih_last_error().decode())


def _ptr(a):
    return C.c_void_p(a.ctypes.data)


class TreeCollection:
    """Flat view of a parsed tree file: taxon names by leaf id, and per tree the
    parent-id vector + branch lengths in the reference's node-id convention."""

    def __init__(self, handle):
        lib = load()
        if not handle:
            raise _err()
        try:
            self.taxon_names = [lib.mih_taxon_name(handle, i).decode()
                                for i in range(lib.mih_taxon_count(handle))]
            self.parent_ids, self.branch_lengths = [], []
            for t in range(lib.mih_tree_count(handle)):
                nodes = lib.mih_node_count(handle, t)
                pid = np.empty(nodes - 1, np.int32)
                bl = np.empty(nodes, np.float64)
                lib.mih_copy_tree(handle, t, _ptr(pid), _ptr(bl))
                self.parent_ids.append(pid)
                self.branch_lengths.append(bl)
            self._handle = handle
        except Exception:
            lib.mih_trees_free(handle)
            raise

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h:
            load().mih_trees_free(h)
            self._handle = None

    def tree_count(self):
        return len(self.parent_ids)

    def taxon_count(self):
        return len(self.taxon_names)

    @staticmethod
    def of_newick_file(path):
        return TreeCollection(load().mih_parse_newick_file(path.encode()))

    @staticmethod
    def of_nexus_file(path):
        return TreeCollection(load().mih_parse_nexus_file(path.encode()))

    @staticmethod
    def of_newick_string(newick):
        return TreeCollection(load().mih_parse_newick_string(newick.encode()))

    def site_pattern(self, fasta_path, protein=False):
        """SitePattern(alignment, tag_taxon_map): (patterns [n][P] int32, weights [P], sites).
        protein: amino-acid alphabet ARNDCQEGHILKMFPSTWYV -> 0..19, gaps / ambiguity -> 20."""
        lib = load()
        make = lib.mih_site_pattern_from_fasta_protein if protein else lib.mih_site_pattern_from_fasta
        h = make(fasta_path.encode(), self._handle)
        if not h:
            raise _err()
        try:
            n, P = lib.mih_sequence_count(h), lib.mih_pattern_count(h)
            pats = np.empty((n, P), np.int32)
            w = np.empty(P, np.float64)
            lib.mih_copy_site_pattern(h, _ptr(pats), _ptr(w))
            return pats, w, lib.mih_site_count(h)
        finally:
            lib.mih_site_pattern_free(h)

    def dates_from_taxon_names(self):
        out = np.empty(len(self.taxon_names))
        if load().mih_dates_from_taxon_names(self._handle, _ptr(out)):
            raise _err()
        return out


def time_tree_from_branch_lengths(parent_ids, branch_lengths, tip_dates):
    pid = np.ascontiguousarray(parent_ids, np.int32)
    bl = np.ascontiguousarray(branch_lengths, np.float64)
    d = np.ascontiguousarray(tip_dates, np.float64)
    n = len(d)
    h, b, r = np.empty(2 * n - 1), np.empty(2 * n - 1), np.empty(n - 1)
    if load().mih_time_tree_from_branch_lengths(n, _ptr(pid), _ptr(bl), _ptr(d), _ptr(h),
                                                _ptr(b), _ptr(r)):
        raise _err()
    return h, b, r


def time_tree_from_height_ratios(parent_ids, tip_dates, height_ratios):
    pid = np.ascontiguousarray(parent_ids, np.int32)
    d = np.ascontiguousarray(tip_dates, np.float64)
    ra = np.ascontiguousarray(height_ratios, np.float64)
    n = len(d)
    bl, h, b = np.empty(2 * n - 1), np.empty(2 * n - 1), np.empty(2 * n - 1)
    if load().mih_time_tree_from_height_ratios(n, _ptr(pid), _ptr(d), _ptr(ra), _ptr(bl),
                                               _ptr(h), _ptr(b)):
        raise _err()
    return bl, h, b
