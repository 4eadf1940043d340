"""The built-in 20-state table (libsbn_amd/csrc/aa_tables.h, handed out by mi_wag_model)
against a second transcription kept as a fixture in PAML's wag.dat layout
(tests/golden/wag_paml.dat), plus the invariants a mistyped entry would break.

No GPU needed: mi_wag_model is host code of libmi_phylo.so.  The reference has no 20-state
model (substitution_model.cpp:6-15); this pins the table the 20-state parity tests hand to
BOTH the engine and the oracle, which by themselves would not notice a wrong digit.
"""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ORDER = "ARNDCQEGHILKMFPSTWYV"


def _paml_table():
    """Own parser of the wag.dat layout: 19 rows of the lower triangle (row i has i entries,
    i = 1..19), then 20 frequencies; text after them is commentary."""
    rows, freqs, letters = [], [], None
    with open(os.path.join(GOLDEN, "wag_paml.dat")) as fh:
        for line in fh:
            tok = line.split()
            if not tok:
                continue
            try:
                vals = [float(t) for t in tok]
            except ValueError:
                if len(tok) == 20 and all(len(t) == 1 for t in tok):
                    letters = "".join(tok)
                if len(rows) == 19 and len(freqs) == 20:
                    continue
                raise
            if len(rows) < 19:
                assert len(vals) == len(rows) + 1, f"row {len(rows) + 1} has {len(vals)} entries"
                rows.append(vals)
            elif len(freqs) < 20:
                freqs.extend(vals)
    assert len(rows) == 19 and len(freqs) == 20
    S = np.zeros((20, 20))
    for i, r in enumerate(rows, start=1):
        for j, v in enumerate(r):
            S[i, j] = S[j, i] = v
    return S, np.array(freqs), letters


def _engine_table():
    import libsbn_amd.engine as E
    ex, fr = E.wag_model()  # upper triangle row by row (the reference's GTR-rate order)
    S = np.zeros((20, 20))
    k = 0
    for i in range(20):
        for j in range(i + 1, 20):
            S[i, j] = S[j, i] = ex[k]
            k += 1
    assert k == 190
    return S, fr


def test_builtin_wag_equals_second_transcription():
    S2, f2, letters = _paml_table()
    S1, f1 = _engine_table()
    assert letters == ORDER
    assert np.array_equal(S1, S2)  # digit for digit: both are decimal literals
    # mi_wag_model normalises the frequencies to sum to exactly 1; the file holds the
    # published 7-digit values
    assert abs(f2.sum() - 1.0) < 5e-7
    assert np.allclose(f1, f2 / f2.sum(), rtol=0, atol=1e-15)
    assert abs(f1.sum() - 1.0) < 1e-15


def test_wag_invariants():
    S, f = _engine_table()
    ix = {a: i for i, a in enumerate(ORDER)}
    assert np.array_equal(S, S.T) and np.all(np.diag(S) == 0)
    off = S[np.triu_indices(20, 1)]
    assert np.all(off > 0) and len(np.unique(off)) == 190  # no entry typed twice
    # what is known about WAG beyond its digits: the largest exchangeabilities are the
    # conservative pairs, in this order; the smallest is Cys-Glu
    pairs = sorted(((S[i, j], ORDER[i] + ORDER[j]) for i in range(20) for j in range(i + 1, 20)),
                   reverse=True)
    assert [p for _, p in pairs[:6]] == ["IV", "FY", "DE", "QE", "ND", "RK"]
    assert pairs[-1][1] == "CE" and pairs[-1][0] == 0.021352
    assert S[ix["I"], ix["V"]] == 7.8213 and S[ix["F"], ix["Y"]] == 6.45428
    # frequencies: Ala and Leu the most frequent, Trp the rarest
    assert ORDER[int(np.argmax(f))] == "A" and ORDER[int(np.argsort(f)[-2])] == "L"
    assert ORDER[int(np.argmin(f))] == "W"
    # the rate matrix built by the reference's GTR recipe (substitution_model.cpp:39-80) is
    # reversible with these frequencies: detailed balance and zero row sums
    Q = S * f[None, :]
    np.fill_diagonal(Q, -Q.sum(axis=1))
    Q /= -(f * np.diag(Q)).sum()
    assert np.allclose(f[:, None] * Q, (f[:, None] * Q).T, rtol=0, atol=1e-17)
    assert np.allclose(f @ Q, 0, atol=1e-16)
    assert abs(-(f * np.diag(Q)).sum() - 1.0) < 1e-14
