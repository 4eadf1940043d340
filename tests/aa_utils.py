"""Inputs of the 20-state parity tests (S-WAG of SURVEY.md section 8d, scaled down)."""
import numpy as np

import oracle_lib as O
import tree_utils as TU


def random_aa_alignment(n, P, rng, gap_fraction=0.05):
    tips = rng.integers(0, 20, size=(n, P)).astype(np.int32)
    tips[rng.random((n, P)) < gap_fraction] = 20
    weights = rng.integers(1, 6, size=P).astype(np.float64)
    return tips, weights


def random_reversible_model(rng):
    ex = rng.gamma(1.0, 1.0, size=190) + 1e-3
    fr = rng.dirichlet(5 * np.ones(20))
    return ex, fr


def params_for(site, T, rng):
    """Parameter rows [Weibull shape?][clock rate] of a model without substitution block."""
    if site == "constant":
        return np.ones((T, 1))
    pr = np.ones((T, 2))
    pr[:, 0] = rng.uniform(0.3, 2.0, size=T)
    return pr


def oracle_spec(n, P, site, clock="strict"):
    return O.make_spec(n, P, "reversible", site, clock, s=20)
