"""ctypes front end for oracle/liboracle.so (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(REPO, "oracle")
GOLDEN = os.path.join(REPO, "tests", "golden")
DATA = os.path.join(GOLDEN, "data")

SUBST = {"JC69": 0, "GTR": 1, "reversible": 2, "WAG": 2}
SITE = {"constant": 0, "weibull": 1}
CLOCK = {"none": 0, "strict": 1}


class Spec(C.Structure):
    _fields_ = [(k, C.c_int32) for k in (
        "taxon_count", "pattern_count", "state_count", "category_count",
        "subst_model", "site_model", "clock_model", "use_tip_states")]


def _host_stamp():
    """liboracle.so is built -march=native; rebuild when the host CPU changes (the
    snapshot travels from the build container to the GPU box with the .so in it)."""
    import hashlib
    try:
        with open("/proc/cpuinfo") as fh:
            txt = fh.read()
        keep = [ln for ln in txt.splitlines() if ln.startswith(("model name", "flags"))][:2]
    except OSError:
        keep = []
    return hashlib.sha1("\n".join(keep).encode()).hexdigest()


_variant = "f64"
_libs = {}


def select(variant):
    """'f64' (default; the reference's precision) or 'ld' (same code, x87 long double)."""
    global _variant
    assert variant in ("f64", "ld")
    _variant = variant


def set_transition_mode(mode):
    """0: BEAGLE's exp form (default); 1: expm1 form. Applies to the selected variant."""
    lib().orc_set_transition_mode(int(mode))


def set_generic_states(on):
    """1: s == 4 also runs through the oracle's s-generic loops (the code the 20-state
    parity tests rely on)."""
    lib().orc_set_generic_states(int(on))


def set_reversible_model(exchangeabilities, freqs):
    """Table of the 'reversible' substitution model (process-wide, per oracle variant)."""
    ex, fr = f64(exchangeabilities), f64(freqs)
    s = len(fr)
    assert len(ex) == s * (s - 1) // 2
    for variant in ("f64", "ld"):
        keep = _variant
        select(variant)
        rc = lib().orc_set_reversible_model(s, _p(ex, C.c_double), _p(fr, C.c_double))
        select(keep)
        assert rc == 0


def lib():
    if _variant in _libs:
        return _libs[_variant]
    name = "liboracle.so" if _variant == "f64" else "liboracle_ld.so"
    so = os.path.join(ORACLE_DIR, name)
    stamp_path = os.path.join(ORACLE_DIR, "." + name + ".stamp")
    stamp = _host_stamp()
    # (several processes may come here at once -- the stress sweeps run in parallel on a fresh
    # box: the check and the rebuild happen under a file lock, so that nobody loads a library
    # that another process is still writing)
    import fcntl
    with open(os.path.join(ORACLE_DIR, "." + name + ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        have = open(stamp_path).read() if os.path.exists(stamp_path) else ""
        src_newer = (not os.path.exists(so) or os.path.getmtime(so) < max(
            os.path.getmtime(os.path.join(ORACLE_DIR, f))
            for f in ("phylo_oracle.c", "phylo_oracle.h")))
        if src_newer or have != stamp:
            subprocess.run(["make", "-B", "-C", ORACLE_DIR, name], check=True, capture_output=True)
            with open(stamp_path, "w") as fh:
                fh.write(stamp)
        L = C.CDLL(so)
    L.orc_last_error.restype = C.c_char_p
    L.orc_core_log_likelihood.restype = C.c_double
    L.orc_core_branch_gradient.restype = C.c_double
    _libs[_variant] = L
    return L


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def parse_site_spec(site):
    """'constant' -> (0, 1); 'weibull+K' -> (1, K) (site_model.cpp:10-25)."""
    if site == "constant":
        return 0, 1
    if site.startswith("weibull"):
        k = int(site.split("+")[1]) if "+" in site else 4
        return 1, k
    raise ValueError("Site model not known: " + site)


def make_spec(n, P, subst="JC69", site="constant", clock="strict", s=4, use_tip_states=1):
    site_kind, K = parse_site_spec(site)
    return Spec(n, P, s, K, SUBST[subst], site_kind, CLOCK[clock], use_tip_states)


def param_count(spec):
    return lib().orc_param_count(C.byref(spec))


def param_layout(spec):
    a, b, c, d = (C.c_int(), C.c_int(), C.c_int(), C.c_int())
    lib().orc_param_layout(C.byref(spec), C.byref(a), C.byref(b), C.byref(c), C.byref(d))
    return {"GTR rates": a.value, "frequencies": b.value, "Weibull shape": c.value,
            "clock rate": d.value}


def load_struct(name):
    with open(os.path.join(GOLDEN, name + ".struct.json")) as fh:
        return json.load(fh)


def load_kats():
    with open(os.path.join(GOLDEN, "reference_kats.json")) as fh:
        return json.load(fh)


def read_fasta(path):
    """alignment.cpp:40-72 semantics: name = whole header line after '>'."""
    seqs, name, cur = {}, None, []
    with open(path) as fh:
        for line in fh.read().splitlines():
            if not line:
                continue
            if line[0] == ">":
                if name:
                    seqs[name] = "".join(cur)
                name, cur = line[1:], []
            else:
                cur.append(line)
    if name:
        seqs[name] = "".join(cur)
    return seqs


def site_pattern_compress(seq_rows):
    """seq_rows: list of equal-length strings, row i = taxon id i."""
    n, L = len(seq_rows), len(seq_rows[0])
    buf = "".join(seq_rows).encode()
    pats = np.zeros((n, L), dtype=np.int32)
    w = np.zeros(L, dtype=np.float64)
    Pn = C.c_int32()
    rc = lib().orc_site_pattern_compress(n, L, buf, _p(pats, C.c_int32), _p(w, C.c_double),
                                         C.byref(Pn))
    if rc:
        raise RuntimeError(lib().orc_last_error().decode())
    return pats[:, :Pn.value].copy(), w[:Pn.value].copy()


def detrifurcate(n, parent_ids, bl):
    c0 = np.zeros(n - 1, np.int32)
    c1 = np.zeros(n - 1, np.int32)
    out = np.zeros(2 * n - 1)
    pi, b = i32(parent_ids), f64(bl)
    rc = lib().orc_detrifurcate(n, _p(pi, C.c_int32), _p(b, C.c_double), _p(c0, C.c_int32),
                                _p(c1, C.c_int32), _p(out, C.c_double))
    if rc:
        raise RuntimeError(lib().orc_last_error().decode())
    return c0, c1, out


def traversal_triples(n, c0, c1):
    post = np.zeros(3 * (n - 1), np.int32)
    pre = np.zeros(3 * (2 * n - 2), np.int32)
    lib().orc_postorder_triples(n, _p(i32(c0), C.c_int32), _p(i32(c1), C.c_int32),
                                _p(post, C.c_int32))
    lib().orc_preorder_triples(n, _p(i32(c0), C.c_int32), _p(i32(c1), C.c_int32),
                               _p(pre, C.c_int32))
    return post, pre


def weibull_rates(K, shape):
    r, w, d = np.zeros(K), np.zeros(K), np.zeros(K)
    lib().orc_weibull_rates(K, C.c_double(shape), _p(r, C.c_double), _p(w, C.c_double),
                            _p(d, C.c_double))
    return r, w, d


def stick_breaking(y):
    y = f64(y)
    x = np.zeros(len(y) + 1)
    lib().orc_stick_breaking(len(x), _p(y, C.c_double), _p(x, C.c_double))
    return x


def stick_breaking_inverse(x):
    x = f64(x)
    y = np.zeros(len(x) - 1)
    lib().orc_stick_breaking_inverse(len(x), _p(x, C.c_double), _p(y, C.c_double))
    return y


class Model(C.Structure):
    _fields_ = [("s", C.c_int), ("K", C.c_int), ("pi", C.c_double * 20),
                ("Q", C.c_double * 400), ("V", C.c_double * 400), ("Vinv", C.c_double * 400),
                ("lam", C.c_double * 20), ("gtr_rates", C.c_double * 190),
                ("n_gtr_rates", C.c_int), ("cat_rates", C.c_double * 64),
                ("cat_weights", C.c_double * 64), ("cat_rate_derivs", C.c_double * 64)]


def model_set(spec, params):
    m = Model()
    pr = f64(params)
    rc = lib().orc_model_set(C.byref(spec), _p(pr, C.c_double), C.byref(m))
    if rc:
        raise RuntimeError(lib().orc_last_error().decode())
    return m


def _tips_w(tips, weights):
    return i32(tips), f64(weights)


def unrooted_log_likelihoods(spec, tips, weights, parent_ids, bl, params, rescaling=False,
                             nthreads=1):
    tips, weights = _tips_w(tips, weights)
    pid, b, pr = i32(parent_ids), f64(bl), f64(params)
    T = pid.shape[0]
    out = np.zeros(T)
    rc = lib().orc_unrooted_log_likelihoods(
        C.byref(spec), _p(tips, C.c_int32), _p(weights, C.c_double), T, _p(pid, C.c_int32),
        _p(b, C.c_double), _p(pr, C.c_double), int(rescaling), nthreads, _p(out, C.c_double))
    if rc:
        raise RuntimeError(lib().orc_last_error().decode())
    return out


def unrooted_gradients(spec, tips, weights, parent_ids, bl, params, rescaling=False,
                       nthreads=1):
    tips, weights = _tips_w(tips, weights)
    pid, b, pr = i32(parent_ids), f64(bl), f64(params)
    T, n = pid.shape[0], spec.taxon_count
    ll = np.zeros(T)
    g = np.zeros((T, 2 * n - 1))
    site = np.zeros(T)
    subst = np.zeros((T, 8))
    rc = lib().orc_unrooted_gradients(
        C.byref(spec), _p(tips, C.c_int32), _p(weights, C.c_double), T, _p(pid, C.c_int32),
        _p(b, C.c_double), _p(pr, C.c_double), int(rescaling), nthreads, _p(ll, C.c_double),
        _p(g, C.c_double), _p(site, C.c_double), _p(subst, C.c_double))
    if rc:
        raise RuntimeError(lib().orc_last_error().decode())
    out = {"log_likelihood": ll, "branch_lengths": g}
    if spec.category_count > 1:
        out["site_model"] = site
    if spec.subst_model == SUBST["GTR"]:
        out["substitution_model"] = subst
    return out


def time_tree_init(n, parent_ids, bl, tip_dates):
    N = 2 * n - 1
    h, bd, ra = np.zeros(N), np.zeros(N), np.zeros(n - 1)
    pid, b, d = i32(parent_ids), f64(bl), f64(tip_dates)
    rc = lib().orc_time_tree_init(n, _p(pid, C.c_int32), _p(b, C.c_double), _p(d, C.c_double),
                                  _p(h, C.c_double), _p(bd, C.c_double), _p(ra, C.c_double))
    if rc:
        raise RuntimeError(lib().orc_last_error().decode())
    return h, bd, ra


def rooted_log_likelihoods(spec, tips, weights, parent_ids, bl, params, rates, heights, bounds,
                           with_jacobian=True, rescaling=False, nthreads=1):
    tips, weights = _tips_w(tips, weights)
    pid, b, pr = i32(parent_ids), f64(bl), f64(params)
    r, h, bd = f64(rates), f64(heights), f64(bounds)
    T = pid.shape[0]
    out = np.zeros(T)
    rc = lib().orc_rooted_log_likelihoods(
        C.byref(spec), _p(tips, C.c_int32), _p(weights, C.c_double), T, _p(pid, C.c_int32),
        _p(b, C.c_double), _p(pr, C.c_double), _p(r, C.c_double), _p(h, C.c_double),
        _p(bd, C.c_double), int(with_jacobian), int(rescaling), nthreads, _p(out, C.c_double))
    if rc:
        raise RuntimeError(lib().orc_last_error().decode())
    return out


def rooted_gradients(spec, tips, weights, parent_ids, bl, params, rates, rate_counts, heights,
                     bounds, ratios, rescaling=False, nthreads=1):
    tips, weights = _tips_w(tips, weights)
    pid, b, pr = i32(parent_ids), f64(bl), f64(params)
    r, h, bd, ra = f64(rates), f64(heights), f64(bounds), f64(ratios)
    rcnt = i32(rate_counts)
    T, n = pid.shape[0], spec.taxon_count
    ll = np.zeros(T)
    gr = np.zeros((T, n - 1))
    gc = np.zeros((T, 2 * n - 2))
    site = np.zeros(T)
    subst = np.zeros((T, 8))
    rc = lib().orc_rooted_gradients(
        C.byref(spec), _p(tips, C.c_int32), _p(weights, C.c_double), T, _p(pid, C.c_int32),
        _p(b, C.c_double), _p(pr, C.c_double), _p(r, C.c_double), _p(rcnt, C.c_int32),
        _p(h, C.c_double), _p(bd, C.c_double), _p(ra, C.c_double), int(rescaling), nthreads,
        _p(ll, C.c_double), _p(gr, C.c_double), _p(gc, C.c_double), _p(site, C.c_double),
        _p(subst, C.c_double))
    if rc:
        raise RuntimeError(lib().orc_last_error().decode())
    out = {"log_likelihood": ll, "ratios_root_height": gr, "clock_model": gc}
    if spec.category_count > 1:
        out["site_model"] = site
    if spec.subst_model == SUBST["GTR"]:
        out["substitution_model"] = subst
    return out


def parse_dates_from_names(names):
    """taxon_name_munging.cpp:46-78: trailing _<number>, then max - date."""
    import re
    rx = re.compile(r"^.+_(\d*\.?\d+(?:[eE][-+]?\d+)?)$")
    dates = []
    for nm in names:
        m = rx.match(nm)
        if not m:
            raise RuntimeError("Couldn't parse a date from:" + nm)
        dates.append(float(m.group(1)))
    mx = max(dates)
    return np.array([mx - d for d in dates])


def struct_arrays(st):
    """Flat arrays of a *.struct.json fixture."""
    tips = np.array(st["patterns"], dtype=np.int32)
    weights = np.array(st["weights"], dtype=np.float64)
    pids = np.array([t["parent_ids"] for t in st["trees"]], dtype=np.int32)
    bls = np.array([t["branch_lengths"] for t in st["trees"]], dtype=np.float64)
    return tips, weights, pids, bls


def unrooted_by_pattern_blocks(spec, tips, weights, parent_ids, bl, params, rescaling=True,
                               gradient=True, blocks=None, threads=None):
    """Every output of the unrooted path is a sum over site patterns (rescaling is per
    pattern): the oracle evaluated on `blocks` disjoint pattern blocks -- one call per block,
    on a thread each (ctypes releases the GIL; the oracle's process-wide settings are only
    read) -- and the per-block results added in block order.  This is how a full-size
    configs[4] evaluation (512 taxa x 50 000 patterns x 4 categories: ~0.8 TFLOP for one
    gradient) finishes in tens of seconds on the host cores."""
    from concurrent.futures import ThreadPoolExecutor
    tips, weights = _tips_w(tips, weights)
    P = tips.shape[1]
    threads = threads or min(32, os.cpu_count() or 1)
    blocks = blocks or max(1, min(P // 64, 4 * threads))
    cuts = [P * i // blocks for i in range(blocks + 1)]
    lib()  # load (and, if need be, build) once, before the threads start

    def one(i):
        lo, hi = cuts[i], cuts[i + 1]
        sub = Spec(spec.taxon_count, hi - lo, spec.state_count, spec.category_count,
                   spec.subst_model, spec.site_model, spec.clock_model, spec.use_tip_states)
        t, w = np.ascontiguousarray(tips[:, lo:hi]), np.ascontiguousarray(weights[lo:hi])
        if gradient:
            return unrooted_gradients(sub, t, w, parent_ids, bl, params, rescaling, 1)
        return {"log_likelihood": unrooted_log_likelihoods(sub, t, w, parent_ids, bl, params,
                                                           rescaling, 1)}

    with ThreadPoolExecutor(threads) as pool:
        parts = list(pool.map(one, range(blocks)))
    out = {k: np.zeros_like(v) for k, v in parts[0].items()}
    for part in parts:
        for k, v in part.items():
            out[k] += v
    return out
