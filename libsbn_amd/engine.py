"""Python mirror of libsbn's Engine (src/engine.hpp:26-54) on top of the C ABI.

Names follow the reference: PhyloModelSpecification(substitution, site, clock),
Engine.log_likelihoods / gradients return what Engine::LogLikelihoods /
Engine::Gradients return (a vector of doubles / one PhyloGradient per tree with
the gradient map keys "branch_lengths", "site_model", "substitution_model",
"ratios_root_height", "clock_model").  Errors surface as RuntimeError, as
pybind11 does for the reference's Failwith (src/sugar.hpp:67-78).
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _capi

SUBST = {"JC69": 0, "GTR": 1, "WAG": 2, "reversible": 2}
CLOCK = {"none": 0, "strict": 1}


@dataclass
class PhyloModelSpecification:
    """src/phylo_model.hpp:13-17."""
    substitution: str = "JC69"
    site: str = "constant"
    clock: str = "strict"


def wag_model():
    """The built-in WAG table as (exchangeabilities[190], frequencies[20]) -- upper triangle
    row by row, amino-acid order ARNDCQEGHILKMFPSTWYV (mi_wag_model)."""
    ex, fr = np.zeros(190), np.zeros(20)
    rc = _capi.load().mi_wag_model(ex.ctypes.data_as(_capi.F64P), fr.ctypes.data_as(_capi.F64P))
    if rc != 0:
        raise RuntimeError(_capi.last_error())
    return ex, fr


class PhyloGradient:
    """src/tree_gradient.hpp:10-19.  (A plain __slots__ class: a call returns one object
    per tree, and dataclass construction was most of the Python-side time of a call.)"""
    __slots__ = ("log_likelihood", "gradient")

    def __init__(self, log_likelihood, gradient):
        self.log_likelihood = log_likelihood
        self.gradient = gradient

    def __repr__(self):
        return f"PhyloGradient(log_likelihood={self.log_likelihood!r}, gradient={self.gradient!r})"


def _parse_site(site):
    # src/site_model.cpp:10-25
    if site == "constant":
        return 0, 1
    if site.startswith("weibull"):
        idx = site.find("+")
        return 1, (int(site[idx + 1:]) if idx >= 0 else 4)
    raise RuntimeError("Site model not known: " + site)


def _np(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _ptr(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


class Engine:
    """One MI355X engine: tips + pattern weights resident in HBM."""

    def __init__(self, model_specification, patterns, weights, use_tip_states=True,
                 device=-1, thread_count=1, tip_partials=None, reversible_model=None,
                 shard_devices=None, shard_mode="trees", device_tips=None):
        """patterns: [taxon][pattern] compact states (SitePattern::GetPatterns), or None when
        tip_partials ([taxon][pattern][s], SitePattern::GetPartials) is given with
        use_tip_states=False.  Substitution "WAG" / "reversible" makes a 20-state engine
        (s = 20); reversible_model = (exchangeabilities[190], frequencies[20]) replaces the
        built-in WAG table.  shard_devices (a list of HIP device ordinals, repeats allowed)
        makes ONE handle that drives several devices / logical shards
        (mi_engine_create_sharded): shard_mode "trees" deals the trees of a call in
        contiguous blocks, "patterns" gives every shard a block of site patterns and adds
        the per-tree results (unrooted calls).  Such a handle serves the numpy entry points;
        the *_device ones need one engine per device.  device_tips = (device pointer of the
        int32 [taxon][pattern] states, device pointer of the float64 weights, taxon count,
        pattern count) makes the engine from tips that are on the device already
        (mi_engine_create_device_tips; e.g. what site_pattern_compress_device(...,
        keep_on_device=True) hands out): patterns / weights are then None."""
        if thread_count == 0:  # src/engine.cpp:14-16
            raise RuntimeError("Thread count needs to be strictly positive.")
        self._lib = _capi.load()
        self._h = None
        if model_specification.substitution not in SUBST:
            raise RuntimeError("Substitution model not known: " +
                               model_specification.substitution)
        if model_specification.clock not in CLOCK:
            raise RuntimeError("Clock model not known: " + model_specification.clock)
        site_kind, K = _parse_site(model_specification.site)
        states = 20 if SUBST[model_specification.substitution] == 2 else 4
        self.state_count = states
        if device_tips is not None:
            # (this branch used to return before any validation and silently ignored
            # what does not combine with device-resident tips)
            for name, val in (("patterns", patterns), ("weights", weights), ("tip_partials", tip_partials),
                              ("reversible_model", reversible_model), ("shard_devices", shard_devices)):
                if val is not None:
                    raise RuntimeError(f"device_tips does not combine with {name}: the tips and weights "
                                       "are the device arrays, a 20-state engine made this way uses "
                                       "the built-in WAG table, and a sharded handle uploads its tips "
                                       "from host arrays")
            d_states, d_weights, n, P = device_tips
            if not d_states or not d_weights or int(n) < 3 or int(P) < 1:
                raise RuntimeError("device_tips = (states pointer, weights pointer, taxon count >= 3, "
                                   "pattern count >= 1)")
            self.taxon_count, self.pattern_count, self.category_count = n, P, K
            self.node_count = 2 * n - 1
            self.spec = _capi.EngineSpec(n, P, states, K, SUBST[model_specification.substitution],
                                         site_kind, CLOCK[model_specification.clock],
                                         1 if use_tip_states else 0, device, 0)
            h = C.c_void_p()
            self._check(self._lib.mi_engine_create_device_tips(
                C.byref(self.spec), C.c_void_p(d_states), C.c_void_p(d_weights), C.byref(h)))
            self._h = h
            self.param_count = self._lib.mi_engine_param_count(h)
            self.is_gtr = model_specification.substitution == "GTR"
            return
        weights = _np(weights, np.float64)
        if tip_partials is not None:
            tip_partials = _np(tip_partials, np.float64)
            n, P = tip_partials.shape[:2]
            if tip_partials.shape != (n, P, states):
                raise RuntimeError(f"tip partials must be [taxon][pattern][{states}]")
        if patterns is not None:
            patterns = _np(patterns, np.int32)
            n, P = patterns.shape
        if weights.shape != (P,):
            raise RuntimeError("pattern weights must have one entry per site pattern")
        self.taxon_count, self.pattern_count, self.category_count = n, P, K
        self.node_count = 2 * n - 1
        self.spec = _capi.EngineSpec(n, P, states, K, SUBST[model_specification.substitution],
                                     site_kind, CLOCK[model_specification.clock],
                                     1 if use_tip_states else 0, device, 0)
        h = C.c_void_p()
        ex = fr = None
        if reversible_model is not None:  # validated before ANY create call reads the arrays
            if states != 20:
                raise RuntimeError("reversible_model needs a 20-state substitution model")
            ex, fr = (_np(x, np.float64) for x in reversible_model)
            if ex.shape != (190,) or fr.shape != (20,):
                raise RuntimeError("reversible_model must be (exchangeabilities[190], "
                                   "frequencies[20])")
        if patterns is not None and patterns.size and int(patterns.max()) > states:
            # (compact states are 0..s-1, s = gap; a 20-state pattern matrix handed to a
            # 4-state engine -- or the reverse, caught below for DNA codes only by value --
            # would be read as another alphabet)
            raise RuntimeError(f"compact tip states exceed the {states}-state alphabet "
                               f"(largest code {int(patterns.max())})")
        if shard_devices is not None:
            devs = _np(shard_devices, np.int32)
            mode = {"trees": 0, "patterns": 1}[shard_mode]
            rc = self._lib.mi_engine_create_sharded(
                C.byref(self.spec), len(devs), devs.ctypes.data_as(_capi.I32P), mode, _ptr(ex),
                _ptr(fr), _ptr(patterns), _ptr(tip_partials), _ptr(weights), C.byref(h))
        elif states == 20:
            rc = self._lib.mi_engine_create_reversible(
                C.byref(self.spec), _ptr(ex), _ptr(fr), _ptr(patterns), _ptr(tip_partials),
                _ptr(weights), C.byref(h))
        else:
            rc = self._lib.mi_engine_create(C.byref(self.spec), _ptr(patterns),
                                            _ptr(tip_partials), _ptr(weights), C.byref(h))
        self._check(rc)
        self._h = h
        self.param_count = self._lib.mi_engine_param_count(h)
        self.is_gtr = model_specification.substitution == "GTR"

    def close(self):
        if self._h is not None:
            self._lib.mi_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(_capi.last_error())

    # Engine::GetPhyloModelBlockSpecification (src/engine.cpp:48-52)
    def block_specification(self):
        out = {}
        for i in range(self._lib.mi_engine_block_count(self._h)):
            name, start, length = C.c_char_p(), C.c_int32(), C.c_int32()
            self._check(self._lib.mi_engine_block(self._h, i, C.byref(name), C.byref(start),
                                                  C.byref(length)))
            out[name.value.decode()] = (start.value, length.value)
        return out

    def _params(self, params, T):
        pr = _np(params if params is not None else np.zeros((T, self.param_count)), np.float64)
        if pr.ndim == 1 and self.param_count:
            pr = pr.reshape(-1, self.param_count) if pr.size % self.param_count == 0 else pr
        if self.param_count == 0:
            pr = np.zeros((T, 0))
        if pr.shape != (T, self.param_count):
            # fat_beagle.hpp:138 / block_specification.cpp:72-81
            raise RuntimeError("We param_matrix needs as many rows as we have trees "
                               f"and {self.param_count} columns; got {pr.shape}.")
        return np.ascontiguousarray(pr)

    # ---- host-pointer path (numpy in, numpy out) -------------------------------
    def log_likelihoods(self, parent_ids, branch_lengths, params=None, rescaling=False):
        """Engine::LogLikelihoods(const UnrootedTreeCollection&, ...)."""
        n = self.taxon_count
        pid = _np(parent_ids, np.int32).reshape(-1, 2 * n - 3)
        T = pid.shape[0]
        if T == 0:  # FatBeagleParallelize over an empty collection returns an empty vector
            return np.empty(0)
        bl = _np(branch_lengths, np.float64).reshape(T, 2 * n - 2)
        pr = self._params(params, T)
        out = np.empty(T)
        self._check(self._lib.mi_engine_log_likelihoods_unrooted(
            self._h, T, _ptr(pid), _ptr(bl), _ptr(pr), int(rescaling), _ptr(out)))
        return out

    def gradients(self, parent_ids, branch_lengths, params=None, rescaling=False,
                  gradient_blocks=None):
        """Engine::Gradients(const UnrootedTreeCollection&, ...) -> [PhyloGradient].

        gradient_blocks (an extension): None = every block the reference returns; or a
        collection of block names the caller will read ("branch_lengths" always comes) --
        a caller that only uses the branch-length gradient of a GTR model (vip does) saves
        the 16 finite-difference passes behind "substitution_model" and the extra pass
        behind "site_model"; what is returned is bit-identical to the full call."""
        n, N = self.taxon_count, self.node_count
        pid = _np(parent_ids, np.int32).reshape(-1, 2 * n - 3)
        T = pid.shape[0]
        if T == 0:
            return []
        bl = _np(branch_lengths, np.float64).reshape(T, 2 * n - 2)
        pr = self._params(params, T)
        ll, g = np.empty(T), np.empty((T, N))
        want = None if gradient_blocks is None else set(gradient_blocks)
        site = np.empty(T) if want is None or "site_model" in want else None
        subst = np.empty((T, 8)) if want is None or "substitution_model" in want else None
        self._check(self._lib.mi_engine_gradients_unrooted(
            self._h, T, _ptr(pid), _ptr(bl), _ptr(pr), int(rescaling), _ptr(ll), _ptr(g),
            _ptr(site), _ptr(subst)))
        return self._phylo_gradients(ll, {"branch_lengths": g}, site, subst)

    def gradients_reduced(self, parent_ids, branch_lengths, params, branch_index, index_count,
                          tree_weights=None, rescaling=False):
        """One variational-inference step's reductions behind the gradient call
        (mi_engine_gradients_unrooted_reduced; vip/burrito.py:143-166,
        vip/branch_model.py:125-132): returns (sum_t w_t logL_t, sum_t w_t site gradient_t,
        index_gradient[index_count] = scatter-add of the branch gradients by branch_index
        (negative = skip), per-tree log-likelihoods)."""
        n, N = self.taxon_count, self.node_count
        pid = _np(parent_ids, np.int32).reshape(-1, 2 * n - 3)
        T = pid.shape[0]
        bl = _np(branch_lengths, np.float64).reshape(T, 2 * n - 2)
        pr = self._params(params, T)
        bi = _np(branch_index, np.int32).reshape(T, N)
        tw = None if tree_weights is None else _np(tree_weights, np.float64).reshape(T)
        sums, ig, ll = np.zeros(2), np.zeros(max(index_count, 1)), np.empty(T)
        self._check(self._lib.mi_engine_gradients_unrooted_reduced(
            self._h, T, _ptr(pid), _ptr(bl), _ptr(pr), int(rescaling), _ptr(bi), _ptr(tw),
            int(index_count), _ptr(sums), _ptr(ig), _ptr(ll)))
        return sums[0], sums[1], ig[:index_count], ll

    def _phylo_gradients(self, ll, blocks, site, subst):
        """Per-tree PhyloGradient objects over row views of the freshly allocated result
        arrays of one call (no per-tree copies: 1000 trees cost ~0.3 ms instead of ~1.1)."""
        if self.category_count > 1 and site is not None:
            blocks = dict(blocks, site_model=site.reshape(-1, 1))
        if self.is_gtr and subst is not None:
            blocks = dict(blocks, substitution_model=subst)
        names = list(blocks)
        rows = [list(blocks[k]) for k in names]  # lists of row views
        lls = ll.tolist()
        if len(names) == 1:
            (a,), (ra,) = names, rows
            return [PhyloGradient(l, {a: x}) for l, x in zip(lls, ra)]
        if len(names) == 2:
            (a, b), (ra, rb) = names, rows
            return [PhyloGradient(l, {a: x, b: y}) for l, x, y in zip(lls, ra, rb)]
        return [PhyloGradient(l, dict(zip(names, r))) for l, *r in zip(lls, *rows)]

    def rooted_log_likelihoods(self, parent_ids, branch_lengths, params=None, rates=None,
                               node_heights=None, node_bounds=None, rescaling=False,
                               with_jacobian=True):
        """Engine::LogLikelihoods(const RootedTreeCollection&) (with_jacobian=True) or
        Engine::UnrootedLogLikelihoods(const RootedTreeCollection&) (False)."""
        n, N = self.taxon_count, self.node_count
        pid = _np(parent_ids, np.int32).reshape(-1, N - 1)
        T = pid.shape[0]
        if T == 0:
            return np.empty(0)
        bl = _np(branch_lengths, np.float64).reshape(T, N)
        pr = self._params(params, T)
        r = None if rates is None else _np(rates, np.float64).reshape(T, N - 1)
        h = None if node_heights is None else _np(node_heights, np.float64).reshape(T, N)
        b = None if node_bounds is None else _np(node_bounds, np.float64).reshape(T, N)
        out = np.empty(T)
        self._check(self._lib.mi_engine_log_likelihoods_rooted(
            self._h, T, _ptr(pid), _ptr(bl), _ptr(pr), _ptr(r), _ptr(h), _ptr(b),
            int(with_jacobian), int(rescaling), _ptr(out)))
        return out

    def rooted_gradients(self, parent_ids, branch_lengths, params, rates, rate_counts,
                         node_heights, node_bounds, height_ratios, rescaling=False):
        """Engine::Gradients(const RootedTreeCollection&, ...) -> [PhyloGradient]."""
        n, N = self.taxon_count, self.node_count
        pid = _np(parent_ids, np.int32).reshape(-1, N - 1)
        T = pid.shape[0]
        if T == 0:
            return []
        bl = _np(branch_lengths, np.float64).reshape(T, N)
        pr = self._params(params, T)
        r = _np(rates, np.float64).reshape(T, N - 1)
        rc = _np(rate_counts, np.int32).reshape(T)
        h = _np(node_heights, np.float64).reshape(T, N)
        b = _np(node_bounds, np.float64).reshape(T, N)
        ra = _np(height_ratios, np.float64).reshape(T, n - 1)
        ll, gr, gc = np.empty(T), np.empty((T, n - 1)), np.empty((T, N - 1))
        site, subst = np.empty(T), np.empty((T, 8))
        self._check(self._lib.mi_engine_gradients_rooted(
            self._h, T, _ptr(pid), _ptr(bl), _ptr(pr), _ptr(r), _ptr(rc), _ptr(h), _ptr(b),
            _ptr(ra), int(rescaling), _ptr(ll), _ptr(gr), _ptr(gc), _ptr(site), _ptr(subst)))
        out = []
        for t in range(T):
            gm = {"ratios_root_height": gr[t].copy(),
                  "clock_model": gc[t, :1].copy() if rc[t] == 1 else gc[t].copy()}
            if self.category_count > 1:
                gm["site_model"] = site[t:t + 1].copy()
            if self.is_gtr:
                gm["substitution_model"] = subst[t].copy()
            out.append(PhyloGradient(float(ll[t]), gm))
        return out

    # ---- device-pointer path (raw pointers, e.g. torch tensors' data_ptr()) ------
    def reserve(self, tree_count, for_gradients):
        self._check(self._lib.mi_engine_reserve(self._h, tree_count, int(for_gradients)))

    def log_likelihoods_device(self, stream, T, parent_ids, branch_lengths, params, out_ll,
                               rescaling=False):
        self._check(self._lib.mi_engine_log_likelihoods_unrooted_device(
            self._h, stream, T, parent_ids, branch_lengths, params, int(rescaling), out_ll))

    def gradients_device(self, stream, T, parent_ids, branch_lengths, params, out_ll,
                         out_branch, out_site=None, out_subst=None, rescaling=False):
        self._check(self._lib.mi_engine_gradients_unrooted_device(
            self._h, stream, T, parent_ids, branch_lengths, params, int(rescaling), out_ll,
            out_branch, out_site, out_subst))

    def reserve_reduced(self, tree_count, index_count):
        """mi_engine_reserve_reduced: workspace of a fused-reduction call (graph capture)."""
        self._check(self._lib.mi_engine_reserve_reduced(self._h, int(tree_count), int(index_count)))

    def check_status(self, stream=None):
        self._check(self._lib.mi_engine_check_status(self._h, stream))

    def profile_begin(self, max_calls):
        self._check(self._lib.mi_engine_profile_begin(self._h, max_calls))

    def profile_collect(self, capacity):
        out = (C.c_double * capacity)()
        cnt = C.c_int32()
        self._check(self._lib.mi_engine_profile_collect(self._h, out, capacity, C.byref(cnt)))
        return [out[i] for i in range(cnt.value)]

    def profile_begin_phases(self, max_calls):
        self._check(self._lib.mi_engine_profile_begin_phases(self._h, max_calls))

    def profile_collect_phases(self, capacity):
        """(total ms per call, [set-up, post-order, pre-order / main walk, rest] ms per call,
        evaluations in the first walk launch) -- include/mi_phylo.h"""
        out = (C.c_double * capacity)()
        ph = (C.c_double * (4 * capacity))()
        cnt, first = C.c_int32(), C.c_int32()
        self._check(self._lib.mi_engine_profile_collect_phases(
            self._h, out, ph, capacity, C.byref(cnt), C.byref(first)))
        return ([out[i] for i in range(cnt.value)],
                [[ph[4 * i + k] for k in range(4)] for i in range(cnt.value)], first.value)

    def shard_devices(self):
        """HIP device ordinal of every shard (a single-device engine: its one device)."""
        k = self._lib.mi_engine_shard_count(self._h)
        return [self._lib.mi_engine_shard_device(self._h, i) for i in range(k)]

    def last_call_info(self):
        name, ev, gev = C.c_char_p(), C.c_int64(), C.c_int64()
        self._check(self._lib.mi_engine_last_call_info(self._h, C.byref(name), C.byref(ev),
                                                       C.byref(gev)))
        return name.value.decode(), ev.value, gev.value

    def last_call_path(self):
        """mi_engine_last_call_path: one line saying which kernels / stores the last call used."""
        return self._lib.mi_engine_last_call_path(self._h).decode()

    def last_call_launches(self):
        """(walk-kernel launches of the last call, arena-budget back-offs since creation)"""
        a, b = C.c_int32(), C.c_int32()
        self._check(self._lib.mi_engine_last_call_launches(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value


class DevicePatterns:
    """The compressed alignment left on the device (mi_site_pattern_compress_device): device
    pointers of the int32 [taxon][pattern] matrix and the float64 weights; `as_device_tips()`
    is what Engine(device_tips=...) takes; freed with mi_device_free when released."""

    def __init__(self, patterns_ptr, weights_ptr, taxon_count, pattern_count):
        self.patterns_ptr, self.weights_ptr = patterns_ptr, weights_ptr
        self.taxon_count, self.pattern_count = taxon_count, pattern_count

    def as_device_tips(self):
        return (self.patterns_ptr, self.weights_ptr, self.taxon_count, self.pattern_count)

    def release(self):
        lib = _capi.load()
        for ptr in (self.patterns_ptr, self.weights_ptr):
            if ptr:
                lib.mi_device_free(C.c_void_p(ptr))
        self.patterns_ptr = self.weights_ptr = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


def site_pattern_compress_device(codes, device=0, keep_on_device=False):
    """SitePattern::Compress on the GPU (include/mi_phylo.h: mi_site_pattern_compress).

    codes: [taxon][site] symbol codes 0..3, 4 = gap / ambiguous.  Returns (patterns
    [taxon][P] int32, weights [P] float64, milliseconds of the column-hashing kernel) in
    the reference's pattern order; keep_on_device=True: (DevicePatterns, milliseconds) -- the
    matrix and the weights stay in device memory (mi_site_pattern_compress_device)."""
    lib = _capi.load()
    c = _np(codes, np.int8)
    n, L = c.shape
    if keep_on_device:
        count, ms = C.c_int32(0), C.c_double(0.0)
        d_pat, d_w = C.c_void_p(), C.c_void_p()
        if lib.mi_site_pattern_compress_device(int(device), n, L, _ptr(c), C.byref(count),
                                               C.byref(d_pat), C.byref(d_w), C.byref(ms)):
            raise RuntimeError(_capi.last_error())
        return DevicePatterns(d_pat.value, d_w.value, n, count.value), ms.value
    patterns = np.empty((n, L), dtype=np.int32)
    weights = np.empty(L, dtype=np.float64)
    count = C.c_int32(0)
    ms = C.c_double(0.0)
    if lib.mi_site_pattern_compress(int(device), n, L, _ptr(c), C.byref(count), _ptr(patterns),
                                    _ptr(weights), C.byref(ms)):
        raise RuntimeError(_capi.last_error())
    P = count.value
    return patterns.reshape(-1)[:n * P].reshape(n, P).copy(), weights[:P].copy(), ms.value
