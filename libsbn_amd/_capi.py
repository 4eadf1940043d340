"""ctypes binding of include/mi_phylo.h (the C ABI of libmi_phylo.so).

There is no CPU fallback: if the HIP library is missing the import fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MI_PHYLO_LIBRARY: another build of the same library (kernel A/B runs: tools/ab_kernels.py)
LIB_PATH = os.environ.get("MI_PHYLO_LIBRARY") or os.path.join(_HERE, "libmi_phylo.so")

I32P = C.POINTER(C.c_int32)
F64P = C.POINTER(C.c_double)


class EngineSpec(C.Structure):
    """mi_engine_spec (include/mi_phylo.h)."""
    _fields_ = [(k, C.c_int32) for k in (
        "taxon_count", "pattern_count", "state_count", "category_count", "subst_model",
        "site_model", "clock_model", "use_tip_states", "device", "reserved")]


# Every symbol include/mi_phylo.h declares: (restype, argtypes)
_V = C.c_void_p
SYMBOLS = {
    "mi_abi_version": (C.c_int32, []),
    "mi_last_error": (C.c_char_p, []),
    "mi_device_count": (C.c_int32, []),
    "mi_engine_create_sharded":
        (C.c_int32, [C.POINTER(EngineSpec), C.c_int32, I32P, C.c_int32, _V, _V, _V, _V, _V,
                     C.POINTER(_V)]),
    "mi_engine_shard_count": (C.c_int32, [_V]),
    "mi_engine_shard_device": (C.c_int32, [_V, C.c_int32]),
    "mi_shard_range": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, I32P, I32P]),
    "mi_engine_gradients_unrooted_reduced":
        (C.c_int32, [_V, C.c_int32, _V, _V, _V, C.c_int32, _V, _V, C.c_int32, _V, _V, _V]),
    "mi_engine_gradients_unrooted_reduced_device":
        (C.c_int32, [_V, _V, C.c_int32, _V, _V, _V, C.c_int32, _V, _V, C.c_int32, _V, _V, _V]),
    "mi_engine_create": (C.c_int32, [C.POINTER(EngineSpec), _V, _V, _V, C.POINTER(_V)]),
    "mi_engine_create_reversible":
        (C.c_int32, [C.POINTER(EngineSpec), _V, _V, _V, _V, _V, C.POINTER(_V)]),
    "mi_wag_model": (C.c_int32, [F64P, F64P]),
    "mi_engine_destroy": (None, [_V]),
    "mi_engine_param_count": (C.c_int32, [_V]),
    "mi_engine_block_count": (C.c_int32, [_V]),
    "mi_engine_block": (C.c_int32, [_V, C.c_int32, C.POINTER(C.c_char_p), I32P, I32P]),
    "mi_engine_log_likelihoods_unrooted": (C.c_int32, [_V, C.c_int32, _V, _V, _V, C.c_int32, _V]),
    "mi_engine_gradients_unrooted":
        (C.c_int32, [_V, C.c_int32, _V, _V, _V, C.c_int32, _V, _V, _V, _V]),
    "mi_engine_log_likelihoods_rooted":
        (C.c_int32, [_V, C.c_int32, _V, _V, _V, _V, _V, _V, C.c_int32, C.c_int32, _V]),
    "mi_engine_gradients_rooted":
        (C.c_int32, [_V, C.c_int32, _V, _V, _V, _V, _V, _V, _V, _V, C.c_int32, _V, _V, _V, _V,
                     _V]),
    "mi_engine_log_likelihoods_unrooted_device":
        (C.c_int32, [_V, _V, C.c_int32, _V, _V, _V, C.c_int32, _V]),
    "mi_engine_gradients_unrooted_device":
        (C.c_int32, [_V, _V, C.c_int32, _V, _V, _V, C.c_int32, _V, _V, _V, _V]),
    "mi_engine_log_likelihoods_rooted_device":
        (C.c_int32, [_V, _V, C.c_int32, _V, _V, _V, _V, _V, _V, C.c_int32, C.c_int32, _V]),
    "mi_engine_gradients_rooted_device":
        (C.c_int32, [_V, _V, C.c_int32, _V, _V, _V, _V, _V, _V, _V, _V, C.c_int32, _V, _V, _V,
                     _V, _V]),
    "mi_engine_reserve": (C.c_int32, [_V, C.c_int32, C.c_int32]),
    "mi_engine_reserve_reduced": (C.c_int32, [_V, C.c_int32, C.c_int32]),
    "mi_engine_check_status": (C.c_int32, [_V, _V]),
    "mi_engine_profile_begin": (C.c_int32, [_V, C.c_int32]),
    "mi_engine_profile_collect": (C.c_int32, [_V, F64P, C.c_int32, I32P]),
    "mi_engine_profile_begin_phases": (C.c_int32, [_V, C.c_int32]),
    "mi_engine_profile_collect_phases": (C.c_int32, [_V, F64P, F64P, C.c_int32, I32P, I32P]),
    "mi_engine_last_call_info":
        (C.c_int32, [_V, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "mi_engine_last_call_path": (C.c_char_p, [_V]),
    "mi_engine_last_call_launches": (C.c_int32, [_V, I32P, I32P]),
    "mi_site_pattern_compress":
        (C.c_int32, [C.c_int32, C.c_int32, C.c_int64, _V, I32P, _V, _V, F64P]),
    "mi_site_pattern_compress_device":
        (C.c_int32, [C.c_int32, C.c_int32, C.c_int64, _V, I32P, C.POINTER(_V), C.POINTER(_V), F64P]),
    "mi_device_free": (None, [_V]),
    "mi_engine_create_device_tips": (C.c_int32, [C.POINTER(EngineSpec), _V, _V, C.POINTER(_V)]),
}

_lib = None


def load():
    """Load libmi_phylo.so and bind every declared symbol (raises if absent)."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: torch ships its own libamdhip64, libmi_phylo.so is linked
    # against /opt/rocm's.  Whichever is loaded FIRST serves both (same SONAME); loaded the
    # other way round, the second runtime finds no device ("No HIP GPUs are available" from
    # torch, or "no HIP device available" from mi_engine_create).  The Python mirror is used
    # next to torch (device tensors, torch.distributed), so when torch is installed it is
    # loaded first; MI_PHYLO_NO_TORCH_PRELOAD=1 skips that for torch-free use.
    import sys
    if "torch" not in sys.modules and os.environ.get("MI_PHYLO_NO_TORCH_PRELOAD") != "1":
        import importlib.util
        if importlib.util.find_spec("torch") is not None:
            import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950). libsbn_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    return load().mi_last_error().decode()
