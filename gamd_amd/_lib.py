"""ctypes binding of libgamd_hip.so (C ABI: include/gamd_hip.h).

There is deliberately no fallback: if the HIP library is missing or no GPU is
present, loading / gamd_create fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GAMD_LIB: load another build of the SAME C ABI instead (tools/ point it at libgamd_hip_prof.so, the -DGAMD_PROFILING
# build with instrumented kernel variants).  It must exist: there is no fallback either way.
LIB_PATH = os.environ.get("GAMD_LIB") or os.path.join(_HERE, "libgamd_hip.so")


class GamdConfig(C.Structure):
    _fields_ = [("n_atoms", C.c_int32), ("kind", C.c_int32), ("n_layers", C.c_int32),
                ("use_bond", C.c_int32), ("nbr_flavour", C.c_int32), ("device", C.c_int32),
                ("cutoff", C.c_float), ("box", C.c_float * 3), ("edge_capacity", C.c_int64),
                ("keep_stages", C.c_int32), ("edge_dtype", C.c_int32),
                ("encoding_size", C.c_int32), ("edge_embedding_dim", C.c_int32), ("hidden_dim", C.c_int32),
                ("no_expand_edge", C.c_int32), ("neighbor_skin", C.c_float), ("self_loop_mode", C.c_int32),
                ("kernel_select", C.c_int32), ("small_tile_limit", C.c_int32), ("n_boxes", C.c_int32)]


# common tail of both integrator parameter blocks (masses per species, length unit, rigid water)
_MD_EXT = [("mass_h_amu", C.c_float), ("length_per_nm", C.c_float), ("rigid_water", C.c_int32),
           ("r_oh", C.c_float), ("r_hh", C.c_float), ("remove_cm_motion", C.c_int32)]


class GamdNhcParams(C.Structure):
    _fields_ = [("dt_ps", C.c_float), ("mass_amu", C.c_float), ("temperature_k", C.c_float),
                ("frequency_per_ps", C.c_float), ("chain_length", C.c_int32), ("num_mts", C.c_int32),
                ("num_yoshidasuzuki", C.c_int32), ("reset", C.c_int32), ("ndf", C.c_double)] + _MD_EXT


class GamdMdParams(C.Structure):
    _fields_ = [("dt_ps", C.c_float), ("mass_amu", C.c_float), ("temperature_k", C.c_float),
                ("gamma_per_ps", C.c_float), ("seed", C.c_uint64), ("first_step", C.c_uint64)] + _MD_EXT


# every symbol include/gamd_hip.h declares: name -> (restype, argtypes)
_vp, _i32, _i64 = C.c_void_p, C.c_int32, C.c_int64
SYMBOLS = {
    "gamd_version": (C.c_char_p, []),
    "gamd_last_error": (C.c_char_p, []),
    "gamd_create": (_i32, [C.POINTER(GamdConfig), C.POINTER(_vp)]),
    "gamd_destroy": (_i32, [_vp]),
    "gamd_load_weight": (_i32, [_vp, C.c_char_p, _vp, C.POINTER(_i64), _i32]),
    "gamd_finalize_weights": (_i32, [_vp]),
    "gamd_set_scaler": (_i32, [_vp, C.c_double, C.c_double]),
    "gamd_set_bonds": (_i32, [_vp, _vp, _i64]),
    "gamd_forces_async": (_i32, [_vp, _vp, _vp, C.POINTER(C.c_float), _vp, _vp, _vp]),
    "gamd_set_node_features": (_i32, [_vp, _vp]),
    "gamd_get_device_flags": (_i32, [_vp, C.POINTER(_i32)]),
    "gamd_sync_status": (_i32, [_vp, _vp]),
    "gamd_forces_host": (_i32, [_vp, _vp, _vp, C.POINTER(C.c_float), _vp, _i32, _vp]),
    "gamd_forces": (_i32, [_vp, _vp, _vp, C.POINTER(C.c_float), _vp, _vp, _vp]),
    "gamd_forces_edges": (_i32, [_vp, _vp, _vp, C.POINTER(C.c_float), _vp, _vp, _i64, _vp, _vp, _vp]),
    "gamd_build_neighbors": (_i32, [_vp, _vp, _vp, C.POINTER(C.c_float), _vp]),
    "gamd_get_counts": (_i32, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "gamd_get_skin_stats": (_i32, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "gamd_debug_get": (_i32, [_vp, _i32, _vp, C.c_size_t]),
    "gamd_md_run": (_i32, [_vp, _vp, _vp, _vp, _vp, C.POINTER(C.c_float), C.POINTER(GamdMdParams), _i64, _vp]),
    "gamd_md_run_nhc": (_i32, [_vp, _vp, _vp, _vp, _vp, C.POINTER(C.c_float), C.POINTER(GamdNhcParams), _vp, _i64, _vp]),
    "gamd_profile": (_i32, [_vp, _vp, _vp, C.POINTER(C.c_float), _vp, _vp, C.c_char_p, C.c_size_t,
                            C.POINTER(C.c_float), _i32, C.POINTER(_i32)]),
    "gamd_timing_enable": (_i32, [_vp, _i32]),
    "gamd_timing_read": (_i32, [_vp, _vp, C.POINTER(C.c_double), C.POINTER(_i64)]),
    "gamd_timing_read_stages": (_i32, [_vp, _vp, C.POINTER(C.c_double), C.POINTER(_i64)]),
    "gamd_timing_read_steps": (_i32, [_vp, _vp, C.POINTER(C.c_float), _i64, C.POINTER(_i64)]),
}

_lib = None


def load() -> C.CDLL:
    """Load the shared library (once) and type every entry point."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension was not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C gamd_amd/csrc`). "
            "gamd_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class GamdError(RuntimeError):
    pass


def check(status: int, what: str) -> int:
    if status < 0:
        msg = load().gamd_last_error().decode("utf-8", "replace")
        raise GamdError(f"{what} failed with status {status}: {msg}")
    return status
