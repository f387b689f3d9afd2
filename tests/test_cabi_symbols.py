"""The C-ABI library loads and exports every symbol include/gamd_hip.h declares (no compute
calls: there is no GPU here)."""
import os
import re
import ctypes

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "gamd_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gamd_[a-z_0-9]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    from gamd_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.load()


def test_header_symbols_exported(lib):
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/gamd_hip.h but not exported"


def test_binding_table_matches_header(lib):
    from gamd_amd import _lib
    assert sorted(_lib.SYMBOLS) == _declared()


def test_version_and_error_strings(lib):
    assert lib.gamd_version().startswith(b"gamd_hip")
    assert isinstance(lib.gamd_last_error(), bytes)


def test_struct_layout_matches_header():
    from gamd_amd._lib import GamdConfig, GamdMdParams, GamdNhcParams
    assert ctypes.sizeof(GamdConfig) == 96 and GamdConfig.n_boxes.offset == 88 and GamdConfig.neighbor_skin.offset == 72 and GamdConfig.edge_capacity.offset == 40 and GamdConfig.edge_dtype.offset == 52
    assert GamdConfig.self_loop_mode.offset == 76 and GamdConfig.kernel_select.offset == 80 and GamdConfig.small_tile_limit.offset == 84
    assert GamdConfig.encoding_size.offset == 56 and GamdConfig.no_expand_edge.offset == 68
    assert ctypes.sizeof(GamdMdParams) == 56 and GamdMdParams.seed.offset == 16 and GamdMdParams.mass_h_amu.offset == 32
    assert ctypes.sizeof(GamdNhcParams) == 64 and GamdNhcParams.ndf.offset == 32 and GamdNhcParams.mass_h_amu.offset == 40


def test_create_rejects_bad_arguments_without_gpu(lib):
    from gamd_amd._lib import GamdConfig
    h = ctypes.c_void_p()
    c = GamdConfig()
    c.n_atoms, c.n_layers, c.cutoff = 0, 4, 1.0
    assert lib.gamd_create(ctypes.byref(c), ctypes.byref(h)) < 0
    assert b"n_atoms" in lib.gamd_last_error()
