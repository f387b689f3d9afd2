"""No hot kernel of the shipped library may spill registers to scratch memory.  Parses the amdhsa
metadata of the gfx950 code objects embedded in gamd_amd/libgamd_hip.so (tools/kernel_resources.py: llvm-objdump --offloading
+ llvm-readelf --notes); CPU only, runs wherever the ROCm LLVM tools are installed."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_resources as kr  # noqa: E402

HOT = re.compile(r"k_conv_edge|k_edge_encode|k_node")


@pytest.fixture(scope="module")
def resources():
    from gamd_amd import _lib
    if not os.path.exists(os.path.join(kr.LLVM_BIN, "llvm-readelf")):
        pytest.skip("ROCm LLVM tools not installed")
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return kr.kernel_resources(_lib.LIB_PATH)


def test_every_kernel_family_is_present(resources):
    names = " ".join(resources)
    for k in ("k_conv_edge<", "k_conv_edge_small<", "k_conv_edge_bf16", "k_conv_edge_f16x3", "k_conv_edge_wide<2, 2>",
              "k_edge_encode<44", "k_edge_encode_small<45", "k_edge_encode_wide<44, 2, false>", "k_edge_encode_wide<4, 2, true>", "k_node<0, false>",
              "k_node<0, true>", "k_node_wide<2, false>", "k_node_wide<2, true>", "k_conv_edge_f16x3_wide<2, 2>", "k_conv_edge_bf16_wide<2, 2>",
              "k_edge_update<2>", "k_conv_edge_wide_d<2, 2, 2>", "k_edge_encode_wide_d<45, 2, 2>", "k_node_wide_d<2, 2>", "k_skin_check", "k_filter<true>", "k_com_partial"):
        assert k in names, k
    assert len(resources) >= 55


def test_no_hot_kernel_spills_to_scratch(resources):
    bad = {n: (v.get("vgpr_spill_count", 0), v.get("private_segment_fixed_size", 0)) for n, v in resources.items()
           if HOT.search(n) and (v.get("vgpr_spill_count", 0) > 0 or v.get("private_segment_fixed_size", 0) > 0)}
    assert not bad, f"hot kernels with VGPR spills / scratch: {bad}"


def test_register_budgets_match_the_occupancy_the_kernels_are_written_for(resources):
    """2 waves per SIMD (256 registers) for the 512-thread persistent kernels, 3 workgroups per CU for k_node (<= 168)."""
    def regs(prefix):
        hit = [v for n, v in resources.items() if n.startswith(prefix) or ("::" + prefix) in n]
        assert hit, prefix
        return max(v["vgpr_count"] + v.get("agpr_count", 0) for v in hit)
    assert regs("k_conv_edge<") <= 256 and regs("k_edge_encode<") <= 256 and regs("k_conv_edge_bf16") <= 256
    assert regs("k_conv_edge_f16x3") <= 256          # two waves per SIMD since round 4 (one wave with 498 registers before)
    assert regs("k_node<") <= 168
