#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS figures of the gfx950 code objects embedded in libgamd_hip.so.

`llvm-objdump --offloading` extracts the code objects from the shared library, `llvm-readelf --notes` prints their
amdhsa metadata.  Used by tests/test_kernel_resources.py (no hot kernel may spill to scratch) and by hand:

    python tools/kernel_resources.py [path/to/libgamd_hip.so]
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM_BIN = os.environ.get("GAMD_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
          "group_segment_fixed_size", "max_flat_workgroup_size")


def demangle(names):
    filt = next((f for f in (os.path.join(LLVM_BIN, "llvm-cxxfilt"), shutil.which("c++filt") or "") if f and os.path.exists(f)), None)
    if filt is None:
        return {n: n for n in names}
    out = subprocess.run([filt] + list(names), capture_output=True, text=True, check=True).stdout.splitlines()
    return dict(zip(names, out))


def kernel_resources(lib_path):
    """{demangled kernel name: {field: int}} for every kernel of every embedded gfx950 code object."""
    lib_path = os.path.abspath(lib_path)
    tmp = tempfile.mkdtemp(prefix="gamd_co_")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib_path, local)
        subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "--offloading", local], cwd=tmp, capture_output=True,
                       text=True, check=True)
        res = {}
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM_BIN, "llvm-readelf"), "--notes", os.path.join(tmp, f)],
                                   capture_output=True, text=True, check=True).stdout
            # one YAML-ish block per kernel under amdhsa.kernels; fields are `.key: value` lines, blocks start at `- .`
            for block in re.split(r"\n\s*- \.", notes):
                m = re.search(r"\.symbol:\s+'?([^\s']+?)\.kd'?", block)
                if not m:
                    continue
                vals = {}
                for k in FIELDS:
                    mm = re.search(r"\.%s:\s+(\d+)" % k, block)
                    if mm:
                        vals[k] = int(mm.group(1))
                res[m.group(1)] = vals
        names = demangle(sorted(res))
        return {names[k]: v for k, v in res.items()}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                              "gamd_amd", "libgamd_hip.so")
    res = kernel_resources(lib)
    print(f"{'kernel':70s} vgpr agpr sgpr vspill sspill scratch   lds")
    for name in sorted(res):
        v = res[name]
        short = re.sub(r"\(anonymous namespace\)::", "", name)
        short = re.sub(r"\(.*\)$", "", short)
        print(f"{short[:70]:70s} {v.get('vgpr_count', 0):4d} {v.get('agpr_count', 0):4d} {v.get('sgpr_count', 0):4d} "
              f"{v.get('vgpr_spill_count', 0):6d} {v.get('sgpr_spill_count', 0):6d} {v.get('private_segment_fixed_size', 0):7d} "
              f"{v.get('group_segment_fixed_size', 0):6d}")


if __name__ == "__main__":
    main()
