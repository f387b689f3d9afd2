"""bench.py keeps the driver's contract: flags, exactly one JSON line on stdout, the required keys."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", *flags],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_default_line_has_the_contract_keys():
    d = _run("--no-cpu-baseline")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert d["unit"] == "atom-steps/s" and "10k-atom LJ box" in d["metric"]
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 10000 * 6 / (d["ms_per_step"] * 6e-3)) / d["value"] < 1e-6
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.3 < r["frac"] < 1.0
    assert r["launches"] == 4 * 6                         # four conv layers per timed step, timed live with HIP events


def test_split_fp16_line_is_labelled_as_such():
    d = _run("--no-cpu-baseline", "--edge-dtype", "f16x3")
    assert d["dtype"].startswith("f16x3") and d["roofline"]["kernel"] == "k_conv_edge_f16x3"
    assert d["roofline"]["peak"] == 2500.0 and d["roofline"]["frac"] < 1.0


def test_graft_entry_smoke_runs():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.smoke()
