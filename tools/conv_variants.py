#!/usr/bin/env python3
"""A/B timing of k_conv_edge variants (profiling build, libgamd_hip_prof.so) on the C2 workload.

    python tools/conv_variants.py 3592 0 520 1544 ...      (3592 = production, bit masks in conv_edge.hip)

Every variant runs in its own process (the variant is latched per process by GAMD_CONV_VARIANT).  Prints one line per
variant: average conv-layer time (HIP events around the launch, as bench.py does), TFLOP/s of the 131 072 FLOP/edge
figure, and whether the forces are bit-identical to variant 0's.  --cycles adds the s_memtime breakdown of variant | 1.
GPU box only; not part of the product."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.environ.get("CV_LIB") or os.path.join(ROOT, "gamd_amd", "libgamd_hip_prof.so")
SEGMENTS = ["p1 bias init", "p1 gemm+post", "boundary 1 (late waves: barrier + DMA issue)", "S gather issue",
            "boundary 1 (early waves) + p2 gemm+post", "boundary 2 (late waves)", "hn gather issue",
            "boundary 2 (early waves) + idx loads + bias", "p3 gemm+post", "boundary 3 + DMA issue", "p4 gemm+segment-sum",
            "boundary 4 (late waves)", "e-prefetch issue", "boundary 4 (early waves)", "piece stores + D gather issue"]


def child():
    sys.path.insert(0, ROOT)
    import hashlib
    import numpy as np
    import torch
    from gamd_amd.engine import GamdForce
    from gamd_amd.weights import ModelConfig, make_state_dict, SHIPPED_SCALERS
    from gamd_amd.workloads import lj_box
    n = int(os.environ.get("CV_ATOMS", "10000"))
    pos, box = lj_box(n)
    sd = make_state_dict(ModelConfig(kind="lj"), 0, 7.0, 2.2)
    eng = GamdForce(sd, n, box, 3.0 * 3.4, scaler=SHIPPED_SCALERS["lj"])
    p = torch.from_numpy(pos).float().cuda()
    for _ in range(3):
        out = eng.forward(p, inplace=True)
    reps = 25
    eng.timing_enable(True)
    for _ in range(reps):
        out = eng.forward(p, inplace=True)
    ms, cnt = eng.timing_read()
    eng.timing_enable(False)
    E = eng.counts()[0]
    n_tiles = (E + 31) // 32
    main_tiles = n_tiles
    rec = {"variant": int(os.environ.get("GAMD_CONV_VARIANT", "0")), "E": E, "main_tiles": main_tiles, "conv_us": ms / cnt * 1e3,
           "tflops": E * 131072 / (ms / cnt * 1e-3) / 1e12, "launches": cnt,
           "sha": hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]}
    if rec["variant"] & 1:
        t = eng._dbg(5, (256, 8, 16), np.int64).astype(np.float64)
        tot = t[:, :, :15].sum(-1)
        rec["per_wave"] = [[float(t[:, wv, i].mean()) for i in range(15)] for wv in range(8)]
        rec["cycles"] = {"per_wave_total": float(tot.mean()),
                         "segments": {nm: [float(t[:, :, i].mean()), float(t[:, :4, i].mean()), float(t[:, 4:, i].mean())]
                                      for i, nm in enumerate(SEGMENTS)}}
    print("CVJSON " + json.dumps(rec))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    variants = [int(a) for a in args] or [0]
    base_sha = None
    rows = []
    for v in variants:
        env = dict(os.environ, GAMD_LIB=PROF, GAMD_CONV_VARIANT=str(v), CV_CHILD="1")
        p = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in p.stdout.splitlines() if l.startswith("CVJSON ")]
        if p.returncode != 0 or not line:
            print(f"variant {v}: FAILED rc={p.returncode}\n{p.stderr[-1500:]}")
            continue
        r = json.loads(line[0][7:])
        if base_sha is None:
            base_sha = r["sha"]
        rows.append(r)
        print(f"variant {v:3d}: conv {r['conv_us']:7.1f} us  {r['tflops']:6.1f} TF  frac {r['tflops'] / 157.3:.3f}  "
              f"E={r['E']}  forces {'identical to first' if r['sha'] == base_sha else 'DIFFER from first'}", flush=True)
        if "cycles" in r:
            c = r["cycles"]
            tiles_per_wave = r["main_tiles"] / 8 / 256
            print(f"    s_memtime ticks per tile (last launch), all / waves 0-3 / waves 4-7; per-wave total {c['per_wave_total']:.0f}")
            for nm, (a, o, y) in c["segments"].items():
                print(f"    | {nm:24s} | {a / tiles_per_wave:8.0f} | {o / tiles_per_wave:8.0f} | {y / tiles_per_wave:8.0f} |")
            print(f"    | sum | {c['per_wave_total'] / tiles_per_wave:8.0f} |   (ideal 4 x 2 x 256 MFMA x 64 = 131072)")
            if "--perwave" in sys.argv:
                print("    per wave (columns = waves 0..7), ticks per tile:")
                for i, nm in enumerate(SEGMENTS):
                    print(f"    | {nm[:40]:40s} | " + " | ".join(f"{r['per_wave'][wv][i] / tiles_per_wave:7.0f}" for wv in range(8)) + " |")
    return 0


if __name__ == "__main__":
    if os.environ.get("CV_CHILD") == "1":
        child()
    else:
        sys.exit(main())
