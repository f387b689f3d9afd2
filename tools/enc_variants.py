#!/usr/bin/env python3
"""A/B timing of k_edge_encode variants (profiling build) on the C2 workload: python tools/enc_variants.py 0 16 32 ...
One process per variant (GAMD_ENC_VARIANT is latched per process).  Prints the event-timed edge_encode stage, TFLOP/s of
the 76 800 FLOP/edge figure, and the max-norm relative difference of e and of the forces to the first variant's."""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.environ.get("CV_LIB") or os.path.join(ROOT, "gamd_amd", "libgamd_hip_prof.so")


def child():
    sys.path.insert(0, ROOT)
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
        out = eng.forward(p)
    ms = []
    for _ in range(15):
        ms.append(dict(eng.profile(p))["edge_encode"])
    E = eng.counts()[0]
    e = eng.debug_e()[:: 97]
    np.savez(os.environ["CV_OUT"], e=e, f=out.cpu().numpy())
    t = float(np.median(ms))
    print("EVJSON " + json.dumps({"variant": int(os.environ.get("GAMD_ENC_VARIANT", "0")), "E": E, "enc_us": t * 1e3,
                                   "tflops": E * 76800 / (t * 1e-3) / 1e12}))


def main():
    import numpy as np
    variants = [int(a) for a in sys.argv[1:]] or [0]
    base = None
    with tempfile.TemporaryDirectory() as td:
        for i, v in enumerate(variants):
            out = os.path.join(td, f"v{i}.npz")
            env = dict(os.environ, GAMD_LIB=PROF, GAMD_ENC_VARIANT=str(v), CV_CHILD="1", CV_OUT=out)
            p = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=600)
            line = [l for l in p.stdout.splitlines() if l.startswith("EVJSON ")]
            if p.returncode != 0 or not line:
                print(f"variant {v}: FAILED rc={p.returncode}\n{p.stderr[-1500:]}")
                continue
            r = json.loads(line[0][7:])
            d = np.load(out)
            if base is None:
                base = {k: d[k] for k in d}
            de = float(np.abs(d["e"] - base["e"]).max() / np.abs(base["e"]).max())
            df = float(np.abs(d["f"] - base["f"]).max() / np.abs(base["f"]).max())
            print(f"variant {v:3d}: encode {r['enc_us']:7.1f} us  {r['tflops']:6.1f} TF  frac {r['tflops'] / 157.3:.3f}  "
                  f"E={r['E']}  vs first: e {de:.2e}  forces {df:.2e}", flush=True)


if __name__ == "__main__":
    if os.environ.get("CV_CHILD") == "1":
        child()
    else:
        main()
