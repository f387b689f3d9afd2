#!/usr/bin/env python3
"""Per-call latency of the reference-shaped host-buffer boundary, predict_forces(np.float64[N,3]) -> np.float64[N,3]
(one sync per call, as the reference's OpenMM drivers use it): python tools/predict_forces_latency.py.  GPU box only."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gamd_amd import compat, workloads                                   # noqa: E402
from gamd_amd.weights import ModelConfig, make_state_dict              # noqa: E402

import hashlib                                                           # noqa: E402

for n, rc in ((258, 7.5), (10000, 10.2)):
    pos, box = workloads.lj_box(n, seed=3)
    m = compat.ParticleNetLightningLJ(None, make_state_dict(ModelConfig(kind="lj"), 0, 7.0, 2.2), num_atoms=n,
                                      box_size=float(box), cutoff=rc)
    m.cuda(); m.eval()
    p = pos.astype(np.float64)
    # the pinned one-synchronisation form (default) against the round-5 form (pageable copy in, kernels + sync, copy out + sync):
    # alternating blocks on the same handle, same positions -> same bits
    for rnd in range(2):
        for legacy in (False, True):
            if legacy:
                os.environ["GAMD_PREDICT_LEGACY"] = "1"
            else:
                os.environ.pop("GAMD_PREDICT_LEGACY", None)
            # (per-call times, median next to the mean: Python's cyclic garbage collector runs its first FULL collection around the
            #  450th call of a process — 39 ms with torch's object graph loaded, gone with gc.disable() / gc.freeze() —, which a
            #  300-call mean shows as + 0.13 ms whichever form is running at the time)
            for _ in range(20 if rnd else (600 if n < 1000 else 50)):
                f = m.predict_forces(p)
            reps = 300 if n < 1000 else 100
            ts = []
            for k in range(reps):
                t0 = time.perf_counter()
                f = m.predict_forces(p + 1e-4 * (k & 1))
                ts.append(time.perf_counter() - t0)
            sha = hashlib.sha256(np.ascontiguousarray(f).tobytes()).hexdigest()[:12]
            print(f"{n:6d} atoms: predict_forces p50 {np.median(ts) * 1e3:.3f} ms, mean {np.mean(ts) * 1e3:.3f} ms per call "
                  f"({'three syncs' if legacy else 'pinned, one sync'}; {f.dtype}, {f.shape}, sha {sha})")
    os.environ.pop("GAMD_PREDICT_LEGACY", None)
