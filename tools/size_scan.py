#!/usr/bin/env python3
"""Step time of the LJ workload (rho* = 0.5, cutoff 3 sigma, fp32, skin reuse, BAOAB on device) over the system size:
python tools/size_scan.py [N ...].  GPU box only; the C2 line of bench.py is the N = 10 000 point."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gamd_amd import workloads                                           # noqa: E402
from gamd_amd.engine import GamdForce                                    # noqa: E402
from gamd_amd.weights import ModelConfig, make_state_dict, SHIPPED_SCALERS   # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [1000, 2000, 5000, 10000, 20000, 50000, 100000, 200000]
sd = make_state_dict(ModelConfig(kind="lj"), 0, 7.0, 2.2)
rc = 3.0 * workloads.LJ_SIGMA
print("| atoms | edges | ms / step | atom-steps/s | GB device memory |")
print("|---|---|---|---|---|")
for n in sizes:
    pos, box = workloads.lj_box(n)
    eng = GamdForce(sd, n, box, rc, scaler=SHIPPED_SCALERS["lj"], neighbor_skin=rc / 6)
    x = torch.from_numpy(pos).float().cuda()
    v = torch.from_numpy(workloads.maxwell_boltzmann(n, temperature_k=100.0, seed=1)).float().cuda()
    f = eng.forward(x, denormalize=True).clone()
    steps = 100 if n <= 20000 else 30
    eng.md_run(x, v, f, 10, temperature_k=100.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.md_run(x, v, f, steps, temperature_k=100.0, first_step=10)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    free, total = torch.cuda.mem_get_info()
    print(f"| {n} | {eng.counts()[0]} | {dt * 1e3:.3f} | {n / dt:.3e} | {(total - free) / 2**30:.2f} |")
    assert torch.isfinite(x).all() and eng.last_status in (0, 1)
    eng.close()
    del eng, x, v, f
    torch.cuda.empty_cache()
