import os, sys, hashlib, subprocess
ROOT="/root/repo"
code = r'''
import sys, hashlib, numpy as np, torch
sys.path.insert(0, "/root/repo")
from gamd_amd import workloads
from gamd_amd.engine import GamdForce
from gamd_amd.weights import ModelConfig, make_state_dict, SHIPPED_SCALERS
pos, box, species, bonds = workloads.water_box(2000, mol_per_20A3=251.0, seed=3456, jitter=0.0, wrap=False)
sd = make_state_dict(ModelConfig(kind="water", use_bond=True), 3, 2.9, 1.1)
eng = GamdForce(sd, pos.shape[0], box, 4.2, bond=bonds, scaler=SHIPPED_SCALERS["tip4p"], edge_dtype="bf16")
out = eng.forward(torch.from_numpy(pos).float(), species=species).cpu().numpy()
print("SHA", hashlib.sha256(out.tobytes()).hexdigest()[:16], float(np.abs(out).max()))
'''
for v in sys.argv[1:]:
    env = dict(os.environ, GAMD_LIB=os.path.join(ROOT, "gamd_amd", "libgamd_hip_prof.so"), GAMD_BF16_VARIANT=v)
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print(v, p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-300:])
