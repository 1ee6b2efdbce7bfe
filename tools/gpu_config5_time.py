#!/usr/bin/env python3
"""Forward + backward time of the per-GPU shard of BASELINE configs[4] (B=1024, n=500, m=1) with per-kernel device times."""
import os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import lqp_py_amd as L
from lqp_py_amd import _lib
from lqp_py_amd.synthetic import create_qp_data
from lqp_py_amd.solve_box_qp_admm_torch import last_forward_status
dev = torch.device("cuda:0")
B, n = int(os.environ.get("B", 1024)), int(os.environ.get("N", 500))
FWD_ONLY = os.environ.get("FWD_ONLY", "0") == "1"      # (BASELINE configs 2 / 4 are forward-only)
inp = [t.to(dev) for t in create_qp_data(n, B, seed=0)]
extra = {"linsolve": os.environ["LINSOLVE"]} if os.environ.get("LINSOLVE") else {}
qp = L.SolveBoxQP(control=dict(L.box_qp_control(eps_abs=1e-5, eps_rel=1e-5), sync=False, **extra))
ones = torch.ones(B, n, 1, device=dev)
def step():
    Q = inp[0].detach().requires_grad_(True); p = inp[1].detach().requires_grad_(True)
    x = qp(Q, p, *inp[2:])
    if not FWD_ONLY:
        x.backward(ones)
for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
st = last_forward_status(dev)
print("B=%d step %.3f ms  %.0f QPs/s  iters %d  mode %d  linsolve %d" % (B, dt * 1e3, B / dt, st["iters"], st["mode_used"], st["linsolve_used"]))
_lib.profile(enable=True, reset=True)
for _ in range(reps):
    step()
torch.cuda.synchronize()
prof = _lib.profile()
_lib.profile(enable=False)
print({k: round(v[0] / reps, 4) for k, v in prof.items() if v[1]})
