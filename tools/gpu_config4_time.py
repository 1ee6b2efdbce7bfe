#!/usr/bin/env python3
"""Forward time of BASELINE configs[3] (B=128, n=1000, m=1) with per-kernel device times."""
import os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import lqp_py_amd as L
from lqp_py_amd import _lib
from lqp_py_amd.synthetic import create_qp_data
from lqp_py_amd.solve_box_qp_admm_torch import last_forward_status
dev = torch.device("cuda:0")
B, n = 128, int(os.environ.get("N", 1000))
inp = [t.to(dev) for t in create_qp_data(n, B, seed=0)]
qp = L.SolveBoxQP(control=dict(L.box_qp_control(eps_abs=1e-5, eps_rel=1e-5), sync=False))
for _ in range(2):
    qp(*inp)
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    qp(*inp)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
st = last_forward_status(dev)
print("forward %.3f ms  iters %d  linsolve %d  factorisations %d" % (dt * 1e3, st["iters"], st["linsolve_used"], st["n_factor"]))
_lib.profile(enable=True, reset=True)
for _ in range(reps):
    qp(*inp)
torch.cuda.synchronize()
prof = _lib.profile()
_lib.profile(enable=False)
print({k: round(v[0] / reps, 4) for k, v in prof.items() if v[1]})
