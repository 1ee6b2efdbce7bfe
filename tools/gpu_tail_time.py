#!/usr/bin/env python3
"""Time a forward that needs the continuation kernel (adaptive-rho refactorisation after iteration 100)."""
import os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from lqp_py_amd import _lib
import lqp_py_amd.solve_box_qp_admm_torch as L
from lqp_py_amd.synthetic import create_qp_data
from lqp_py_amd.control import box_qp_control
dev = torch.device("cuda:0")
inp = [t.to(dev) for t in create_qp_data(500, 128, seed=0)]
for ls in ("spd", "lu"):
    ctl = box_qp_control(eps_abs=1e-5, eps_rel=1e-5, rho=100.0, linsolve=ls)
    sol = L.torch_solve_box_qp(*inp, dict(ctl))
    torch.cuda.synchronize()
    _lib.profile(enable=True, reset=True)
    t0 = time.perf_counter()
    for _ in range(5):
        sol = L.torch_solve_box_qp(*inp, dict(ctl))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    pr = {k: round(v[0] / 5, 3) for k, v in _lib.profile().items() if v[1]}
    _lib.profile(enable=False)
    print(ls, "iters", sol["iter"], "n_factor", sol["_stats"]["n_factor"], f"{dt*1e3:.3f} ms", pr, flush=True)
