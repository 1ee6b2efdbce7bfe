#!/usr/bin/env python3
"""Forward solves that take many iterations (tight tolerance) or adapt rho: time per iteration and the launches used."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lqp_py_amd as L
from lqp_py_amd import _lib
from lqp_py_amd.synthetic import create_qp_data
dev = torch.device("cuda:0")
B, n = 128, 500
inp = [t.to(dev) for t in create_qp_data(n, B, seed=0)]
cases = [("eps 1e-5", dict(eps_abs=1e-5, eps_rel=1e-5), inp),
         ("eps 1e-7", dict(eps_abs=1e-7, eps_rel=1e-7), inp),
         ("eps 1e-7, max_iters 400", dict(eps_abs=1e-9, eps_rel=1e-9, max_iters=400), inp),
         ("Q x 50, rho = 100 (adapts)", dict(eps_abs=1e-5, eps_rel=1e-5, rho=100.0, scale=False), [inp[0] * 50] + inp[1:]),
         ("rho = 0.01 (adapts)", dict(eps_abs=1e-5, eps_rel=1e-5, rho=0.01), inp)]
for name, kw, data in cases:
    ctl = dict(L.box_qp_control(**kw), sync=False)
    layer = L.SolveBoxQP(control=ctl)          # (a module: what it learnt about the bounds is remembered between calls, no host look)
    with torch.no_grad():
        for _ in range(3): layer(*data)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): layer(*data)
    torch.cuda.synchronize(); L.synchronize()
    dt = (time.perf_counter() - t0) / 10
    st = L.solve_box_qp_admm_torch.last_forward_status(dev)
    _lib.profile(enable=True, reset=True)
    with torch.no_grad():
        for _ in range(3): layer(*data)
    torch.cuda.synchronize()
    pr = {k: (round(v[0] / 3, 3), v[1] // 3) for k, v in _lib.profile().items() if v[1]}
    _lib.profile(enable=False)
    it = st["iters"] + 1
    print(f"{name:32s}: {dt*1e3:8.3f} ms  iters {it:5d}  n_factor {st['n_factor']}  {dt*1e6/it:6.2f} us/iter overall  {pr}", flush=True)
