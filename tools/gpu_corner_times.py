#!/usr/bin/env python3
"""Pipelined forward+backward step and its kernel classes at parameter corners the benchmark configurations do not visit
(equality-row counts, box-only, float64, small n): anything far off its neighbours is a kernel with a slow path."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lqp_py_amd as L
from lqp_py_amd import _lib
from lqp_py_amd.synthetic import create_qp_data
dev = torch.device("cuda:0")
cases = [(128, 500, 0, "f32"), (128, 500, 1, "f32"), (128, 500, 2, "f32"), (128, 500, 4, "f32"), (128, 500, 16, "f32"), (128, 500, 17, "f32"),
         (128, 200, 8, "f32"), (128, 64, 2, "f32"), (128, 100, 3, "f64"), (128, 250, 0, "f64"), (128, 250, 4, "f64"), (32, 500, 8, "f32"),
         (128, 1000, 4, "f32"), (16, 1000, 0, "f32")]
for B, n, m, dt in cases:
    dtype = torch.float64 if dt == "f64" else torch.float32
    Q, p, _, _, lb, ub = create_qp_data(n, B, seed=3, with_eq=False)
    g = torch.Generator().manual_seed(4)
    A = torch.randn(B, m, n, generator=g) if m else None
    b = (A @ (0.5 * (lb + ub))) if m else None
    inp = [None if t is None else t.to(dtype).to(dev) for t in (Q, p, A, b, lb, ub)]
    layer = L.SolveBoxQP(control=dict(L.box_qp_control(eps_abs=1e-5, eps_rel=1e-5), sync=False))
    cot = torch.ones(B, n, 1, dtype=dtype, device=dev)
    def step():
        Qg = inp[0].detach().requires_grad_(True); pg = inp[1].detach().requires_grad_(True)
        layer(Qg, pg, *inp[2:]).backward(cot)
    for _ in range(5): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize(); L.synchronize()
    dtm = (time.perf_counter() - t0) / 20
    st = L.solve_box_qp_admm_torch.last_forward_status(dev)
    _lib.profile(enable=True, reset=True)
    for _ in range(5): step()
    torch.cuda.synchronize()
    pr = {k: round(v[0] / 5, 3) for k, v in _lib.profile().items() if v[1]}
    _lib.profile(enable=False)
    print(f"B {B:4d} n {n:4d} m {m:2d} {dt}: {dtm*1e3:7.3f} ms/step  iters {st['iters']:3d} linsolve {st['linsolve_used']}  {pr}", flush=True)
