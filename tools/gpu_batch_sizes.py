#!/usr/bin/env python3
"""Pipelined forward+backward step over batch sizes that are / are not multiples of 8 (two workgroups per QP: partners b and b + B
land on one XCD only when B % 8 == 0)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lqp_py_amd as L
from lqp_py_amd.synthetic import create_qp_data
dev = torch.device("cuda:0")
n = int(os.environ.get("N", 500))
for B in [int(a) for a in sys.argv[1:]] or [96, 100, 104, 120, 125, 128, 30, 32, 60, 64]:
    inp = [t.to(dev) for t in create_qp_data(n, B, seed=1)]
    layer = L.SolveBoxQP(control=dict(L.box_qp_control(eps_abs=1e-5, eps_rel=1e-5), sync=False))
    cot = torch.ones(B, n, 1, device=dev)
    def step():
        Q = inp[0].detach().requires_grad_(True); p = inp[1].detach().requires_grad_(True)
        layer(Q, p, *inp[2:]).backward(cot)
    for _ in range(10): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40): step()
    torch.cuda.synchronize(); L.synchronize()
    dt = (time.perf_counter() - t0) / 40
    print(f"B {B:4d} n {n}: {dt*1e3:.4f} ms/step  {B/dt:.0f} QPs/s", flush=True)
