#!/usr/bin/env python3
"""Host-side cost of one pipelined forward / backward call (time until the call returns, GPU idle before):
what sits in front of the first kernel when a step starts from an idle queue (experiment_1's protocol)."""
import os, sys, time, cProfile, pstats, io
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import lqp_py_amd as L
from lqp_py_amd.synthetic import create_qp_data
dev = torch.device("cuda:0")
B, n = 128, 500
inp = [t.to(dev) for t in create_qp_data(n, B, seed=0)]
ones = torch.ones(B, n, 1, device=dev)
for sync in (False, True):
    layer = L.SolveBoxQP(control=dict(L.box_qp_control(eps_abs=1e-5, eps_rel=1e-5), sync=sync))
    def step():
        Q = inp[0].detach().requires_grad_(True); p = inp[1].detach().requires_grad_(True)
        t0 = time.perf_counter(); x = layer(Q, p, *inp[2:]); t1 = time.perf_counter(); x.backward(ones); t2 = time.perf_counter()
        return t1 - t0, t2 - t1
    for _ in range(5): step(); torch.cuda.synchronize()
    f, b = [], []
    for _ in range(30):
        torch.cuda.synchronize(); L.synchronize()
        a, c = step(); f.append(a); b.append(c)
    torch.cuda.synchronize(); L.synchronize()
    f.sort(); b.sort()
    print(f"sync={sync}: host time of forward call {f[15]*1e6:.0f} us (min {f[0]*1e6:.0f}), backward call {b[15]*1e6:.0f} us (min {b[0]*1e6:.0f})")
layer = L.SolveBoxQP(control=dict(L.box_qp_control(eps_abs=1e-5, eps_rel=1e-5), sync=False))
pr = cProfile.Profile()
for _ in range(50):
    torch.cuda.synchronize()
    Q = inp[0].detach().requires_grad_(True); p = inp[1].detach().requires_grad_(True)
    pr.enable(); x = layer(Q, p, *inp[2:]); x.backward(ones); pr.disable()
L.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18); print(s.getvalue()[:3500])
