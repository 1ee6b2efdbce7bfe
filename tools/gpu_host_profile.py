#!/usr/bin/env python3
"""cProfile of the HOST side of pipelined layer calls (forward + backward), 200 steps: which Python functions the ~100 us
in front of / behind the library calls are made of."""
import os, sys, cProfile, pstats, io
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lqp_py_amd as L
from lqp_py_amd.synthetic import create_qp_data
dev = torch.device("cuda:0")
B, n = int(os.environ.get("B", 128)), int(os.environ.get("N", 500))
inp = [t.to(dev) for t in create_qp_data(n, B, seed=0)]
ones = torch.ones(B, n, 1, device=dev)
layer = L.SolveBoxQP(control=dict(L.box_qp_control(eps_abs=1e-5, eps_rel=1e-5), sync=bool(int(os.environ.get("SYNC", "0")))))
def step():
    Q = inp[0].detach().requires_grad_(True); p = inp[1].detach().requires_grad_(True)
    layer(Q, p, *inp[2:]).backward(ones)
for _ in range(10): step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(200): step()
pr.disable()
torch.cuda.synchronize(); L.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45); print(s.getvalue()[:9000])
