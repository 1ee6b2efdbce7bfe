#!/usr/bin/env python3
"""unroll=True at the headline size: step time, the library's kernel classes, and what is left (the eager scaling chain)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lqp_py_amd as L
from lqp_py_amd import _lib
from tools.profile_workload import device_batch
dev = torch.device("cuda:0")
B, n = 128, 500
data = device_batch(dev, B, n, 79)
layer = L.SolveBoxQP(control=L.box_qp_control(eps_rel=1e-5, eps_abs=1e-5, verbose=False, unroll=True))
cot = torch.ones_like(data[1])
def step():
    Q = data[0].detach().requires_grad_(True); p = data[1].detach().requires_grad_(True)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True); t2 = torch.cuda.Event(enable_timing=True)
    t0.record(); x = layer(Q, p, *data[2:]); t1.record(); x.backward(cot); t2.record()
    return t0, t1, t2
for _ in range(2): step()
torch.cuda.synchronize()
ev = [step() for _ in range(5)]
torch.cuda.synchronize()
f = sorted(a.elapsed_time(b) for a, b, _ in ev)[2]; bw = sorted(b.elapsed_time(c) for _, b, c in ev)[2]
_lib.profile(enable=True, reset=True)
for _ in range(3): step()
torch.cuda.synchronize()
pr = {k: round(v[0] / 3, 4) for k, v in _lib.profile().items() if v[1]}
_lib.profile(enable=False)
print(f"unroll B={B} n={n}: forward {f:.3f} ms, backward {bw:.3f} ms (median of 5); library kernel classes per step (ms): {pr}; sum {sum(pr.values()):.3f}")
