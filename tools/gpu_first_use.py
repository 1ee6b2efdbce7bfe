#!/usr/bin/env python3
"""Is a step on never-seen tensors slower than the steady state, or are the first steps of a process slow whatever they touch?
Ten batches; the first-use pass over batches 1..9 is timed after k warm-up steps on batch 0 alone (k = 1, 5, 30)."""
import os, sys, time, subprocess
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) < 2:
    for k in (1, 5, 30, 100):
        subprocess.run([sys.executable, os.path.abspath(__file__), str(k)])
    sys.exit(0)
k = int(sys.argv[1])
import lqp_py_amd as L
from lqp_py_amd.synthetic import create_qp_data
dev = torch.device("cuda:0")
B, n = 128, 500
data = [[t.to(dev) for t in create_qp_data(n, B, seed=s)] for s in range(10)]
ones = torch.ones(B, n, 1, device=dev)
layer = L.SolveBoxQP(control=dict(L.box_qp_control(eps_abs=1e-5, eps_rel=1e-5), sync=False))
def step(i):
    Q, p, A, b, lb, ub = data[i % 10]
    Q = Q.detach().requires_grad_(True); p = p.detach().requires_grad_(True)
    layer(Q, p, A, b, lb, ub).backward(ones)
for _ in range(k): step(0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(1, 10): step(i)
torch.cuda.synchronize()
first = (time.perf_counter() - t0) / 9
for i in range(10): step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(40): step(i)
torch.cuda.synchronize(); L.synchronize()
steady = (time.perf_counter() - t0) / 40
print(f"warm-up steps on batch 0: {k:2d}   first use of batches 1..9: {first*1e3:.4f} ms/step   steady state: {steady*1e3:.4f} ms/step   ratio {first/steady:.3f}")
