#!/usr/bin/env python3
"""Where the host time of ONE pipelined forward call goes: before the library call (Python), inside it (its five launches),
after it -- what stands between a step that starts from an idle queue and its first kernel (experiment_1's protocol)."""
import os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import lqp_py_amd as L
from lqp_py_amd import _lib
from lqp_py_amd.synthetic import create_qp_data
dev = torch.device("cuda:0")
B, n = 128, 500
inp = [t.to(dev) for t in create_qp_data(n, B, seed=0)]
lib = _lib.load()
real = lib.lqp_boxqp_forward
marks = {}
def wrapped(*a):
    marks["c0"] = time.perf_counter()
    r = real(*a)
    marks["c1"] = time.perf_counter()
    return r
lib.lqp_boxqp_forward = wrapped
layer = L.SolveBoxQP(control=dict(L.box_qp_control(eps_abs=1e-5, eps_rel=1e-5), sync=False))
pre, inc, post = [], [], []
for it in range(40):
    Q = inp[0].detach().requires_grad_(True); p = inp[1].detach().requires_grad_(True)
    torch.cuda.synchronize(); L.synchronize()
    t0 = time.perf_counter()
    x = layer(Q, p, *inp[2:])
    t1 = time.perf_counter()
    if it >= 10:
        pre.append(marks["c0"] - t0); inc.append(marks["c1"] - marks["c0"]); post.append(t1 - marks["c1"])
torch.cuda.synchronize(); L.synchronize()
med = lambda v: sorted(v)[len(v) // 2] * 1e6
print(f"forward call: {med(pre):.0f} us of Python before the library call, {med(inc):.0f} us inside it, {med(post):.0f} us after")
