#!/usr/bin/env python3
"""The layer alone at the minibatch of experiment 2 (B = 32, n = 500, default synchronous calls): forward / backward wall time with
only p requiring a gradient and with Q too, plus the kernel classes of one step -- what of experiment 2's epoch is the layer."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import lqp_py_amd as L
from lqp_py_amd import _lib
from lqp_py_amd.synthetic import create_qp_data
dev = torch.device("cuda:0")
B, n = 32, 500
Q, p, A, b, lb, ub = [t.to(dev) for t in create_qp_data(n, B, seed=0)]
layer = L.SolveBoxQP(control=L.box_qp_control(eps_rel=1e-5, eps_abs=1e-5, verbose=False, reduce='max'))
ones = torch.ones(B, n, 1, device=dev)
for mode in ("p only", "Q and p"):
    for it in range(8):
        pp = p.detach().requires_grad_(True)
        QQ = Q.detach().requires_grad_(mode != "p only")
        torch.cuda.synchronize()
        t0 = time.perf_counter(); z = layer(QQ, pp, A, b, lb, ub); torch.cuda.synchronize(); t1 = time.perf_counter()
        z.backward(ones); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(mode, "forward %.3f ms backward %.3f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
_lib.profile(enable=True, reset=True)
pp = p.detach().requires_grad_(True)
z = layer(Q, pp, A, b, lb, ub); z.backward(ones); torch.cuda.synchronize()
print({k: round(v[0], 4) for k, v in _lib.profile().items() if v[1]})
