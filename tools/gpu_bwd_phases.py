#!/usr/bin/env python3
"""In-kernel cycle breakdown of the Cholesky backward (debug counters of thread 0)."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import lqp_py_amd as L
from lqp_py_amd import _lib
from lqp_py_amd.synthetic import create_qp_data
dev = torch.device("cuda:0")
lib = _lib.load()
B, n = 128, 500
inp = [t.to(dev) for t in create_qp_data(n, B, seed=0)]
Q = inp[0].clone().requires_grad_(True)
ctl = L.box_qp_control(eps_abs=1e-5, eps_rel=1e-5)
x = L.SolveBoxQP(control=ctl)(Q, *inp[1:])
dbg = torch.zeros(B * 8, dtype=torch.int64, device=dev)
lib.lqp_debug_set_lu_counters(_lib.ptr(dbg))
x.backward(torch.ones_like(x))
torch.cuda.synchronize()
lib.lqp_debug_set_lu_counters(None)
c = dbg.view(B, 8).double()
print("mean cycles: factor %.0f  solves %.0f  schur+finish %.0f   mean Kb %.2f (min %d max %d)" % (
    c[:, 0].mean(), c[:, 1].mean(), c[:, 2].mean(), c[:, 3].mean(), int(c[:, 3].min()), int(c[:, 3].max())))
