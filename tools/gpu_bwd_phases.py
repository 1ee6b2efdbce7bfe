#!/usr/bin/env python3
"""In-kernel cycle breakdown of the Cholesky backward (debug counters of thread 0)."""
import os, sys
os.environ.setdefault("LQP_ENV_NOCACHE", "1")      # (this tool flips library knobs between solves)
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
ctl = L.box_qp_control(eps_abs=1e-5, eps_rel=1e-5)
grads = {}
for la in (0, 1):
    os.environ["LQP_BWD_LOOKAHEAD"] = str(la)
    Q = inp[0].clone().requires_grad_(True)
    p = inp[1].clone().requires_grad_(True)
    x = L.SolveBoxQP(control=ctl)(Q, p, *inp[2:])
    dbg = torch.zeros(B * 8, dtype=torch.int64, device=dev)
    lib.lqp_debug_set_lu_counters(_lib.ptr(dbg))
    x.backward(torch.ones_like(x))
    torch.cuda.synchronize()
    lib.lqp_debug_set_lu_counters(None)
    grads[la] = (Q.grad.clone(), p.grad.clone())
    c = dbg.view(B, 8).double()
    print("lookahead %d  mean cycles: factor %.0f  solves %.0f  schur+finish %.0f   mean Kb %.2f (min %d max %d)" % (
        la, c[:, 0].mean(), c[:, 1].mean(), c[:, 2].mean(), c[:, 3].mean(), int(c[:, 3].min()), int(c[:, 3].max())))
    if la:
        print("   chain waves: waits %.0f  pivot blocks %.0f   tile waves: wait for W %.0f  rest of step %.0f" % (
            c[:, 4].mean(), c[:, 5].mean(), c[:, 6].mean(), c[:, 7].mean()))
    else:       # (the same words then hold the two halves of the solves)
        print("   solves: forward substitution %.0f  backward substitution %.0f" % (c[:, 4].mean(), c[:, 5].mean()))
print("bit-identical gradients:", torch.equal(grads[0][0], grads[1][0]) and torch.equal(grads[0][1], grads[1][1]),
      " finite:", bool(torch.isfinite(grads[1][0]).all()))
