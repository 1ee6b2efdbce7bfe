#!/usr/bin/env python3
"""In-kernel cycle breakdown of the one-workgroup loop kernel on the symmetric path (thread 0 of every QP): where an
iteration goes at small n (BASELINE config 2: n = 100).  N=100 B=128 by default."""
import os, sys
os.environ.setdefault("LQP_ENV_NOCACHE", "1")
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import lqp_py_amd as L
from lqp_py_amd import _lib
from lqp_py_amd.synthetic import create_qp_data
dev = torch.device("cuda:0")
lib = _lib.load()
B, n = int(os.environ.get("B", 128)), int(os.environ.get("N", 100))
inp = [None if t is None else t.to(dev) for t in create_qp_data(n, B, seed=0, with_eq=False, unit_box=True)]
ctl = dict(L.box_qp_control(eps_abs=1e-5, eps_rel=1e-5))
sol = L.torch_solve_box_qp(*inp, dict(ctl))
dbg = torch.zeros(B * 8, dtype=torch.int64, device=dev)
lib.lqp_debug_set_lu_counters(_lib.ptr(dbg))
sol = L.torch_solve_box_qp(*inp, dict(ctl))
torch.cuda.synchronize()
lib.lqp_debug_set_lu_counters(None)
it = sol["iter"] + 1
c = dbg.view(B, 8).double().mean(0).tolist()
names = ["right-hand side", "product", "combine", "update + check + barrier"]
print(f"n={n} B={B}: {it} iterations, linsolve {sol['_stats']['linsolve_used']}, loop workgroups {sol['_stats']['loop_workgroups']}")
for nm, v in zip(names, c[:4]):
    print("%-28s %8.0f cycles per iteration" % (nm, v / it))
print("%-28s %8.0f" % ("sum", sum(c[:4]) / it))
