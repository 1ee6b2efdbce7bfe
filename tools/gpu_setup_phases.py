#!/usr/bin/env python3
"""In-kernel cycle breakdown of k_fwd_setup (debug counters of thread 0)."""
import os, sys
os.environ.setdefault("LQP_ENV_NOCACHE", "1")      # (this tool flips library knobs between solves)
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import lqp_py_amd as L
from lqp_py_amd import _lib
from lqp_py_amd.synthetic import create_qp_data
dev = torch.device("cuda:0")
os.environ["LQP_DBG_SETUP"] = "1"          # only the setup kernel writes the debug words
lib = _lib.load()
B, n = 128, 500
inp = [t.to(dev) for t in create_qp_data(n, B, seed=0)]
ctl = L.box_qp_control(eps_abs=1e-5, eps_rel=1e-5, max_iters=1)
dbg = torch.zeros(B * 8, dtype=torch.int64, device=dev)
L.torch_solve_box_qp(*inp, dict(ctl))
lib.lqp_debug_set_lu_counters(_lib.ptr(dbg))
L.torch_solve_box_qp(*inp, dict(ctl))
torch.cuda.synchronize()
lib.lqp_debug_set_lu_counters(None)
c = dbg.view(B, 8).double().mean(0)
names = ["zero xchg, |p|", "column maxima (reads Q)", "D", "quantiles / beta", "scaling pass", "rho, eq block", "bounds, state"]
for nm, v in zip(names, c.tolist()):
    print("%-28s %9.0f cycles" % (nm, v))
