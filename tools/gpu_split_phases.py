#!/usr/bin/env python3
"""In-kernel cycle breakdown of the two-workgroup loop (debug counters of wave 0, both workgroups of a QP)."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from lqp_py_amd import _lib
import lqp_py_amd.solve_box_qp_admm_torch as L
from lqp_py_amd.synthetic import create_qp_data
from lqp_py_amd.control import box_qp_control
dev = torch.device("cuda:0")
lib = _lib.load()
B, n = int(os.environ.get("BATCH", "128")), int(os.environ.get("N_X", "500"))
inp = [t.to(dev) for t in create_qp_data(n, B, seed=0)]
ctl = box_qp_control(eps_abs=1e-5, eps_rel=1e-5, linsolve="spd")
sol = L.torch_solve_box_qp(*inp, dict(ctl))
dbg = torch.zeros(B * 8, dtype=torch.int64, device=dev)
lib.lqp_debug_set_lu_counters(_lib.ptr(dbg))
sol = L.torch_solve_box_qp(*inp, dict(ctl))
torch.cuda.synchronize()
lib.lqp_debug_set_lu_counters(None)
c = dbg.view(B, 8).double().mean(0)
it = sol["iter"] + 1
r = (c / it).tolist()
print(f"part 0, wave 0, cycles/iter: product {r[0]:.0f} combine+publish {r[1]:.0f} poll {r[2]:.0f} update {r[3]:.0f} check(avg) {r[4]:.0f} "
      f"end-barrier {r[5]:.0f} | total {sum(r[:6]):.0f}  (iters {it}, mode {sol['_stats']['mode_used']})")
