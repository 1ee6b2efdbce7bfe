#!/usr/bin/env python3
"""Cycles of k_fwd_setup's phases (LQP_DBG_SETUP=1: stamps of thread 0 per problem): [lu|hard64]"""
import os, sys
os.environ["LQP_DBG_SETUP"] = "1"
os.environ.setdefault("LQP_ENV_NOCACHE", "1")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lqp_py_amd as L
from lqp_py_amd import _lib
from lqp_py_amd.synthetic import create_hard_qp_data
from tools.profile_workload import device_batch
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "lu"
if which == "hard64":
    data = create_hard_qp_data(250, 0.85, range(128), dtype=torch.float64, device=dev); extra = {}
else:
    data = device_batch(dev, 128, 500, 0); extra = {"linsolve": "lu"}
B = data[0].shape[0]
layer = L.SolveBoxQP(control=dict(L.box_qp_control(eps_abs=1e-5, eps_rel=1e-5), **extra))
lib = _lib.load()
layer(*data); torch.cuda.synchronize()
dbg = torch.zeros(B * 16, dtype=torch.int64, device=dev)
lib.lqp_debug_set_lu_counters(_lib.ptr(dbg))
layer(*data); torch.cuda.synchronize()
lib.lqp_debug_set_lu_counters(None)
c = dbg[:B * 8].view(B, 8).double()
names = ["zeroes+small loads", "column maxima", "scaling vector", "store D", "scale pass (Qs, M)", "equality block", "bounds/state", "-"]
print(which, " ".join(f"[{n}] {v/1e3:.1f}k" for n, v in zip(names, c.mean(0).tolist())), f"| sum {c[:, :7].sum(1).mean()/1e3:.1f}k cycles")
