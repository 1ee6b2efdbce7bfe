#!/usr/bin/env python3
"""Host timeline of one SYNCHRONOUS step (the layer's default: reference semantics): where the host time between the
library calls goes.  Medians over 40 steps, microseconds."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lqp_py_amd as L
from lqp_py_amd import _lib
from lqp_py_amd.synthetic import create_qp_data
dev = torch.device("cuda:0")
B, n = 128, 500
inp = [t.to(dev) for t in create_qp_data(n, B, seed=0)]
ones = torch.ones(B, n, 1, device=dev)
lib = _lib.load()
marks = {}
def wrap(name):
    real = getattr(lib, name)
    def f(*a):
        marks[name + "_in"] = time.perf_counter()
        r = real(*a)
        marks[name + "_out"] = time.perf_counter()
        return r
    setattr(lib, name, f)
for nm in ("lqp_boxqp_forward", "lqp_boxqp_forward_finish", "lqp_boxqp_backward_fp"):
    wrap(nm)
# finer marks inside pre_fwd / bwd_tail: entry of the autograd function, entry of _forward_solve, return of the backward function
import lqp_py_amd.solve_box_qp_admm_torch as SB
_fs = SB._forward_solve
def _fs_marked(*a, **k):
    marks["forward_solve_in"] = time.perf_counter()
    return _fs(*a, **k)
SB._forward_solve = _fs_marked
_run = SB._fp_backward_run
def _run_marked(prep, g):
    marks["bwd_run_in"] = time.perf_counter()
    r = _run(prep, g)
    marks["bwd_run_out"] = time.perf_counter()
    return r
SB._fp_backward_run = _run_marked
layer = L.SolveBoxQP(control=L.box_qp_control(eps_abs=1e-5, eps_rel=1e-5))
rows = []
prev_end = None
for it in range(50):
    t0 = time.perf_counter()
    Q = inp[0].detach().requires_grad_(True); p = inp[1].detach().requires_grad_(True)
    t0b = time.perf_counter()
    x = layer(Q, p, *inp[2:])
    t1 = time.perf_counter()
    x.backward(ones)
    t2 = time.perf_counter()
    m = dict(marks)
    if it >= 10:
        rows.append(dict(
            pre_fwd=m["lqp_boxqp_forward_in"] - t0,
            pre_fwd_leaves=t0b - t0,
            pre_fwd_to_solve=m["forward_solve_in"] - t0b,
            pre_fwd_in_solve=m["lqp_boxqp_forward_in"] - m["forward_solve_in"],
            to_bwd_run=m["bwd_run_in"] - t1,
            bwd_run_pre=m["lqp_boxqp_backward_fp_in"] - m["bwd_run_in"],
            bwd_run_post=m["bwd_run_out"] - m["lqp_boxqp_backward_fp_out"],
            bwd_engine_tail=t2 - m["bwd_run_out"],
            enqueue_fwd=m["lqp_boxqp_forward_out"] - m["lqp_boxqp_forward_in"],
            views=m.get("lqp_boxqp_forward_finish_in", m["lqp_boxqp_forward_out"]) - m["lqp_boxqp_forward_out"],
            wait_fwd=m.get("lqp_boxqp_forward_finish_out", m["lqp_boxqp_forward_out"]) - m.get("lqp_boxqp_forward_finish_in", m["lqp_boxqp_forward_out"]),
            fwd_tail=t1 - m.get("lqp_boxqp_forward_finish_out", m["lqp_boxqp_forward_out"]),
            to_bwd_call=m["lqp_boxqp_backward_fp_in"] - t1,
            bwd_call=m["lqp_boxqp_backward_fp_out"] - m["lqp_boxqp_backward_fp_in"],
            bwd_tail=t2 - m["lqp_boxqp_backward_fp_out"],
            step=t2 - t0))
torch.cuda.synchronize()
med = lambda k: sorted(r[k] for r in rows)[len(rows) // 2] * 1e6
print("synchronous step, host timeline (us, medians):")
for k in rows[0]:
    print(f"  {k:14s} {med(k):8.1f}")
print("  GPU kernels per step ~800 us (forward ~640, backward ~180)")
