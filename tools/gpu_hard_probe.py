#!/usr/bin/env python3
"""The reference's hard distribution in float64 (experiments/experiment_1_hard.py: n = 250, m = 16, B = 128): per-kernel-class times."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lqp_py_amd as L
from lqp_py_amd import _lib
from lqp_py_amd.solve_box_qp_admm_torch import last_forward_status
from lqp_py_amd.synthetic import create_hard_qp_data
dev = torch.device("cuda:0")
dt_ = torch.float64 if (len(sys.argv) < 2 or sys.argv[1] == "f64") else torch.float32
NH = int(os.environ.get("N", "250"))
hard = create_hard_qp_data(NH, 0.85, range(int(os.environ.get("B", "128"))), dtype=dt_, device=dev)
SYNC = bool(int(os.environ.get("SYNC", "0")))          # SYNC=1: the layer's default synchronous calls (what experiment_1_hard.py gets)
layer = L.SolveBoxQP(control=dict(L.box_qp_control(eps_abs=1e-5, eps_rel=1e-5), sync=SYNC))
cot = torch.ones_like(hard[1])
def step():
    Q = hard[0].detach().requires_grad_(True); p = hard[1].detach().requires_grad_(True)
    layer(Q, p, *hard[2:]).backward(cot)
for _ in range(10): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(40): step()
torch.cuda.synchronize(); L.synchronize()
dt = (time.perf_counter() - t0) / 40
st = last_forward_status(dev)
_lib.profile(enable=True, reset=True)
for _ in range(5): step()
torch.cuda.synchronize()
pr = {k: round(v[0] / 5, 4) for k, v in _lib.profile().items() if v[1]}
_lib.profile(enable=False)
print(f"{dt_} sync={SYNC}: {dt*1e3:.3f} ms/step  iters {st['iters']} linsolve {st['linsolve_used']} mode {st['mode_used']}  {pr}")
