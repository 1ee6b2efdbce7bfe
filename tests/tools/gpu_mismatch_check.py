#!/usr/bin/env python3
"""For the borderline cases of tools/gpu_sweep.py: which x-update agrees with the CPU oracle's iteration count?"""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import lqp_py_amd.solve_box_qp_admm_torch as L
from oracle import boxqp_oracle as O
dev = torch.device("cuda:0")
TOL = dict(eps_abs=1e-5, eps_rel=1e-5)
agree = {"lu": 0, "spd": 0}; total = 0
for (n, m, B) in [(7, 2, 5), (127, 5, 3), (129, 0, 2), (257, 1, 2), (512, 16, 3), (65, 1, 2), (255, 16, 2), (33, 1, 1), (64, 3, 2)]:
    torch.manual_seed(n * 31 + m)
    Q, p, _, _, lb, ub = O.create_qp_data(n, B, seed=n + 7, with_eq=False)
    A = torch.randn(B, m, n) if m else None
    b = 0.05 * torch.randn(B, m, 1) if m else None
    for opts in (dict(), dict(scale=False, rho=2.0), dict(rho=0.7), dict(check_solved=1), dict(adaptive_rho=False)):
        for bounds in ("both", "lb", "ub"):
            lbx = lb if bounds in ("both", "lb") else torch.full_like(lb, -float("inf"))
            ubx = ub if bounds in ("both", "ub") else torch.full_like(ub, float("inf"))
            ctl = O.make_control(**TOL); ctl.update(opts)
            ref = O.solve_box_qp(Q, p, A, b, lbx, ubx, dict(ctl))
            a = [None if t is None else t.to(dev) for t in (Q, p, A, b, lbx, ubx)]
            its = {}
            for ls in ("lu", "spd"):
                c2 = dict(ctl); c2["linsolve"] = ls
                s = L.torch_solve_box_qp(*a, c2)
                its[ls] = s["iter"]
                agree[ls] += int(s["iter"] == ref["iter"])
            total += 1
            if its["lu"] != ref["iter"] or its["spd"] != ref["iter"]:
                print(f"n={n} m={m} opts={list(opts)} bounds={bounds}: oracle {ref['iter']} lu {its['lu']} spd {its['spd']}", flush=True)
print(f"{total} cases: lu agrees with the oracle's iteration count in {agree['lu']}, spd in {agree['spd']}")
