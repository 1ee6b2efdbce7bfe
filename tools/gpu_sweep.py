#!/usr/bin/env python3
"""Robustness sweep: symmetric-inverse vs LU x-update on odd shapes / control combinations, then timings of the
other BASELINE configurations (forward-only and forward+backward)."""
import os, sys, time, itertools
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import lqp_py_amd as LA
import lqp_py_amd.solve_box_qp_admm_torch as L
from lqp_py_amd.synthetic import create_qp_data
from lqp_py_amd.control import box_qp_control

dev = torch.device("cuda:0")
TOL = dict(eps_abs=1e-5, eps_rel=1e-5)
worst = 0.0
cases = 0
for (n, m, B) in [(1, 0, 1), (2, 1, 3), (3, 0, 2), (7, 2, 5), (33, 1, 1), (63, 0, 2), (64, 3, 2), (65, 1, 2), (127, 5, 3),
                  (129, 0, 2), (255, 16, 2), (257, 1, 2), (500, 1, 9), (511, 2, 1), (512, 16, 3)]:
    torch.manual_seed(n * 31 + m)
    Q, p, _, _, lb, ub = create_qp_data(n, B, seed=n + 7, with_eq=False)
    A = torch.randn(B, m, n) if m else None
    b = 0.05 * torch.randn(B, m, 1) if m else None
    for opts in (dict(), dict(scale=False), dict(adaptive_rho=False), dict(rho=0.7), dict(scale=False, rho=2.0),
                 dict(rho=torch.linspace(0.5, 1.5, B).view(B, 1, 1)), dict(check_solved=1), dict(max_iters=37)):
        for bounds in ("both", "lb", "ub"):
            lbx = lb if bounds in ("both", "lb") else torch.full_like(lb, -float("inf"))
            ubx = ub if bounds in ("both", "ub") else torch.full_like(ub, float("inf"))
            a = [None if t is None else t.to(dev) for t in (Q, p, A, b, lbx, ubx)]
            s = {}
            for ls in ("lu", "spd"):
                ctl = box_qp_control(linsolve=ls, **TOL)
                ctl.update(opts)
                s[ls] = L.torch_solve_box_qp(*a, ctl)
            assert s["spd"]["_stats"]["linsolve_used"] == 2 and s["lu"]["_stats"]["linsolve_used"] == 1
            d = max(float((s["lu"][k] - s["spd"][k]).abs().max()) for k in ("x", "z", "u", "lams") + (("nus",) if m else ()))
            it = (s["lu"]["iter"], s["spd"]["iter"])
            cases += 1
            worst = max(worst, d)
            if d > 5e-5 or it[0] != it[1]:
                print(f"MISMATCH n={n} m={m} B={B} opts={ {k: (v if not torch.is_tensor(v) else 'tensor') for k, v in opts.items()} } bounds={bounds}: diff {d:.2e} iters {it}", flush=True)
print(f"sweep: {cases} cases, worst spd-vs-lu difference {worst:.2e}", flush=True)

def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

for (n, B, eq, bwd) in [(100, 128, False, False), (1000, 128, True, False), (500, 1024, True, True), (500, 16, True, True),
                        (500, 64, True, True), (250, 128, True, True)]:
    inp = [None if t is None else t.to(dev) for t in create_qp_data(n, B, seed=0, with_eq=eq)]
    for ls in ("lu", "auto"):
        ctl = LA.box_qp_control(linsolve=ls, **TOL)
        layer = LA.SolveBoxQP(control=ctl)
        ones = torch.ones(B, n, 1, device=dev)
        def step():
            Qg = inp[0].detach().requires_grad_(bwd)
            x = layer(Qg, *inp[1:])
            if bwd: x.backward(ones)
        dt = timeit(step)
        print(f"n={n} B={B} eq={eq} {'fwd+bwd' if bwd else 'fwd'} linsolve={ls}: {dt*1e3:.3f} ms/step, {B/dt:,.0f} QPs/s", flush=True)
LA.synchronize()
