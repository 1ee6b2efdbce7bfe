#!/usr/bin/env python3
"""Randomised sweep of round 5's multi-workgroup paths against their one-workgroup forms (same inputs, A/B by environment switch):
wide LU (bits), dense loop on W workgroups (iterates), two-workgroup unroll sweep (gradients).  Usage: gpu_round5_sweep.py [seed]"""
import os, sys, random
os.environ.setdefault("LQP_ENV_NOCACHE", "1")
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import lqp_py_amd as L
from lqp_py_amd import lu_layer
from oracle import boxqp_oracle as O
dev = torch.device("cuda:0")
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0

def ab(name, fn):
    out = {}
    for flag in ("1", "0"):
        os.environ[name] = flag
        out[flag] = fn()
    os.environ.pop(name, None)
    return out["1"], out["0"]

for _ in range(8):                                   # wide LU
    N, B, dt = rnd.randint(1025, 2048), rnd.randint(1, 12), rnd.choice([torch.float32, torch.float32, torch.float64])
    if dt == torch.float64: B = min(B, 4)
    A = torch.randn(B, N, N, dtype=dt, generator=torch.Generator().manual_seed(N)).to(dev)
    (LU1, P1), (LU0, P0) = ab("LQP_LU_WIDE", lambda: lu_layer.lu_factor(A))
    ok = torch.equal(P1, P0) and torch.equal(LU1, LU0)
    bad += not ok
    print(f"wide LU      N={N:4d} B={B:2d} {str(dt)[6:]:8s} bits equal: {ok}", flush=True)
for _ in range(8):                                   # dense loop on W workgroups
    n, B, m, dt = rnd.randint(257, 1300), rnd.randint(1, 8), rnd.choice([0, 1, 3, 20]), rnd.choice([torch.float32, torch.float64])
    if dt == torch.float64: n = min(n, 1000)
    Q, p, A, b, lb, ub = O.create_qp_data(n, B, seed=n)
    if m == 0: A = b = None
    elif m > 1:
        A = torch.randn(B, m, n, generator=torch.Generator().manual_seed(n + 1)); b = A @ (0.5 * (lb + ub))
    inp = [None if t is None else t.to(dt).to(dev) for t in (Q, p, A, b, lb, ub)]
    ctl = dict(O.make_control(eps_abs=1e-5, eps_rel=1e-5), linsolve="lu")
    s1, s0 = ab("LQP_LOOP_DENSE_W", lambda: L.torch_solve_box_qp(*inp, dict(ctl)))
    e = float((s1["x"] - s0["x"]).abs().max()); tol = 1e-9 if dt == torch.float64 else 5e-5
    ok = s1["iter"] == s0["iter"] and e < tol * max(1.0, float(s0["x"].abs().max()))
    bad += not ok
    print(f"dense loop   n={n:4d} B={B:2d} m={m:2d} {str(dt)[6:]:8s} W={s1['_stats']['loop_workgroups']:2d} iter {s1['iter']} / {s0['iter']}  |dx| {e:.1e}: {ok}", flush=True)
for _ in range(6):                                   # two-workgroup unroll sweep
    n, B, m = rnd.randint(257, 512), rnd.randint(1, 8), rnd.choice([0, 1, 2])
    Q, p, A, b, lb, ub = O.create_qp_data(n, B, seed=n)
    if m == 0: A = b = None
    elif m > 1:
        A = torch.randn(B, m, n, generator=torch.Generator().manual_seed(n + 1)); b = A @ (0.5 * (lb + ub))
    cot = torch.randn(B, n, 1, generator=torch.Generator().manual_seed(n + 2)).to(dev)
    def run():
        lv = [None if t is None else t.clone().to(dev).requires_grad_(True) for t in (Q, p, A, b, lb, ub)]
        L.SolveBoxQP(control=L.box_qp_control(unroll=True, eps_abs=1e-5, eps_rel=1e-5))(*lv).backward(cot)
        return [None if t is None else t.grad for t in lv]
    g1, g0 = ab("LQP_UNROLL_SPLIT", run)
    worst = max(float((a - c).abs().max()) / max(1e-3, float(c.abs().max())) for a, c in zip(g1, g0) if c is not None)
    ok = worst < 1e-4
    bad += not ok
    print(f"unroll sweep n={n:4d} B={B:2d} m={m} worst relative gradient difference {worst:.1e}: {ok}", flush=True)
print("ALL OK" if bad == 0 else f"{bad} MISMATCHES")
