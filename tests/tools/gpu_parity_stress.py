"""Randomised parity sweep of the float16-pipe paths (forward sweep: every K, two / four workgroups, turns; backward Cholesky)
against the float64 oracle at a pinned iteration count.  Prints the worst cases."""
import os, sys, itertools, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["LQP_ENV_NOCACHE"] = "1"
import lqp_py_amd as L
from oracle import boxqp_oracle as O
dev = torch.device("cuda:0")
torch.manual_seed(0)
worst = []
cases = []
for n in (129, 190, 257, 320, 333, 400, 449, 500, 512):
    for B in (1, 3, 17, 64, 130):
        for m in (0, 1, 3, 16):
            cases.append((n, B, m))
import random
random.seed(int(os.environ.get("SEED", "1")))
random.shuffle(cases)
cases = cases[:int(os.environ.get("NCASES", "60"))]
t_start = time.time()
for idx, (n, B, m) in enumerate(cases):
    kind = int(os.environ.get("KIND", idx % 4))
    Q, p, _, _, lb, ub = O.create_qp_data(n, B, seed=1000 + idx + 7919 * int(os.environ.get("SEED", "1")), with_eq=False)
    g = torch.Generator().manual_seed(idx)
    A = torch.randn(B, m, n, generator=g) if m else None
    b = 0.1 * torch.randn(B, m, 1, generator=g) if m else None
    kw = {}
    if kind == 1:
        Q = Q * (10.0 ** (6 * torch.rand(B, 1, 1, generator=g) - 3))        # badly scaled problems (auto-scaling on)
    elif kind == 2:
        kw = dict(scale=False, rho=0.05, adaptive_rho=False)
    elif kind == 3:
        d = 10.0 ** (3 * torch.rand(B, n, generator=g) - 1.5)              # row / column scaling: D Q D
        Q = d.unsqueeze(2) * Q * d.unsqueeze(1)
    iters = 40
    if kind == 4:
        d = 10.0 ** (2 * torch.rand(B, n, generator=g) - 1.0)
        Q = d.unsqueeze(2) * Q * d.unsqueeze(1)
        kw = dict(scale=False)
    if kind == 5:
        iters = 260
    ctl = O.make_control(eps_abs=1e-12, eps_rel=1e-12, max_iters=iters + 1, **kw)
    inp = (Q, p, A, b, lb, ub)
    t64 = O.solve_box_qp(*[None if t is None else t.double() for t in inp], dict(ctl))
    cot = torch.randn(B, n, 1, generator=g)
    g64 = O.solve_box_qp_grad(cot.double(), t64["x"], t64["u"], t64["lams"], t64["nus"], Q.double(), None if A is None else A.double(), lb.double(), ub.double(), t64["rho"])
    args = [None if t is None else t.to(dev) for t in inp]
    leaves = [args[0].clone().requires_grad_(True), args[1].clone().requires_grad_(True)]
    layer = L.SolveBoxQP(control=dict(ctl))
    x = layer(leaves[0], leaves[1], *args[2:])
    x.backward(cot.to(dev))
    st = L.solve_box_qp_admm_torch.last_forward_status(dev)
    sx = max(1.0, float(t64["x"].abs().max()))
    ex = float((x.detach().cpu().double() - t64["x"]).abs().max()) / sx
    sp = float(g64[1].abs().max()) + 1e-30
    ep = float((leaves[1].grad.cpu().double() - g64[1]).abs().max()) / sp
    sq = float(g64[0].abs().max()) + 1e-30
    eq = float((leaves[0].grad.cpu().double() - g64[0]).abs().max()) / sq
    bad = (not torch.isfinite(x).all()) or ex > 1e-5 or ep > 1e-4 or eq > 1e-4
    rec = (ex, ep, eq, n, B, m, kind, st["linsolve_used"], st["loop_workgroups_per_qp"], st["iters"])
    worst.append(rec)
    if bad:
        os.environ["LQP_SPD_F16"] = "0"; os.environ["LQP_ENV_NOCACHE"] = "1"
        x0 = L.SolveBoxQP(control=dict(ctl))(*args)
        os.environ["LQP_SPD_F16"] = "1"
        xl = L.SolveBoxQP(control=dict(ctl, linsolve="lu"))(*args)
        e0 = float((x0.detach().cpu().double() - t64["x"]).abs().max()) / sx
        el = float((xl.detach().cpu().double() - t64["x"]).abs().max()) / sx
        print("    same case: float32 sweep x %.2e   pivoted LU path x %.2e" % (e0, el), flush=True)
    if bad or idx % 10 == 0:
        print(("BAD " if bad else "ok  ") + "n %4d B %4d m %2d kind %d: x %.2e dp %.2e dQ %.2e  linsolve %d wg %d iters %d" % (n, B, m, kind, ex, ep, eq, st["linsolve_used"], st["loop_workgroups_per_qp"], st["iters"]), flush=True)
print("cases", len(worst), "time %.0f s" % (time.time() - t_start))
for key, nm in ((0, "x"), (1, "dp"), (2, "dQ")):
    w = max(worst, key=lambda r: r[key])
    print("worst", nm, "%.2e" % w[key], "at n %d B %d m %d kind %d linsolve %d" % (w[3], w[4], w[5], w[6], w[7]))
