#!/usr/bin/env python3
"""Random shapes through the layer (forward + backward) against the CPU oracle at a fixed iteration count.
usage: python tests/tools/gpu_fuzz.py [cases] [seed]   (FUZZ_BIG_B=1: batches above the two-workgroup limit; FUZZ_CONVERGE=1: run to the
tolerance with a rho that forces refactorisations).  Test infrastructure: the oracle is the checker."""
import os, sys, random, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import lqp_py_amd as L
from oracle import boxqp_oracle as O
dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = {}
t0 = time.time()
for c in range(cases):
    n = rng.choice([3, 17, 64, 65, 130, 257, 300, 321, 384, 400, 449, 500, 512, 520, 600, 700])
    m = rng.choice([0, 1, 1, 2, 3, 7, 16]) if n > 20 else rng.choice([0, 1])
    B = rng.choice([1, 2, 3, 5]) if os.environ.get("FUZZ_BIG_B") is None else rng.choice([129, 200, 260])
    its = rng.choice([21, 41])
    Q, p, _, _, lb, ub = O.create_qp_data(n, B, seed=1000 + c, with_eq=False)
    g = torch.Generator().manual_seed(c)
    A = torch.randn(B, m, n, generator=g) if m else None
    b = 0.1 * torch.randn(B, m, 1, generator=g) if m else None
    kw = dict(max_iters=its, eps_abs=1e-12, eps_rel=1e-12)
    if os.environ.get("FUZZ_CONVERGE"):       # run to the tolerance instead, with a rho that forces refactorisations
        kw = dict(eps_abs=1e-5, eps_rel=1e-5, rho=rng.choice([None, 100.0, 0.01]))
    ref = O.solve_box_qp(Q, p, A, b, lb, ub, O.make_control(**kw))
    args = [None if t is None else t.to(dev) for t in (Q, p, A, b, lb, ub)]
    Qg = args[0].clone().requires_grad_(True); pg = args[1].clone().requires_grad_(True)
    x = L.SolveBoxQP(control=L.box_qp_control(**kw))(Qg, pg, *args[2:])
    cot = torch.randn(B, n, 1, generator=g)
    x.backward(cot.to(dev))
    gr = O.solve_box_qp_grad(cot, ref["x"], ref["u"], ref["lams"], ref["nus"], Q, A, lb, ub, ref["rho"])
    scale = max(1.0, float(ref["x"].abs().max()))
    ex = float((x.detach().cpu() - ref["x"]).abs().max()) / scale
    # the active set is decided by x + u against the bounds (reference :360-365): a problem with an element within
    # rounding of a bound may be masked differently here and there -- its gradient is not comparable
    w = ref["x"] + ref["u"]
    okb = (torch.minimum((w - ub).abs(), (w - lb).abs()).amin(dim=(1, 2)) > 1e-5)
    gs = max(1.0, float(gr[1].abs().max()))
    ep = float((pg.grad.cpu() - gr[1])[okb].abs().max()) / gs if okb.any() else 0.0
    qs = max(1.0, float(gr[0].abs().max()))
    eq = float((Qg.grad.cpu() - gr[0])[okb].abs().max()) / qs if okb.any() else 0.0
    if os.environ.get("FUZZ_CONVERGE"):       # (the stop may fall on another check: tolerance-level agreement)
        bad = not (ex < 1e-3) or not torch.isfinite(x).all()
    else:
        bad = not (ex < 5e-5 and ep < 2e-3 and eq < 2e-3) or not torch.isfinite(x).all()
    print("%s n=%4d m=%2d B=%d its=%d  x %.1e  dp %.1e  dQ %.1e" % ("BAD" if bad else "ok ", n, m, B, its, ex, ep, eq), flush=True)
    worst[(n, m)] = max(worst.get((n, m), 0.0), ex)
print("done in %.0f s" % (time.time() - t0))
