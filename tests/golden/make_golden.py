#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE (ipo-lab/lqp_py) on CPU.

Runs only in the build container, where the reference is mounted read-only at
/root/reference; the GPU box never sees the reference, only the ``.npz`` files
this script writes next to itself.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

Every case records which reference entry point produced it.  Inputs are drawn
with the reference's own generators (experiments/utils.py); the oracle's
restated generators are asserted bit-identical here so the GPU box can
regenerate large inputs without shipping them.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("LQP_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.path.insert(1, REPO)

from lqp_py.solve_box_qp_admm_torch import (SolveBoxQP, torch_solve_box_qp,   # noqa: E402
                                            torch_solve_box_qp_grad)
from lqp_py.control import box_qp_control                                     # noqa: E402
from lqp_py.lu_layer import TorchLU                                           # noqa: E402
from lqp_py.solve_qp_eqcon_torch import (torch_solve_qp_eqcon,                # noqa: E402
                                         torch_solve_qp_eqcon_grad)
from lqp_py.solve_qp_uncon_torch import (torch_solve_qp_uncon,                # noqa: E402
                                         torch_solve_qp_uncon_grad)
from experiments.utils import create_qp_data, generate_hard_qp_torch         # noqa: E402

from oracle import boxqp_oracle as O                                          # noqa: E402

# (thread count left at the torch default; forcing it oversubscribes batched LAPACK)


def npy(t):
    if t is None:
        return np.zeros(0, dtype=np.float32)
    if torch.is_tensor(t):
        return t.detach().cpu().numpy()
    return np.asarray(t)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: npy(v) for k, v in arrays.items()})
    print(f"wrote {name}.npz  ({os.path.getsize(path) / 1024:.0f} KiB)")


def ref_inputs(n, B, seed, with_eq=True, dtype=torch.float32):
    Q, p, A, b, lb, ub, _, _ = create_qp_data(n, B, 2 * n, seed=seed, requires_grad=False)
    if not with_eq:
        A = b = None
    got = O.create_qp_data(n, B, seed=seed, with_eq=with_eq)
    for r, g in zip((Q, p, A, b, lb, ub), got):
        assert (r is None and g is None) or torch.equal(r, g), "oracle generator drifted"
    cast = lambda t: None if t is None else t.to(dtype)
    return tuple(cast(t) for t in (Q, p, A, b, lb, ub))


def checksum(*ts):
    return np.array([float(t.double().sum()) for t in ts if t is not None])


def fp_grads(sol, Q, A, lb, ub, g):
    return torch_solve_box_qp_grad(g, sol["x"], sol["u"], sol["lams"], sol["nus"],
                                   Q, A, lb, ub, sol["rho"])


GRAD_NAMES = ("dQ", "dp", "dA", "db", "dlb", "dub")


def main():
    tol = dict(eps_abs=1e-5, eps_rel=1e-5)
    global save
    only = set(sys.argv[1:])
    if only:                     # regenerate a subset: everything is computed, only named files written
        _save = save
        save = lambda name, **kw: _save(name, **kw) if any(name.startswith(o) for o in only) else None

    # G16: above the 1024 rows the on-chip tiers of the build hold -- B=8 n=1500 m=1 (README.md:49 discusses n_x > 500;
    #      the reference's LAPACK calls take any size), forward + all six FP gradients for a random cotangent
    if not only or any(o.startswith("g16") for o in only):
        Q, p, A, b, lb, ub = ref_inputs(1500, 8, 0)
        sol = torch_solve_box_qp(Q, p, A, b, lb, ub, box_qp_control(**tol))
        torch.manual_seed(16)
        cot = torch.randn(8, 1500, 1)
        gr = fp_grads(sol, Q, A, lb, ub, cot)
        rs = np.random.RandomState(16)
        sb, si, sj = rs.randint(0, 8, 64), rs.randint(0, 1500, 64), rs.randint(0, 1500, 64)
        out = dict(cot=cot, in_sum=checksum(Q, p, lb, ub), sb=sb, si=si, sj=sj,
                   dQ_fro=torch.linalg.matrix_norm(gr[0]), dQ_samples=gr[0][sb, si, sj],
                   **{k: sol[k] for k in ("x", "z", "u", "lams", "nus", "rho", "iter")})
        for nm, t in zip(GRAD_NAMES[1:], gr[1:6]):
            out[nm] = t
        save("g16_b8_n1500_eq", **out)
        print("   n=1500 iter", sol["iter"])
        del Q, sol, gr

    # G17: unroll=True at a size where the loop runs on the two-level kernels -- B=8 n=100 m=1, tol 1e-5 (the published
    #      "ADMM Unroll" rows use the experiment_1 distribution); autograd through the loop, all six gradients
    if not only or any(o.startswith("g17") for o in only):
        Q, p, A, b, lb, ub = ref_inputs(100, 8, 17)
        torch.manual_seed(17)
        cot = torch.randn(8, 100, 1)
        ctl = box_qp_control(unroll=True, **tol)
        leaves = [t.clone().requires_grad_(True) for t in (Q, p, A, b, lb, ub)]
        xu = SolveBoxQP(control=ctl)(*leaves)
        xu.backward(cot)
        out = dict(Q=Q, p=p, A=A, b=b, lb=lb, ub=ub, cot=cot, x=xu.detach())
        for nm, t in zip(GRAD_NAMES, leaves):
            out[nm] = t.grad
        save("g17_unroll_n100", **out)

    # G18: the hard distribution at SURVEY 8(c)'s G11 size -- n=250, m=round(sqrt(250))=16, prob 0.85, seeds 0..127, fp64
    #      (experiments/utils.py:64-131, experiments/experiment_1_hard.py:13-35): the benched `b128_n250_m16_hard_fp64` row.
    #      Inputs are regenerated on the GPU box by the restated generator (asserted bit-identical here).
    if not only or any(o.startswith("g18") for o in only):
        seeds = list(range(128))
        Qh, ph, Ah, bh, lbh, ubh, _, _ = generate_hard_qp_torch(250, 0.85, seeds)
        got = O.create_hard_qp_data(250, 0.85, seeds)
        for r, g in zip((Qh, ph, Ah, bh, lbh, ubh), got):
            assert torch.equal(r.detach(), g), "oracle hard-QP generator drifted"
        Qh, ph, Ah, bh, lbh, ubh = [t.detach() for t in (Qh, ph, Ah, bh, lbh, ubh)]
        assert Ah.shape == (128, 16, 250) and Qh.dtype == torch.float64
        sol = torch_solve_box_qp(Qh, ph, Ah, bh, lbh, ubh, box_qp_control(**tol))
        torch.manual_seed(18)
        gh = torch.randn(128, 250, 1, dtype=torch.float64)
        gr = fp_grads(sol, Qh, Ah, lbh, ubh, gh)
        rs = np.random.RandomState(18)
        sb, si, sj = rs.randint(0, 128, 256), rs.randint(0, 250, 256), rs.randint(0, 250, 256)
        out = dict(cot=gh, in_sum=checksum(Qh, ph, Ah, bh, lbh, ubh), sb=sb, si=si, sj=sj,
                   dQ_fro=torch.linalg.matrix_norm(gr[0]), dQ_samples=gr[0][sb, si, sj],
                   **{k: sol[k] for k in ("x", "u", "nus", "rho", "iter")})
        for nm, t in zip(GRAD_NAMES[1:], gr[1:6]):
            out[nm] = t
        out["dA_fro"] = torch.linalg.matrix_norm(gr[2])      # (dA in full is 4 MB: its norm for every problem, the
        out["dA"] = gr[2][:8]                                 #  entries of the first eight)
        save("g18_hard_f64_n250_m16", **out)
        print("   hard n=250 iter", sol["iter"])
        del Qh, sol, gr

    # G19: the per-GPU shard of BASELINE configs[4] -- B=1024 n=500 m=1 seed 0, float32, forward (experiments/experiment_1.py:12-16
    #      with n_batch = 1024): x and the iteration count (decided by ALL 1024 problems, :312)
    if not only or any(o.startswith("g19") for o in only):
        Q, p, A, b, lb, ub = ref_inputs(500, 1024, 0)
        sol = torch_solve_box_qp(Q, p, A, b, lb, ub, box_qp_control(**tol))
        save("g19_b1024_n500_eq", in_sum=checksum(Q, p, lb, ub), x=sol["x"], u=sol["u"], rho=sol["rho"], iter=sol["iter"])
        print("   B=1024 n=500 iter", sol["iter"])
        del Q, sol
    # G20: above 2048 rows (README.md:49 discusses large n_x; the reference's LAPACK calls at :205-215 take any size) -- B=2 n=3000
    #      m=1 float32, forward + the FP gradients for a random cotangent; inputs regenerated on the GPU box by the restated generator
    if not only or any(o.startswith("g20") for o in only):
        Q, p, A, b, lb, ub = ref_inputs(3000, 2, 20)
        sol = torch_solve_box_qp(Q, p, A, b, lb, ub, box_qp_control(**tol))
        torch.manual_seed(20)
        cot = torch.randn(2, 3000, 1)
        gr = fp_grads(sol, Q, A, lb, ub, cot)
        rs = np.random.RandomState(20)
        sb, si, sj = rs.randint(0, 2, 64), rs.randint(0, 3000, 64), rs.randint(0, 3000, 64)
        out = dict(cot=cot, in_sum=checksum(Q, p, lb, ub), sb=sb, si=si, sj=sj,
                   dQ_fro=torch.linalg.matrix_norm(gr[0]), dQ_samples=gr[0][sb, si, sj],
                   **{k: sol[k] for k in ("x", "z", "u", "lams", "nus", "rho", "iter")})
        for nm, t in zip(GRAD_NAMES[1:], gr[1:6]):
            out[nm] = t
        save("g20_b2_n3000_eq", **out)
        print("   n=3000 iter", sol["iter"])
        del Q, sol, gr
    # G21: unroll=True in float64 with several equality rows -- the tape whose every node is TorchLULayer on the pivoted LU of the KKT
    #      matrix (lqp_py/lu_layer.py:25-58): B=4 n=60 m=3, tol 1e-8, the hard distribution's dtype; autograd through the loop, all six
    #      gradients.  Pins lqp_boxqp_unroll_backward_lu (round 6) against the reference itself.
    if not only or any(o.startswith("g21") for o in only):
        Q, p, A, b, lb, ub = ref_inputs(60, 4, 21, dtype=torch.float64)
        gen = torch.Generator().manual_seed(21)
        A = torch.randn(4, 3, 60, generator=gen, dtype=torch.float64)
        b = 0.1 * torch.randn(4, 3, 1, generator=gen, dtype=torch.float64)
        cot = torch.randn(4, 60, 1, generator=gen, dtype=torch.float64)
        ctl = box_qp_control(unroll=True, eps_abs=1e-8, eps_rel=1e-8)
        leaves = [t.clone().requires_grad_(True) for t in (Q, p, A, b, lb, ub)]
        xu = SolveBoxQP(control=ctl)(*leaves)
        xu.backward(cot)
        out = dict(Q=Q, p=p, A=A, b=b, lb=lb, ub=ub, cot=cot, x=xu.detach())
        for nm, t in zip(GRAD_NAMES, leaves):
            out[nm] = t.grad
        save("g21_unroll_f64_m3", **out)
    # G22: unroll=True through ONE adaptive-rho refactorisation (lqp_py/solve_box_qp_admm_torch.py:237-256 inside the tape: the
    #      reference's autograd runs through the adaptation itself) -- B=4 n=20 m=1 float32, Q x 50 with a given rho = 100 and no
    #      auto-scaling, tol 1e-6: the plain forward of the same problem reports the iteration count and the adapted rho.
    if not only or any(o.startswith("g22") for o in only):
        Q, p, A, b, lb, ub = ref_inputs(20, 4, 22)
        Q = Q * 50
        torch.manual_seed(22)
        cot = torch.randn(4, 20, 1)
        kw = dict(rho=100.0, scale=False, eps_abs=1e-6, eps_rel=1e-6)
        plain = torch_solve_box_qp(Q, p, A, b, lb, ub, box_qp_control(**kw))
        assert int(plain["iter"]) >= 100 and float((plain["rho"] - 100.0).abs().max()) > 1.0, (plain["iter"], plain["rho"])
        ctl = box_qp_control(unroll=True, **kw)
        leaves = [t.clone().requires_grad_(True) for t in (Q, p, A, b, lb, ub)]
        xu = SolveBoxQP(control=ctl)(*leaves)
        xu.backward(cot)
        out = dict(Q=Q, p=p, A=A, b=b, lb=lb, ub=ub, cot=cot, x=xu.detach(), iter=plain["iter"], rho=plain["rho"])
        for nm, t in zip(GRAD_NAMES, leaves):
            out[nm] = t.grad
        save("g22_unroll_rho_event", **out)
        print("   unroll with a rho event: iter", plain["iter"], "rho", plain["rho"].flatten().tolist())
    if only and all(o.startswith(("g16", "g17", "g18", "g19", "g20", "g21", "g22")) for o in only):
        return

    # G14: the NumPy twin (lqp_py/solve_box_qp_admm.py:45-91, single problem, float64) -- SURVEY 8f rank 4
    from lqp_py.solve_box_qp_admm import solve_box_qp as np_solve_box_qp
    out = {}
    for tag, (n, with_eq, ctl) in {"a": (10, False, dict(scale=False, adaptive_rho=False)),
                                   "b": (30, True, dict()),
                                   "c": (64, True, dict(rho=5.0, adaptive_rho_iter=20))}.items():
        Qt, pt, At, bt, lbt, ubt = ref_inputs(n, 1, 40 + n, with_eq=with_eq, dtype=torch.float64)
        Qn, pn = Qt[0].numpy().astype(np.float64), pt[0, :, 0].numpy().astype(np.float64)
        An = At[0].numpy().astype(np.float64) if with_eq else None
        bn = bt[0, :, 0].numpy().astype(np.float64) if with_eq else None
        lbn, ubn = lbt[0, :, 0].numpy().astype(np.float64), ubt[0, :, 0].numpy().astype(np.float64)
        c = box_qp_control(**tol)
        c.update(ctl)
        sol = np_solve_box_qp(Qn, pn, An, bn, lbn, ubn, c)
        out.update({f"{tag}_Q": Qn, f"{tag}_p": pn, f"{tag}_lb": lbn, f"{tag}_ub": ubn,
                    f"{tag}_A": An if with_eq else np.zeros((0, n)), f"{tag}_b": bn if with_eq else np.zeros(0)})
        for k2 in ("x", "z", "u", "lam", "rho", "primal_error", "dual_error", "iter"):
            out[f"{tag}_{k2}"] = np.asarray(sol[k2])
        out[f"{tag}_nu"] = np.asarray(sol["nu"]) if sol["nu"] is not None else np.zeros(0)
        print("   numpy twin", tag, "iter", sol["iter"])
    save("g14_numpy_twin", **out)

    # G15: OptNet with equality constraints only (lqp_py/optnet.py:8-54 -> torch_solve_qp_eqcon / _grad)
    from lqp_py.optnet import OptNet
    from lqp_py.control import optnet_control
    torch.manual_seed(15)
    Qo, po, Ao, bo, _, _ = ref_inputs(20, 4, 15)
    leaves = [t.clone().requires_grad_(True) for t in (Qo, po, Ao, bo)]
    xo = OptNet(control=optnet_control())(leaves[0], leaves[1], leaves[2], leaves[3], None, None)
    coto = torch.randn_like(xo)
    xo.backward(coto)
    save("g15_optnet_eq", Q=Qo, p=po, A=Ao, b=bo, cot=coto, x=xo, dQ=leaves[0].grad, dp=leaves[1].grad,
         dA=leaves[2].grad, db=leaves[3].grad)
    if only and all(o.startswith(("g14", "g15")) for o in only):
        return

    # G1: BASELINE config 1 -- B=32 n=10 box-only, lb=-1 ub=1 (demo_solve_box_qp_torch.py:19-20)
    Q, p, _, _, _, _ = ref_inputs(10, 32, 0, with_eq=False)
    lb, ub = -torch.ones(32, 10, 1), torch.ones(32, 10, 1)
    sol = torch_solve_box_qp(Q, p, None, None, lb, ub, box_qp_control(**tol))
    save("g1_b32_n10_box", Q=Q, p=p, lb=lb, ub=ub, x=sol["x"], z=sol["z"], u=sol["u"],
         lams=sol["lams"], rho=sol["rho"], iter=sol["iter"])

    # G2: B=8 n=50 m=1, full dict + all six FP grads for ones and randn cotangents
    Q, p, A, b, lb, ub = ref_inputs(50, 8, 0)
    sol = torch_solve_box_qp(Q, p, A, b, lb, ub, box_qp_control(**tol))
    torch.manual_seed(123)
    g_rand = torch.randn(8, 50, 1)
    g_ones = torch.ones(8, 50, 1)
    out = dict(Q=Q, p=p, A=A, b=b, lb=lb, ub=ub, g_rand=g_rand,
               **{k: sol[k] for k in ("x", "z", "u", "lams", "nus", "rho", "iter")})
    for tag, g in (("ones", g_ones), ("rand", g_rand)):
        for nm, t in zip(GRAD_NAMES, fp_grads(sol, Q, A, lb, ub, g)):
            out[f"{nm}_{tag}"] = t
    save("g2_b8_n50_eq", **out)

    # G8: fp64 repeat of G2
    Qd, pd, Ad, bd, lbd, ubd = ref_inputs(50, 8, 0, dtype=torch.float64)
    sol = torch_solve_box_qp(Qd, pd, Ad, bd, lbd, ubd, box_qp_control(**tol))
    out = {k: sol[k] for k in ("x", "z", "u", "lams", "nus", "rho", "iter")}
    for nm, t in zip(GRAD_NAMES, fp_grads(sol, Qd, Ad, lbd, ubd, g_rand.double())):
        out[f"{nm}_rand"] = t
    save("g8_b8_n50_eq_f64", **out)

    # G3: BASELINE config 2 -- B=128 n=100 box-only
    Q, p, _, _, lb, ub = ref_inputs(100, 128, 0, with_eq=False)
    sol = torch_solve_box_qp(Q, p, None, None, lb, ub, box_qp_control(**tol))
    save("g3_b128_n100_box", x=sol["x"], u=sol["u"], rho=sol["rho"], iter=sol["iter"],
         in_sum=checksum(Q, p, lb, ub))

    # G4: BASELINE config 3 (headline) -- B=128 n=500 m=1 via the nn.Module + autograd
    Q, p, A, b, lb, ub = ref_inputs(500, 128, 0)
    sol = torch_solve_box_qp(Q, p, A, b, lb, ub, box_qp_control(**tol))
    grads = fp_grads(sol, Q, A, lb, ub, torch.ones(128, 500, 1))
    torch.manual_seed(7)
    g_rand = torch.randn(128, 500, 1)
    grads_r = fp_grads(sol, Q, A, lb, ub, g_rand)
    rs = np.random.RandomState(0)
    sb, si, sj = rs.randint(0, 128, 64), rs.randint(0, 500, 64), rs.randint(0, 500, 64)
    out = dict(x=sol["x"], u=sol["u"], z=sol["z"], lams=sol["lams"], nus=sol["nus"], rho=sol["rho"], iter=sol["iter"],
               in_sum=checksum(Q, p, lb, ub), sb=sb, si=si, sj=sj)
    for tag, gr in (("ones", grads), ("rand", grads_r)):
        dQ = gr[0]
        out[f"dQ_fro_{tag}"] = torch.linalg.matrix_norm(dQ)
        out[f"dQ_samples_{tag}"] = dQ[sb, si, sj]
        for nm, t in zip(GRAD_NAMES[1:], gr[1:6]):
            out[f"{nm}_{tag}"] = t
    save("g4_b128_n500_eq", **out)
    # module/autograd path must agree with the functional path on the same inputs
    Qg = Q.clone().requires_grad_(True)
    pg = p.clone().requires_grad_(True)
    xm = SolveBoxQP(control=box_qp_control(**tol))(Qg, pg, A, b, lb, ub)
    xm.backward(torch.ones(128, 500, 1))
    assert torch.equal(xm.detach(), sol["x"]) and torch.equal(pg.grad, grads[1])

    # G5: BASELINE config 4 -- B=128 n=1000 m=1 forward only
    Q, p, A, b, lb, ub = ref_inputs(1000, 128, 0)
    sol = torch_solve_box_qp(Q, p, A, b, lb, ub, box_qp_control(**tol))
    save("g5_b128_n1000_eq", x=sol["x"], rho=sol["rho"], iter=sol["iter"],
         in_sum=checksum(Q, p, lb, ub))
    del Q, p, A, b, lb, ub, sol

    # G6: adaptive-rho stress -- B=16 n=50 m=1, Q*50, rho=100 (refactorisation fires)
    Q, p, A, b, lb, ub = ref_inputs(50, 16, 3)
    Q = Q * 50
    for tag, sc in (("noscale", False), ("scale", True)):
        ctl = box_qp_control(rho=100.0, scale=sc, **tol)
        sol = torch_solve_box_qp(Q, p, A, b, lb, ub, ctl)
        grads = fp_grads(sol, Q, A, lb, ub, g_rand[:16, :50])
        out = dict(Q=Q, p=p, A=A, b=b, lb=lb, ub=ub, g=g_rand[:16, :50],
                   **{k: sol[k] for k in ("x", "z", "u", "lams", "nus", "rho", "iter")})
        for nm, t in zip(GRAD_NAMES, grads):
            out[nm] = t
        save(f"g6_adaptive_{tag}", **out)
        print("   adaptive iter", sol["iter"], "rho range",
              float(sol["rho"].min()), float(sol["rho"].max()))

    # G7: no inequality (lb=-inf, ub=inf): rho=0 path through the autograd layer,
    #     plus the caller-dict mutation (solve_box_qp_admm_torch.py:37-38)
    Q, p, A, b, _, _ = ref_inputs(20, 4, 5)
    lbi, ubi = torch.full((4, 20, 1), -float("inf")), torch.full((4, 20, 1), float("inf"))
    ctl = box_qp_control(**tol)
    Qg = Q.clone().requires_grad_(True)
    x = SolveBoxQP(control=ctl)(Qg, p, A, b, lbi, ubi)
    assert ctl["rho"] == 0
    sol = torch_solve_box_qp(Q, p, A, b, lbi, ubi, box_qp_control(**tol))
    save("g7_noineq", Q=Q, p=p, A=A, b=b, x=sol["x"], u=sol["u"], lams=sol["lams"],
         nus=sol["nus"], iter=sol["iter"], x_layer=x.detach(), rho_after=np.array(ctl["rho"]))

    # G9: TorchLU fwd/bwd on random SPD + eq-constrained / unconstrained QP
    torch.manual_seed(11)
    R = torch.randn(3, 6, 6)
    S = R @ R.transpose(1, 2) + 6 * torch.eye(6)
    rhs = torch.randn(3, 6, 2)
    Sg, rg = S.clone().requires_grad_(True), rhs.clone().requires_grad_(True)
    lu = TorchLU(A=S)
    y = lu(Sg, rg)
    gy = torch.randn(3, 6, 2)
    y.backward(gy)
    LU, P = torch.linalg.lu_factor(S)
    Q, p, A, b, _, _ = ref_inputs(20, 4, 2)
    es = torch_solve_qp_eqcon(Q, p, A, b)
    gz = torch.randn(4, 20, 1)
    eg = torch_solve_qp_eqcon_grad(gz, es["x"], es["nus"], Q, A)
    us = torch_solve_qp_uncon(Q, p)
    ug = torch_solve_qp_uncon_grad(gz, us["x"], Q)
    save("g9_lu_eqcon", S=S, rhs=rhs, LU=LU, P=P, y=y.detach(), gy=gy, dS=Sg.grad, drhs=rg.grad,
         Q=Q, p=p, A=A, b=b, gz=gz, eq_x=es["x"], eq_nus=es["nus"],
         eq_dQ=eg[0], eq_dp=eg[1], eq_dA=eg[2], eq_db=eg[3],
         un_x=us["x"], un_dQ=ug[0], un_dp=ug[1])

    # G10: scalar rho input.  (a) rho=0.001 with the factory defaults: adaptive rho fires and the
    #      returned rho becomes a (B,1,1) tensor; (b) rho=1.0, adaptive_rho=False: stays a float.
    Q, p, A, b, lb, ub = ref_inputs(30, 4, 4)
    sa = torch_solve_box_qp(Q, p, A, b, lb, ub, box_qp_control(rho=0.001, **tol))
    sb_ = torch_solve_box_qp(Q, p, A, b, lb, ub, box_qp_control(rho=1.0, adaptive_rho=False, **tol))
    assert torch.is_tensor(sa["rho"]) and not torch.is_tensor(sb_["rho"])
    save("g10_scalar_rho", Q=Q, p=p, A=A, b=b, lb=lb, ub=ub,
         a_x=sa["x"], a_u=sa["u"], a_lams=sa["lams"], a_nus=sa["nus"], a_iter=sa["iter"], a_rho=sa["rho"],
         b_x=sb_["x"], b_u=sb_["u"], b_lams=sb_["lams"], b_nus=sb_["nus"], b_iter=sb_["iter"],
         b_rho=np.array(float(sb_["rho"])))
    print("   scalar-rho iters", sa["iter"], sb_["iter"])

    # G12: backward='kkt' through the autograd layer (solve_box_qp_admm_torch.py:435-584), fp32
    Q, p, A, b, lb, ub = ref_inputs(40, 6, 8)
    torch.manual_seed(12)
    cot = torch.randn(6, 40, 1)
    leaves = [t.clone().requires_grad_(True) for t in (Q, p, A, b, lb, ub)]
    xk = SolveBoxQP(control=box_qp_control(backward='kkt', **tol))(*leaves)
    xk.backward(cot)
    out = dict(Q=Q, p=p, A=A, b=b, lb=lb, ub=ub, cot=cot, x=xk.detach())
    for nm, t in zip(GRAD_NAMES, leaves):
        out[nm] = t.grad
    # box-only variant (no equality block)
    leaves2 = [t.clone().requires_grad_(True) for t in (Q, p, lb, ub)]
    xk2 = SolveBoxQP(control=box_qp_control(backward='kkt', **tol))(leaves2[0], leaves2[1], None, None, leaves2[2], leaves2[3])
    xk2.backward(cot)
    out.update(x_box=xk2.detach(), dQ_box=leaves2[0].grad, dp_box=leaves2[1].grad, dlb_box=leaves2[2].grad, dub_box=leaves2[3].grad)
    save("g12_kkt_backward", **out)

    # G13: unroll=True -- autograd through the ADMM loop (:14-15, 216-219, 264-265 + lu_layer.py)
    Q, p, A, b, lb, ub = ref_inputs(20, 4, 9)
    torch.manual_seed(13)
    cot = torch.randn(4, 20, 1)
    ctl = box_qp_control(unroll=True, eps_abs=1e-6, eps_rel=1e-6)
    leaves = [t.clone().requires_grad_(True) for t in (Q, p, A, b, lb, ub)]
    xu = SolveBoxQP(control=ctl)(*leaves)
    xu.backward(cot)
    out = dict(Q=Q, p=p, A=A, b=b, lb=lb, ub=ub, cot=cot, x=xu.detach())
    for nm, t in zip(GRAD_NAMES, leaves):
        out[nm] = t.grad
    save("g13_unroll", **out)

    # G11: hard sparse QP distribution, fp64, n=100 m=10 (experiments/utils.py:64-131)
    seeds = list(range(8))
    Qh, ph, Ah, bh, lbh, ubh, _, _ = generate_hard_qp_torch(100, 0.85, seeds)
    got = O.create_hard_qp_data(100, 0.85, seeds)
    for r, g in zip((Qh, ph, Ah, bh, lbh, ubh), got):
        assert torch.equal(r.detach(), g), "oracle hard-QP generator drifted"
    Qh, ph, Ah, bh, lbh, ubh = [t.detach() for t in (Qh, ph, Ah, bh, lbh, ubh)]
    sol = torch_solve_box_qp(Qh, ph, Ah, bh, lbh, ubh, box_qp_control(**tol))
    torch.manual_seed(9)
    gh = torch.randn(8, 100, 1, dtype=torch.float64)
    grads = fp_grads(sol, Qh, Ah, lbh, ubh, gh)
    out = dict(g=gh, **{k: sol[k] for k in ("x", "z", "u", "lams", "nus", "rho", "iter")})
    for nm, t in zip(GRAD_NAMES, grads):
        out[nm] = t
    save("g11_hard_f64", **out)
    print("   hard iter", sol["iter"])


if __name__ == "__main__":
    main()
