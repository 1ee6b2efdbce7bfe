"""Pin the CPU oracle against golden vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only; runs in the build container and on
the GPU box's host."""
import numpy as np
import pytest
import torch

from oracle import boxqp_oracle as O
from conftest import load_golden

TOL = dict(eps_abs=1e-5, eps_rel=1e-5)
GRADS = ("dQ", "dp", "dA", "db", "dlb", "dub")


def close(a, b, atol=2e-6, rtol=2e-6):
    if a is None or b is None:
        assert a is None and b is None
        return
    torch.testing.assert_close(a, b.to(a.dtype), atol=atol, rtol=rtol)


def test_control_resolution_traps():
    c = O.make_control(check_solved=3, adaptive_rho_max_iter=7)
    r = O.resolve_control(c, 500)
    assert r.check_solved == 20            # factory writes the misspelt key
    assert r.adaptive_rho_max_iter == 1000  # solver reads 'adaptive_max_iter'
    assert r.adaptive_rho_iter == 100
    assert O.resolve_control({}, 1000).adaptive_rho_iter == 90
    assert [O.default_check_interval(n) for n in (10, 50, 100, 250, 500, 1000)] == [1, 10, 10, 20, 20, 30]
    e = O.resolve_control({}, 10)
    assert (e.adaptive_rho, e.adaptive_rho_tol, e.scale) == (False, 5, False)


def test_g1_box_only():
    g = load_golden("g1_b32_n10_box")
    sol = O.solve_box_qp(g["Q"], g["p"], None, None, g["lb"], g["ub"], O.make_control(**TOL))
    assert sol["iter"] == g["iter"]
    for k in ("x", "z", "u", "lams", "rho"):
        close(sol[k], g[k])


@pytest.mark.parametrize("name,dtype", [("g2_b8_n50_eq", torch.float32), ("g8_b8_n50_eq_f64", torch.float64)])
def test_g2_g8_forward_and_fp_grads(name, dtype):
    base = load_golden("g2_b8_n50_eq")
    g = load_golden(name)
    Q, p, A, b, lb, ub = (base[k].to(dtype) for k in ("Q", "p", "A", "b", "lb", "ub"))
    sol = O.solve_box_qp(Q, p, A, b, lb, ub, O.make_control(**TOL))
    assert sol["iter"] == g["iter"]
    tol = 2e-6 if dtype == torch.float32 else 1e-12
    for k in ("x", "z", "u", "lams", "nus", "rho"):
        close(sol[k], g[k], tol, tol)
    tags = ("ones", "rand") if dtype == torch.float32 else ("rand",)
    for tag in tags:
        cot = torch.ones(8, 50, 1, dtype=dtype) if tag == "ones" else base["g_rand"].to(dtype)
        grads = O.solve_box_qp_grad(cot, sol["x"], sol["u"], sol["lams"], sol["nus"], Q, A, lb, ub, sol["rho"])
        for nm, t in zip(GRADS, grads):
            close(t, g[f"{nm}_{tag}"], tol * 10, tol * 10)


def test_g3_config2():
    g = load_golden("g3_b128_n100_box")
    Q, p, _, _, lb, ub = O.create_qp_data(100, 128, seed=0, with_eq=False)
    sol = O.solve_box_qp(Q, p, None, None, lb, ub, O.make_control(**TOL))
    assert sol["iter"] == g["iter"] == 70
    close(sol["x"], g["x"], 2e-5, 1e-5)
    close(sol["rho"], g["rho"], 1e-5, 1e-5)


@pytest.mark.parametrize("tag", ["noscale", "scale"])
def test_g6_adaptive_rho_refactor(tag):
    g = load_golden(f"g6_adaptive_{tag}")
    tr = {}
    ctl = O.make_control(rho=100.0, scale=(tag == "scale"), **TOL)
    sol = O.solve_box_qp(g["Q"], g["p"], g["A"], g["b"], g["lb"], g["ub"], ctl, trace=tr)
    assert sol["iter"] == g["iter"] == 100
    assert tr["n_factor"] == 2
    for k in ("x", "z", "u", "lams", "nus", "rho"):
        close(sol[k], g[k], 1e-5, 1e-5)
    grads = O.solve_box_qp_grad(g["g"], sol["x"], sol["u"], sol["lams"], sol["nus"],
                                g["Q"], g["A"], g["lb"], g["ub"], sol["rho"])
    for nm, t in zip(GRADS, grads):
        close(t, g[nm], 1e-4, 1e-4)


def test_g7_no_inequality_mutates_control():
    g = load_golden("g7_noineq")
    n, B = 20, 4
    lb = torch.full((B, n, 1), -float("inf"))
    ub = torch.full((B, n, 1), float("inf"))
    ctl = O.make_control(**TOL)
    sol = O.layer_forward(g["Q"], g["p"], g["A"], g["b"], lb, ub, ctl)
    assert ctl["rho"] == 0 and sol["iter"] == g["iter"] == 0
    for k in ("x", "u", "lams", "nus"):
        close(sol[k], g[k])
    direct = O.solve_qp_eqcon(g["Q"], g["p"], g["A"], g["b"])
    close(direct["x"], g["x"], 1e-5, 1e-5)


def test_g9_lu_layer_eqcon_uncon():
    g = load_golden("g9_lu_eqcon")
    LU, P = O.lu_factor(g["S"])
    close(LU, g["LU"])
    assert torch.equal(P, g["P"])
    y = O.lu_solve(LU, P, g["rhs"])
    close(y, g["y"])
    dA, db = O.lu_layer_backward(LU, P, y, g["gy"])
    close(dA, g["dS"])
    close(db, g["drhs"])
    es = O.solve_qp_eqcon(g["Q"], g["p"], g["A"], g["b"])
    close(es["x"], g["eq_x"])
    close(es["nus"], g["eq_nus"])
    eg = O.solve_qp_eqcon_grad(g["gz"], es["x"], es["nus"], g["Q"], g["A"])
    for t, k in zip(eg, ("eq_dQ", "eq_dp", "eq_dA", "eq_db")):
        close(t, g[k])
    us = O.solve_qp_uncon(g["Q"], g["p"])
    close(us["x"], g["un_x"])
    ug = O.solve_qp_uncon_grad(g["gz"], us["x"], g["Q"])
    close(ug[0], g["un_dQ"])
    close(ug[1], g["un_dp"])


def test_g10_scalar_rho_types():
    g = load_golden("g10_scalar_rho")
    args = [g[k] for k in ("Q", "p", "A", "b", "lb", "ub")]
    sa = O.solve_box_qp(*args, O.make_control(rho=0.001, **TOL))       # adaptive rho fires -> tensor
    assert torch.is_tensor(sa["rho"]) and sa["iter"] == g["a_iter"]
    close(sa["rho"], g["a_rho"], 1e-5, 1e-5)
    sb = O.solve_box_qp(*args, O.make_control(rho=1.0, adaptive_rho=False, **TOL))
    assert not torch.is_tensor(sb["rho"]) and sb["rho"] == float(g["b_rho"]) == 1.0
    assert sb["iter"] == g["b_iter"]
    for tag, s in (("a", sa), ("b", sb)):
        for k in ("x", "u", "lams", "nus"):
            close(s[k], g[f"{tag}_{k}"], 1e-5, 1e-5)


def test_g11_hard_distribution_fp64():
    g = load_golden("g11_hard_f64")
    Q, p, A, b, lb, ub = O.create_hard_qp_data(100, 0.85, list(range(8)))
    sol = O.solve_box_qp(Q, p, A, b, lb, ub, O.make_control(**TOL))
    assert sol["iter"] == g["iter"]
    for k in ("x", "z", "u", "lams", "nus", "rho"):
        close(sol[k], g[k], 1e-9, 1e-9)
    grads = O.solve_box_qp_grad(g["g"], sol["x"], sol["u"], sol["lams"], sol["nus"], Q, A, lb, ub, sol["rho"])
    for nm, t in zip(GRADS, grads):
        close(t, g[nm], 1e-7, 1e-7)


def test_g18_hard_distribution_fp64_at_the_benched_size():
    """G18 = SURVEY 8(c)'s G11 at its own size (n = 250, m = 16, seeds 0..127, float64): the oracle against the reference's run."""
    g = load_golden("g18_hard_f64_n250_m16")
    Q, p, A, b, lb, ub = O.create_hard_qp_data(250, 0.85, list(range(128)))
    sol = O.solve_box_qp(Q, p, A, b, lb, ub, O.make_control(**TOL))
    assert sol["iter"] == g["iter"] == 60
    for k in ("x", "u", "nus", "rho"):
        close(sol[k], g[k], 1e-9, 1e-9)
    grads = O.solve_box_qp_grad(g["cot"], sol["x"], sol["u"], sol["lams"], sol["nus"], Q, A, lb, ub, sol["rho"])
    for nm in ("dp", "db", "dlb", "dub"):
        close(grads[GRADS.index(nm)], g[nm], 1e-7, 1e-7)
    close(grads[2][:8], g["dA"], 1e-7, 1e-7)
    close(torch.linalg.matrix_norm(grads[0]), g["dQ_fro"], 1e-7, 1e-7)


def test_g20_above_2048_rows():
    """G20 (n = 3000, m = 1, B = 2, float32: above the 2048 rows that bounded the build until round 6): the oracle against the
    reference's run -- the same torch.linalg calls in the same order, so the same iteration count and iterates to float32 rounding."""
    g = load_golden("g20_b2_n3000_eq")
    Q, p, A, b, lb, ub = O.create_qp_data(3000, 2, seed=20)
    sol = O.solve_box_qp(Q, p, A, b, lb, ub, O.make_control(**TOL))
    assert sol["iter"] == g["iter"] == 50
    for k in ("x", "u", "nus", "rho"):
        close(sol[k], g[k], 1e-5, 1e-5)
    grads = O.solve_box_qp_grad(g["cot"], sol["x"], sol["u"], sol["lams"], sol["nus"], Q, A, lb, ub, sol["rho"])
    for nm in ("dp", "db", "dlb", "dub"):
        close(grads[GRADS.index(nm)], g[nm], 1e-4, 1e-4)


class _CpuLU(torch.nn.Module):
    """The taped solve of the unrolled loop on the CPU (lqp_py/lu_layer.py:5-58 restated in the oracle): forward lu_solve with the
    cached factor, backward dA = -lu_solve(g) x^T, db = lu_solve(g) with the SAME factor."""

    class _Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, A, b, LU, piv):
            x = O.lu_solve(LU, piv, b)
            ctx.save_for_backward(LU, piv, x)
            return x

        @staticmethod
        def backward(ctx, g):
            LU, piv, x = ctx.saved_tensors
            dA, db = O.lu_layer_backward(LU, piv, x, g)
            return dA, db, None, None

    def __init__(self, A):
        super().__init__()
        with torch.no_grad():
            self.LU, self.piv = O.lu_factor(A)

    def forward(self, A, b):
        return self._Fn.apply(A, b, self.LU, self.piv)


def test_g21_unroll_tape_float64():
    """G21 (unroll=True, float64, three equality rows, reference-made): the host-side restatement of the taped loop
    (lqp_py_amd.unrolled._eager_unrolled, the form the GPU tests use as the float64 arbiter of every unroll gradient) with the oracle's
    LU layer against the reference's autograd: solution and all six gradients."""
    from lqp_py_amd.unrolled import _eager_unrolled
    from lqp_py_amd import solve_box_qp_admm_torch as SB
    import lqp_py_amd as L
    g = load_golden("g21_unroll_f64_m3")
    leaves = [g[k].clone().requires_grad_(True) for k in ("Q", "p", "A", "b", "lb", "ub")]
    ctl = L.box_qp_control(unroll=True, eps_abs=1e-8, eps_rel=1e-8)
    x = _eager_unrolled(*leaves, SB.resolve_control(ctl, 60), True, True, solver_cls=_CpuLU)
    x.backward(g["cot"])
    close(x.detach(), g["x"], 1e-10, 1e-10)
    for nm, t in zip(GRADS, leaves):
        close(t.grad, g[nm], 1e-8, 1e-8)


def test_kkt_conditions_known_answer():
    """Independent of the reference: returned (x, lams, nus) satisfy the KKT system."""
    Q, p, A, b, lb, ub = O.create_qp_data(40, 6, seed=21, dtype=torch.float64)
    sol = O.solve_box_qp(Q, p, A, b, lb, ub, O.make_control(eps_abs=1e-9, eps_rel=1e-9))
    res = O.kkt_residuals(Q, p, A, b, lb, ub, sol)
    for k, v in res.items():
        assert float(v.max()) < 1e-6, (k, v)


@pytest.mark.parametrize("tag,ctl", [("a", dict(scale=False, adaptive_rho=False)), ("b", dict()),
                                     ("c", dict(rho=5.0, adaptive_rho_iter=20))])
def test_g14_numpy_twin_is_the_same_algorithm(tag, ctl):
    """The reference's NumPy solver (solve_box_qp_admm.py) and its torch solver are one algorithm: the oracle of the
    torch path, run on a batch of one in float64, reproduces the NumPy goldens."""
    g = load_golden("g14_numpy_twin")
    t = lambda k: g[f"{tag}_{k}"].double()
    n = t("p").shape[0]
    has_eq = g[f"{tag}_A"] is not None
    c = O.make_control(**TOL)
    c.update(ctl)
    sol = O.solve_box_qp(t("Q").reshape(1, n, n), t("p").reshape(1, n, 1),
                         t("A").reshape(1, -1, n) if has_eq else None, t("b").reshape(1, -1, 1) if has_eq else None,
                         t("lb").reshape(1, n, 1), t("ub").reshape(1, n, 1), c)
    assert sol["iter"] == int(g[f"{tag}_iter"])
    for k, kk in (("x", "x"), ("z", "z"), ("u", "u"), ("lams", "lam")):
        close(sol[k].reshape(-1), t(kk), 1e-10, 1e-10)
