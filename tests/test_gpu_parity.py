"""Parity of the HIP path (through the C ABI, via the Python shim) against the
CPU oracle and the golden vectors made from the reference.  GPU only.

Tolerances (north_star: residuals to 1e-5 with eps_abs = eps_rel = 1e-5, gradients
rtol 1e-4): fp32 iterates within 1e-5 of the reference-made golden vectors,
gradients within 1e-4 of the gradient's scale; fp64 to 1e-9.  Where fp32
rounding alone exceeds that (the reference's own fp32 result is no closer to an
fp64 solve of the same inputs), the fp64 criterion applies instead:
|HIP - fp64| <= |reference fp32 golden - fp64| + eps.  Every measured error goes to
gpurun_out/parity_report.json (tests/parity_report.py).
"""
import os

import numpy as np
import pytest
import torch

import lqp_py_amd as L
from lqp_py_amd import _lib, lu_layer
from oracle import boxqp_oracle as O
from conftest import load_golden
import parity_report as P
import lqp_py_amd.solve_box_qp_admm_torch as SB

pytestmark = pytest.mark.gpu
TOL = dict(eps_abs=1e-5, eps_rel=1e-5)
GRADS = ("dQ", "dp", "dA", "db", "dlb", "dub")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    _lib.load()                       # fails loudly if the HIP library is missing
    return torch.device("cuda:0")


def err(a, b):
    return float((a.detach().cpu().double() - b.detach().cpu().double()).abs().max())


def rel(a, b):
    b = b.detach().cpu().double()
    return err(a, b) / (float(b.abs().max()) + 1e-300)


X_TOL, G_RTOL = 1e-5, 1e-4           # north_star


def close_or_fp64(case, name, hip, golden, truth64, tol, **tags):
    """north_star tolerance against the golden vector, or -- when fp32 rounding alone is larger than that -- no
    further from an fp64 solve of the same inputs than the reference's own fp32 result (+ tol).  Records both."""
    scale = max(1.0, float(golden.abs().max()))
    e_direct = err(hip, golden)
    e_hip64 = e_ref64 = None
    if truth64 is not None:
        e_hip64, e_ref64 = err(hip, truth64), err(golden, truth64)
    P.record(case, name, e_direct, scale, hip_vs_fp64=e_hip64, reference_fp32_vs_fp64=e_ref64, tol=tol * scale, **tags)
    ok = e_direct <= tol * scale or (e_hip64 is not None and e_hip64 <= e_ref64 + tol * scale)
    assert ok, (case, name, tags, dict(direct=e_direct, hip_vs_fp64=e_hip64, ref_vs_fp64=e_ref64, tol=tol * scale))


def fp64_truth(inp, iters, cots=()):
    """The same solve in float64 with the iteration count pinned (eps tiny, max_iters = iters + 1): what is left
    between this and an fp32 result is rounding.  Returns (solution dict, [gradient tuples])."""
    d = [None if t is None else t.double() for t in inp]
    sol = O.solve_box_qp(*d, O.make_control(eps_abs=1e-12, eps_rel=1e-12, max_iters=iters + 1))
    grads = [O.solve_box_qp_grad(c.double(), sol["x"], sol["u"], sol["lams"], sol["nus"], d[0], d[2], d[4], d[5], sol["rho"])
             for c in cots]
    return sol, grads


def solve(dev, inputs, ctl):
    args = [None if t is None else t.to(dev) for t in inputs]
    return L.torch_solve_box_qp(*args, dict(ctl)), args


# ---------------------------------------------------------------- LU / solves
@pytest.mark.parametrize("N,B,dtype", [(6, 3, torch.float32), (64, 2, torch.float32), (65, 2, torch.float32),
                                       (130, 3, torch.float32), (200, 2, torch.float64)])
def test_lu_factor_matches_lapack(dev, N, B, dtype):
    torch.manual_seed(N)
    A = torch.randn(B, N, N, dtype=dtype)
    LUr, Pr = torch.linalg.lu_factor(A)
    LU, P = lu_layer.lu_factor(A.to(dev))
    assert torch.equal(P.cpu(), Pr)
    tol = 2e-4 if dtype == torch.float32 else 1e-10
    assert rel(LU, LUr) < tol
    rhs = torch.randn(B, N, 3, dtype=dtype)
    xr = torch.linalg.lu_solve(LUr, Pr, rhs)
    stol = 2e-3 if dtype == torch.float32 else 1e-9
    assert rel(lu_layer.lu_solve(LUr.to(dev), Pr.to(dev), rhs.to(dev)), xr) < stol     # torch factor, HIP solve
    assert rel(lu_layer.lu_solve(LU, P, rhs.to(dev)), xr) < stol                         # HIP factor, HIP solve


@pytest.mark.parametrize("N,B,dtype", [(97, 8, torch.float32), (130, 5, torch.float32), (317, 8, torch.float32), (501, 16, torch.float32),
                                       (512, 3, torch.float32), (100, 8, torch.float64), (266, 16, torch.float64), (501, 4, torch.float64)])
def test_two_workgroup_lu(dev, monkeypatch, N, B, dtype):
    """csrc/lqp_lu2.hpp (N <= 512, 2 B workgroups resident): column blocks dealt out to two workgroups, panels handed over
    inside the launch.  Same pivots as the one-workgroup kernel and, in float32, the SAME BITS (every element sees the same
    operations in the same order); float64 pivots are LAPACK's; a zero column is reported at the same index; on and off one XCD."""
    torch.manual_seed(N + B)
    G = torch.randn(B, N, N, dtype=torch.float64)
    A = (G.transpose(1, 2) @ G / N + 0.3 * torch.eye(N, dtype=torch.float64))
    A[:, N - 5:, N - 5:] = 0                       # a KKT-like zero block: pivoting is needed
    A[:, N - 5:, :N - 5] = G[:, :5, :N - 5]; A[:, :N - 5, N - 5:] = G[:, :5, :N - 5].transpose(1, 2)
    A = A.to(dtype)
    monkeypatch.setenv("LQP_LU2", "0")
    LU1, P1 = lu_layer.lu_factor(A.to(dev))
    monkeypatch.setenv("LQP_LU2", "1")
    prof = _lib.profile(enable=True, reset=True)
    LU2, P2 = lu_layer.lu_factor(A.to(dev))
    _lib.profile(enable=False)
    # float32: the same bits (matrix-core trailing update and substitution in the same order in both kernels); float64: the
    # one-workgroup kernel's trailing update runs on the vector unit and its U12 by substitution, this one's on
    # v_mfma_f64_16x16x4 with the inverse of L11 -- the same factorisation to rounding
    same = (lambda X, Y: torch.equal(X, Y)) if dtype == torch.float32 else (lambda X, Y: rel(X, Y) < 1e-12)
    assert torch.equal(P1, P2) and same(LU1, LU2)
    monkeypatch.setenv("LQP_XCD_LOCAL", "0")       # write-through publication (workgroups on different XCDs take this path)
    LU3, P3 = lu_layer.lu_factor(A.to(dev))
    assert torch.equal(P2, P3) and torch.equal(LU2, LU3)
    LUr, Pr = torch.linalg.lu_factor(A)
    if dtype == torch.float64:
        assert torch.equal(P2.cpu(), Pr) and rel(LU2, LUr) < 1e-9
    Pm, Lm, Um = torch.lu_unpack(LU2.cpu().double(), P2.cpu())
    recon = float((Pm @ Lm @ Um - A.double()).abs().max()) / float(A.abs().max())
    assert recon < (1e-4 if dtype == torch.float32 else 1e-12), recon
    Z = A.clone(); Z[:, :, N // 2] = 0
    for flag in ("0", "1"):
        monkeypatch.setenv("LQP_LU2", flag)
        with pytest.raises(RuntimeError, match=rf"U\[{N // 2 + 1},{N // 2 + 1}\] is zero"):
            lu_layer.lu_factor(Z.to(dev))


@pytest.mark.parametrize("N,B,dtype", [(70, 3, torch.float32), (266, 8, torch.float32), (317, 4, torch.float32), (512, 2, torch.float32),
                                       (64, 2, torch.float64), (266, 8, torch.float64), (300, 3, torch.float64), (501, 2, torch.float64),
                                       # float32 above 576 rows: 16-column tiles on v_mfma_f32_16x16x4 (Y would not fit the LDS otherwise)
                                       (700, 2, torch.float32), (1100, 2, torch.float32), (1501, 1, torch.float32)])
def test_inverse_from_the_packed_factor(dev, N, B, dtype):
    """csrc/lqp_dense.hpp, k_lu_inverse: X = M^-1 from the packed factor (blocked triangular solves with N right-hand sides on
    the matrix cores) against torch.linalg.inv of a KKT-like matrix (zero lower-right block: the pivoting matters)."""
    torch.manual_seed(N)
    m = max(1, N // 16)
    n = N - m
    G = torch.randn(B, n, n, dtype=torch.float64)
    A = torch.zeros(B, N, N, dtype=torch.float64)
    A[:, :n, :n] = G.transpose(1, 2) @ G / n + torch.eye(n, dtype=torch.float64)
    Aeq = torch.randn(B, m, n, dtype=torch.float64)
    A[:, n:, :n] = Aeq; A[:, :n, n:] = Aeq.transpose(1, 2)
    Ad = A.to(dtype).to(dev)
    LU, P = lu_layer.lu_factor(Ad)
    buf = lu_layer._PackedFactor(LU, P).buffer()
    X = torch.empty_like(Ad)
    lib = _lib.load()
    _lib.check(lib.lqp_debug_lu_inverse(_lib.stream_ptr(dev), _lib.dtype_code(Ad), B, N, _lib.ptr(buf), _lib.ptr(X)), "lu_inverse")
    torch.cuda.synchronize()
    ref = torch.linalg.inv(A)
    tol = (2e-4 if N <= 512 else 1e-3) if dtype == torch.float32 else 1e-10
    assert rel(X, ref.to(dtype)) < tol
    eye = torch.eye(N, dtype=torch.float64)
    assert float((A @ X.cpu().double() - eye).abs().max()) < ((5e-4 if N <= 512 else 3e-3) if dtype == torch.float32 else 1e-10)


@pytest.mark.parametrize("N,B,dtype", [(1100, 3, torch.float32), (1500, 8, torch.float32), (2048, 2, torch.float32), (1030, 5, torch.float32),
                                       (1100, 2, torch.float64), (1501, 4, torch.float64), (2048, 1, torch.float64),
                                       (2049, 2, torch.float32), (3001, 2, torch.float32), (4096, 1, torch.float32)])
def test_lu_wide_matches_one_workgroup(dev, monkeypatch, N, B, dtype):
    """csrc/lqp_lu_wide.hpp: above 1024 rows (float32 to 4096: k_lu_factor_wide_tall, sixteen panel rows per thread), the batch leaving the chip idle: the pivoted LU on W = #CUs / B workgroups per
    matrix (32-column tiles owned cyclically, the factored panel handed round as a message) against the one-workgroup kernel
    (LQP_LU_WIDE=0): the same pivots; float32: the arithmetic per element being the same, the same factor bit for bit; float64 (the
    trailing update on v_mfma_f64_16x16x4 where the one-workgroup kernel multiplies in registers): to rounding, pivots LAPACK's.
    P A = L U.  A singular matrix reports the same first zero pivot."""
    torch.manual_seed(N)
    A = torch.randn(B, N, N, dtype=dtype)
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("LQP_LU_WIDE", flag)
        LU, piv = lu_layer.lu_factor(A.to(dev))
        out[flag] = (LU.cpu(), piv.cpu())
    assert torch.equal(out["1"][1], out["0"][1])
    if dtype == torch.float32:
        assert torch.equal(out["1"][0], out["0"][0])
    else:
        assert rel(out["1"][0], out["0"][0]) < 1e-11
        assert torch.equal(out["1"][1], torch.linalg.lu_factor(A)[1])
    Pm, Lm, Um = torch.lu_unpack(out["1"][0].double(), out["1"][1])
    assert float((Pm @ Lm @ Um - A.double()).abs().max()) / float(A.abs().max()) < (1e-4 if dtype == torch.float32 else 1e-12)
    A[:, :, 700] = 0.0
    msgs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("LQP_LU_WIDE", flag)
        with pytest.raises(RuntimeError) as e:
            lu_layer.lu_factor(A.to(dev))
        msgs.append(str(e.value))
    assert msgs[0] == msgs[1] and "701" in msgs[0], msgs


@pytest.mark.parametrize("N,B,dtype", [(1025, 2, torch.float64), (1100, 2, torch.float32), (1501, 2, torch.float64),
                                       (2048, 1, torch.float32), (2048, 1, torch.float64), (2049, 2, torch.float32),
                                       (3001, 2, torch.float32), (4096, 1, torch.float32)])
def test_lu_factor_above_1024(dev, N, B, dtype):
    """1024 < N <= 2048: two panel rows per thread (k_lu_factor_big), the matrix in global memory; float32 to 4096: four rows per
    thread, panels of four columns.  float64: the pivots
    ARE LAPACK's; float32 (a near-tie may flip under another summation order at this size): P A = L U to rounding,
    |L| <= 1 (partial pivoting), and the solves."""
    torch.manual_seed(N)
    A = torch.randn(B, N, N, dtype=dtype)
    LUr, Pr = torch.linalg.lu_factor(A)
    LU, piv = lu_layer.lu_factor(A.to(dev))
    LUc, Pc = LU.cpu(), piv.cpu()
    if dtype == torch.float64:
        assert torch.equal(Pc, Pr)
        assert rel(LU, LUr) < 1e-9
    Pm, Lm, Um = torch.lu_unpack(LUc.double(), Pc)
    recon = float((Pm @ Lm @ Um - A.double()).abs().max()) / float(A.abs().max())
    assert recon < (1e-4 if dtype == torch.float32 else 1e-12), recon
    assert float(torch.tril(LUc, -1).abs().max()) <= 1.0 + 1e-6
    # the solves: residual of A x = rhs no worse than LAPACK's own in the same precision (a random dense matrix of this
    # size has cond ~ 1e4-1e5: the float32 residual is ~1e-2 for either)
    rhs = torch.randn(B, N, 2, dtype=dtype)
    resid = lambda x: float((A.double() @ x.cpu().double() - rhs.double()).abs().max())
    r_lapack = resid(torch.linalg.lu_solve(LUr, Pr, rhs))
    r_hip = resid(lu_layer.lu_solve(LU, piv, rhs.to(dev)))                             # HIP factor, HIP solve
    r_mix = resid(lu_layer.lu_solve(LUr.to(dev), Pr.to(dev), rhs.to(dev)))           # torch factor, HIP solve
    P.record(f"lu_above_1024_N{N}_{'f32' if dtype == torch.float32 else 'f64'}", "residual", r_hip, lapack=r_lapack, mixed=r_mix)
    assert r_hip <= 3 * r_lapack + 1e-9 and r_mix <= 3 * r_lapack + 1e-9, (r_hip, r_mix, r_lapack)


@pytest.mark.parametrize("mfma", ["0", "1"])
@pytest.mark.parametrize("n,dtype", [(500, torch.float32), (1000, torch.float32), (500, torch.float64)])
def test_lu_kkt_sized(dev, n, dtype, mfma, monkeypatch):
    monkeypatch.setenv("LQP_LU_MFMA", mfma)
    B = 3
    Q, p, A, b, lb, ub = O.create_qp_data(n, B, seed=1, dtype=dtype)
    M = O.kkt_matrix(Q + 1.2 * torch.eye(n, dtype=dtype), A)
    LUr, Pr = torch.linalg.lu_factor(M)
    LU, P = lu_layer.lu_factor(M.to(dev))
    assert torch.equal(P.cpu(), Pr)
    assert rel(LU, LUr) < (2e-4 if dtype == torch.float32 else 1e-10)


def test_lu_singular_raises(dev):
    A = torch.randn(2, 8, 8)
    A[1, :, 3] = 0
    with pytest.raises(RuntimeError, match="is zero"):
        lu_layer.lu_factor(A.to(dev))


def test_g9_lu_layer_eqcon_uncon(dev):
    g = load_golden("g9_lu_eqcon")
    S, rhs, gy = (g[k].to(dev) for k in ("S", "rhs", "gy"))
    Sg, rg = S.clone().requires_grad_(True), rhs.clone().requires_grad_(True)
    lu = L.TorchLU(A=S)
    assert torch.equal(lu.P.cpu(), g["P"]) and err(lu.LU, g["LU"]) < 1e-5
    y = lu(Sg, rg)
    y.backward(gy)
    assert err(y, g["y"]) < 1e-5 and err(Sg.grad, g["dS"]) < 1e-5 and err(rg.grad, g["drhs"]) < 1e-5
    # a factor made by torch works too (interchangeable LAPACK layout)
    lu2 = L.TorchLU(LU=g["LU"].to(dev), P=g["P"].to(dev))
    assert err(lu2(S, rhs), g["y"]) < 1e-5
    Q, p, A, b, gz = (g[k].to(dev) for k in ("Q", "p", "A", "b", "gz"))
    s = L.torch_solve_qp_eqcon(Q, p, A, b)
    assert err(s["x"], g["eq_x"]) < 2e-5 and err(s["nus"], g["eq_nus"]) < 2e-5
    for t, k in zip(L.torch_solve_qp_eqcon_grad(gz, s["x"], s["nus"], Q, A), ("eq_dQ", "eq_dp", "eq_dA", "eq_db")):
        assert err(t, g[k]) < 5e-5, k
    u = L.torch_solve_qp_uncon(Q, p)
    assert err(u["x"], g["un_x"]) < 1e-4
    ug = L.torch_solve_qp_uncon_grad(gz, u["x"], Q)
    assert err(ug[0], g["un_dQ"]) < 1e-3 and err(ug[1], g["un_dp"]) < 1e-3


# ---------------------------------------------------------------- forward / backward vs goldens
def test_g1_config1_box_only(dev):
    g = load_golden("g1_b32_n10_box")
    sol, _ = solve(dev, (g["Q"], g["p"], None, None, g["lb"], g["ub"]), O.make_control(**TOL))
    assert sol["iter"] == g["iter"] and sol["nus"] is None
    for k in ("x", "z", "u", "lams", "rho"):
        P.record("g1_b32_n10_box", k, err(sol[k], g[k]))
        assert err(sol[k], g[k]) < X_TOL, k


@pytest.mark.parametrize("linsolve", ["lu", "spd"])
@pytest.mark.parametrize("mode", [1, 2])
def test_g2_forward_and_all_fp_grads(dev, mode, linsolve):
    g = load_golden("g2_b8_n50_eq")
    ctl = O.make_control(launch_mode=mode, linsolve=linsolve, **TOL)
    sol, a = solve(dev, tuple(g[k] for k in ("Q", "p", "A", "b", "lb", "ub")), ctl)
    assert sol["iter"] == g["iter"] and sol["_stats"]["mode_used"] == mode
    for k in ("x", "z", "u", "lams", "nus", "rho"):
        P.record("g2_b8_n50_eq", k, err(sol[k], g[k]), linsolve=linsolve, mode=mode)
        assert err(sol[k], g[k]) < X_TOL, k
    for tag, cot in (("ones", torch.ones(8, 50, 1)), ("rand", g["g_rand"])):
        gr = L.torch_solve_box_qp_grad(cot.to(dev), sol["x"], sol["u"], sol["lams"], sol["nus"], a[0], a[2], a[4], a[5], sol["rho"])
        assert gr[6] is None
        for nm, t in zip(GRADS, gr):
            scale = max(1.0, float(g[f"{nm}_{tag}"].abs().max()))
            P.record("g2_b8_n50_eq", f"{nm}_{tag}", err(t, g[f"{nm}_{tag}"]), scale, linsolve=linsolve, mode=mode)
            assert err(t, g[f"{nm}_{tag}"]) < G_RTOL * scale, (tag, nm)


def test_g8_fp64(dev):
    base, g = load_golden("g2_b8_n50_eq"), load_golden("g8_b8_n50_eq_f64")
    sol, a = solve(dev, tuple(base[k].double() for k in ("Q", "p", "A", "b", "lb", "ub")), O.make_control(**TOL))
    assert sol["iter"] == g["iter"]
    for k in ("x", "z", "u", "lams", "nus", "rho"):
        assert err(sol[k], g[k]) < 1e-9, k
    gr = L.torch_solve_box_qp_grad(base["g_rand"].double().to(dev), sol["x"], sol["u"], sol["lams"], sol["nus"],
                                   a[0], a[2], a[4], a[5], sol["rho"])
    for nm, t in zip(GRADS, gr):
        assert rel(t, g[f"{nm}_rand"]) < 1e-7, nm


@pytest.mark.parametrize("sync", [True, False])
@pytest.mark.parametrize("tag", ["noscale", "scale"])
def test_adaptive_rho_refactorises_in_float64(dev, tag, sync):
    """The G6 inputs in float64 (pivoted-LU path): rho = 100 forces a refactorisation at iteration 100.  Since round 5 the
    continuation kernel refactorises in-kernel for float64 as well (no separate rho-update / LU / pack launches per possible
    event).  Against the float64 oracle: same iteration count, one refactorisation, iterates to 1e-9."""
    g = load_golden(f"g6_adaptive_{tag}")
    d = [g[k].double() for k in ("Q", "p", "A", "b", "lb", "ub")]
    kw = dict(rho=100.0, scale=(tag == "scale"), **TOL)
    ref = O.solve_box_qp(*d, O.make_control(**kw))
    sol = L.torch_solve_box_qp(*[t.to(dev) for t in d], dict(O.make_control(**kw), sync=sync))
    L.synchronize()
    assert sol["x"].dtype == torch.float64
    if sync:
        assert sol["iter"] == ref["iter"] and sol["_stats"]["n_factor"] == 2 and sol["_stats"]["linsolve_used"] == 1
    for k in ("x", "z", "u", "lams", "nus"):
        assert err(sol[k], ref[k]) < 1e-9 * max(1.0, float(ref[k].abs().max())), k
    assert err(sol["rho"], ref["rho"]) < 1e-9 * float(ref["rho"].abs().max())


def _dense_case(name):
    """inputs (CPU), control extras, dtype of the dense-tier A/B cases"""
    from lqp_py_amd.synthetic import create_hard_qp_data
    if name == "hard_f64":                     # experiments/experiment_1_hard.py's distribution, a small batch of it
        return create_hard_qp_data(250, 0.85, range(6), dtype=torch.float64), {}
    if name == "box_only_f64":
        Q, p, A, b, lb, ub = O.create_qp_data(100, 5, seed=31)
        return tuple(t.double() for t in (Q, p)) + (None, None) + tuple(t.double() for t in (lb, ub)), {}
    if name == "largest_f64":                  # n = 256: every register column of the loop in use
        Q, p, A, b, lb, ub = O.create_qp_data(256, 3, seed=32)
        A8 = torch.randn(3, 8, 256, generator=torch.Generator().manual_seed(33))
        x0 = 0.5 * (lb + ub)
        return tuple(t.double() for t in (Q, p, A8, A8 @ x0, lb, ub)), {}
    if name == "lu_f32_ragged":                # control['linsolve'] = 'lu' in float32, n % 4 != 0, three equality rows
        Q, p, A, b, lb, ub = O.create_qp_data(37, 7, seed=34)
        A3 = torch.randn(7, 3, 37, generator=torch.Generator().manual_seed(35))
        x0 = 0.5 * (lb + ub)
        return (Q, p, A3, A3 @ x0, lb, ub), {"linsolve": "lu"}
    if name == "nonsymmetric_f64":             # what only the pivoted LU takes: Q + a skew part (x^T S x = 0: the same QP value)
        Q, p, A, b, lb, ub = O.create_qp_data(64, 4, seed=36)
        S = torch.randn(4, 64, 64, generator=torch.Generator().manual_seed(37))
        return tuple(t.double() for t in (Q + 0.05 * (S - S.transpose(1, 2)), p, A, b, lb, ub)), {}
    if name == "one_sided_f64":
        Q, p, A, b, lb, ub = O.create_qp_data(90, 4, seed=38)
        return tuple(t.double() for t in (Q, p, A, b, torch.full_like(lb, -float("inf")), ub)), {}
    raise KeyError(name)


@pytest.mark.parametrize("name", ["hard_f64", "box_only_f64", "largest_f64", "lu_f32_ragged", "nonsymmetric_f64", "one_sided_f64"])
def test_dense_loop_matches_cached_lu_loop(dev, monkeypatch, name):
    """csrc/lqp_dense.hpp: the LU path's loop with the explicit inverse in the registers of two workgroups (n <= 256) against the
    same solve on the cached triangular solves (LQP_LOOP_DENSE=0) and against the CPU oracle: the same iteration count, iterates
    to rounding.  The dense form is the one that ran (two loop workgroups per problem in the device's report)."""
    inp, extra = _dense_case(name)
    f64 = inp[0].dtype == torch.float64
    ctl = dict(O.make_control(**TOL), **extra)
    ref = O.solve_box_qp(*inp, O.make_control(**TOL))
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("LQP_LOOP_DENSE", flag)
        sol, _ = solve(dev, inp, ctl)
        st = sol["_stats"]
        assert st["linsolve_used"] == 1 and st["loop_workgroups"] == (2 if flag == "1" else 1), (flag, st)
        out[flag] = sol
    a, b_ = out["1"], out["0"]
    assert a["iter"] == b_["iter"] == ref["iter"], (a["iter"], b_["iter"], ref["iter"])
    tol = 1e-9 if f64 else 2e-5
    for k in ("x", "z", "u", "lams", "nus"):
        if ref[k] is None:
            continue
        scale = max(1.0, float(ref[k].abs().max()))
        assert err(a[k], b_[k]) < tol * scale, (k, err(a[k], b_[k]))
        assert err(a[k], ref[k]) < tol * scale, (k, err(a[k], ref[k]))


@pytest.mark.parametrize("n,B,m,dtype,extra", [(600, 4, 2, torch.float32, {"linsolve": "lu"}), (1200, 3, 1, torch.float32, {}),
                                               (500, 6, 5, torch.float64, {}), (1000, 2, 3, torch.float64, {}),
                                               (333, 5, 0, torch.float64, {})])
def test_dense_loop_on_many_workgroups(dev, monkeypatch, n, B, m, dtype, extra):
    """csrc/lqp_dense.hpp, k_admm_loop_dense_w: small batches on the LU path above n = 256 -- the explicit inverse (k_lu_inverse; float32
    above 576 rows on 16-column tiles) spread by rows over W workgroups per problem, x all-gathered per iteration -- against the
    cached triangular solves (LQP_LOOP_DENSE_W=0) and the CPU oracle: the same iteration count, iterates to rounding."""
    Q, p, A, b, lb, ub = O.create_qp_data(n, B, seed=60 + n)
    if m == 0:
        A = b = None
    elif m > 1:
        A = torch.randn(B, m, n, generator=torch.Generator().manual_seed(61))
        b = A @ (0.5 * (lb + ub))
    inp = tuple(None if t is None else t.to(dtype) for t in (Q, p, A, b, lb, ub))
    ctl = dict(O.make_control(**TOL), **extra)
    ref = O.solve_box_qp(*inp, O.make_control(**TOL))
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("LQP_LOOP_DENSE_W", flag)
        sol, _ = solve(dev, inp, ctl)
        st = sol["_stats"]
        assert st["linsolve_used"] == 1, st
        assert (st["loop_workgroups"] > 2) == (flag == "1"), (flag, st)
        out[flag] = sol
    a, c = out["1"], out["0"]
    assert a["iter"] == c["iter"] == ref["iter"], (a["iter"], c["iter"], ref["iter"])
    tol = 1e-9 if dtype == torch.float64 else 3e-5
    tol_ref = tol if (dtype == torch.float64 or n <= 1024) else 1e-4       # (float32 rounding of ~80 iterations at n = 1200, either side)
    for k in ("x", "z", "u", "lams", "nus"):
        if ref[k] is None:
            continue
        scale = max(1.0, float(ref[k].abs().max()))
        assert err(a[k], c[k]) < tol * scale, (k, err(a[k], c[k]))
        assert err(a[k], ref[k]) < tol_ref * scale, (k, err(a[k], ref[k]))


@pytest.mark.parametrize("n,B,dtype", [(600, 4, torch.float32), (1100, 3, torch.float32), (500, 5, torch.float64)])
def test_small_batch_tiers_pipelined_and_through_the_layer(dev, n, B, dtype):
    """The small-batch tiers of the LU path (wide LU above 1024 rows, dense loop on W workgroups) in a pipelined call
    (control['sync'] = False: the whole schedule enqueued, nothing waited for) and through the layer with its fixed-point backward:
    the same bits as the synchronous functional call, gradients against the oracle's."""
    Q, p, A, b, lb, ub = O.create_qp_data(n, B, seed=80 + n)
    inp = tuple(t.to(dtype) for t in (Q, p, A, b, lb, ub))
    extra = {"linsolve": "lu"} if n <= 1024 and dtype == torch.float32 else {}
    ctl = dict(O.make_control(**TOL), **extra)
    ref_sol, args = solve(dev, inp, ctl)
    assert ref_sol["_stats"]["loop_workgroups"] > 2
    piped = L.torch_solve_box_qp(*args, dict(ctl, sync=False))
    L.synchronize()
    for k in ("x", "z", "u", "lams", "nus"):
        assert torch.equal(piped[k], ref_sol[k]), k
    leaves = [t.clone().requires_grad_(True) for t in args]
    cot = torch.randn(B, n, 1, generator=torch.Generator().manual_seed(81), dtype=dtype).to(dev)
    x = L.SolveBoxQP(control=dict(L.box_qp_control(**TOL), **extra))(*leaves)
    assert torch.equal(x.detach(), ref_sol["x"])
    x.backward(cot)
    ref = O.solve_box_qp(*inp, O.make_control(**TOL))
    g = O.solve_box_qp_grad(cot.cpu(), ref["x"], ref["u"], ref["lams"], ref["nus"], inp[0], inp[2], inp[4], inp[5], ref["rho"])
    rt = 1e-8 if dtype == torch.float64 else 2e-3
    for nm, t, e in zip(GRADS, leaves, g):
        scale = max(1e-6, float(e.abs().max()))
        assert err(t.grad, e) <= rt * scale, (nm, err(t.grad, e), scale)


@pytest.mark.parametrize("n,B,dtype", [(300, 4, torch.float64), (1100, 2, torch.float64)])
def test_refactorisation_behind_the_small_batch_tiers(dev, n, B, dtype):
    """rho = 100 forces an adaptive-rho refactorisation at iteration 100 (as in G6): the first segment runs on the dense loop of W
    workgroups per problem (and, above 1024 rows, the factorisations on the wide LU, the second one through its device-side gate),
    the continuation on the cached triangular solves.  Against the oracle: the same iteration count, at least two factorisations, iterates to
    rounding.  (float64: in float32 the new rho -- a square root of a ratio of two residuals at the 1e-6 level -- is rounding noise
    amplified, and at n = 1100 the CPU oracle itself stops at 210 or 240 iterations depending on the host's BLAS threading.)"""
    Q, p, A, b, lb, ub = O.create_qp_data(n, B, seed=70 + n)
    inp = tuple(t.to(dtype) for t in (Q, p, A, b, lb, ub))
    kw = dict(rho=100.0, **TOL)
    ref = O.solve_box_qp(*inp, O.make_control(**kw))
    sol, _ = solve(dev, inp, O.make_control(**kw))
    st = sol["_stats"]
    assert st["linsolve_used"] == 1 and st["loop_workgroups"] > 2, st
    assert sol["iter"] == ref["iter"] and st["n_factor"] >= 2 and ref["iter"] >= 100, (sol["iter"], ref["iter"], st)
    if dtype == torch.float64:
        for k in ("x", "z", "u", "lams", "nus"):
            assert err(sol[k], ref[k]) < 1e-9 * max(1.0, float(ref[k].abs().max())), k
    else:
        t64 = O.solve_box_qp(*[t.double() for t in inp], O.make_control(rho=100.0, eps_abs=1e-12, eps_rel=1e-12, max_iters=ref["iter"] + 1))
        for k in ("x", "z", "lams", "nus"):
            close_or_fp64(f"refactor_small_batch_n{n}", k, sol[k], ref[k], t64[k], X_TOL)


@pytest.mark.parametrize("linsolve,mode", [("lu", 2), ("spd", 2), ("spd", 1)])
@pytest.mark.parametrize("tag", ["noscale", "scale"])
def test_g6_adaptive_rho_refactorises(dev, tag, linsolve, mode):
    g = load_golden(f"g6_adaptive_{tag}")
    ctl = O.make_control(rho=100.0, scale=(tag == "scale"), linsolve=linsolve, launch_mode=mode, **TOL)
    sol, a = solve(dev, tuple(g[k] for k in ("Q", "p", "A", "b", "lb", "ub")), ctl)
    assert sol["iter"] == g["iter"] == 100 and sol["_stats"]["n_factor"] == 2
    assert torch.is_tensor(sol["rho"]) and sol["rho"].shape == (16, 1, 1)
    # fp64 run of the same 101 iterations: rho after the refactorisation is sqrt(ratio of two residuals at the 1e-6
    # level), i.e. fp32 rounding noise amplified -- the reference's own fp32 result sits equally far from fp64
    inp = tuple(g[k] for k in ("Q", "p", "A", "b", "lb", "ub"))
    d = [t.double() for t in inp]
    c64 = O.make_control(rho=100.0, scale=(tag == "scale"), eps_abs=1e-12, eps_rel=1e-12, max_iters=101)
    s64 = O.solve_box_qp(*d, c64)
    g64 = O.solve_box_qp_grad(g["g"].double(), s64["x"], s64["u"], s64["lams"], s64["nus"], d[0], d[2], d[4], d[5], s64["rho"])
    case = f"g6_adaptive_{tag}"
    for k in ("x", "z", "lams", "nus"):
        close_or_fp64(case, k, sol[k], g[k], s64[k], X_TOL, linsolve=linsolve, mode=mode)
    close_or_fp64(case, "rho", sol["rho"], g["rho"], s64["rho"], 1e-4, linsolve=linsolve, mode=mode)
    gr = L.torch_solve_box_qp_grad(g["g"].to(dev), sol["x"], sol["u"], sol["lams"], sol["nus"], a[0], a[2], a[4], a[5], sol["rho"])
    for nm, t, t64 in zip(GRADS, gr, g64):
        close_or_fp64(case, nm, t, g[nm], t64, G_RTOL, linsolve=linsolve, mode=mode)


def test_g7_no_inequality_rho0_and_dict_mutation(dev):
    g = load_golden("g7_noineq")
    n, B = 20, 4
    lb = torch.full((B, n, 1), -float("inf"), device=dev)
    ub = torch.full((B, n, 1), float("inf"), device=dev)
    ctl = L.box_qp_control(**TOL)
    Q = g["Q"].to(dev).requires_grad_(True)
    x = L.SolveBoxQP(control=ctl)(Q, g["p"].to(dev), g["A"].to(dev), g["b"].to(dev), lb, ub)
    assert ctl["rho"] == 0                              # caller's dict mutated (reference :37-38)
    assert err(x, g["x_layer"]) < 2e-5
    sol = L.torch_solve_box_qp(g["Q"].to(dev), g["p"].to(dev), g["A"].to(dev), g["b"].to(dev), lb, ub, L.box_qp_control(**TOL))
    assert sol["iter"] == g["iter"] == 0 and sol["rho"] == 0 and not torch.is_tensor(sol["rho"])
    for k in ("x", "u", "lams", "nus"):
        assert err(sol[k], g[k]) < 2e-5, k
    x.sum().backward()
    assert torch.isfinite(Q.grad).all()


def test_g10_scalar_rho_return_types(dev):
    g = load_golden("g10_scalar_rho")
    inp = tuple(g[k] for k in ("Q", "p", "A", "b", "lb", "ub"))
    sa, _ = solve(dev, inp, O.make_control(rho=0.001, **TOL))
    assert torch.is_tensor(sa["rho"]) and sa["iter"] == g["a_iter"] and sa["_stats"]["rho_updated"] == 1
    sb, _ = solve(dev, inp, O.make_control(rho=1.0, adaptive_rho=False, **TOL))
    assert sb["rho"] == 1.0 and not torch.is_tensor(sb["rho"]) and sb["iter"] == g["b_iter"]
    for tag, s in (("a", sa), ("b", sb)):
        for k in ("x", "lams", "nus"):
            assert err(s[k], g[f"{tag}_{k}"]) < 5e-5, (tag, k)
    # a per-problem rho tensor is accepted too
    sc, _ = solve(dev, inp, O.make_control(rho=torch.full((4, 1, 1), 1.0), adaptive_rho=False, **TOL))
    assert torch.is_tensor(sc["rho"]) and err(sc["x"], g["b_x"]) < 5e-5


def test_g11_hard_distribution_fp64(dev):
    g = load_golden("g11_hard_f64")
    inp = O.create_hard_qp_data(100, 0.85, list(range(8)))
    sol, a = solve(dev, inp, O.make_control(**TOL))
    assert sol["iter"] == g["iter"]
    for k in ("x", "z", "u", "lams", "nus", "rho"):
        assert err(sol[k], g[k]) < 1e-7, k
    gr = L.torch_solve_box_qp_grad(g["g"].to(dev), sol["x"], sol["u"], sol["lams"], sol["nus"], a[0], a[2], a[4], a[5], sol["rho"])
    for nm, t in zip(GRADS, gr):
        assert rel(t, g[nm]) < 1e-5, nm


@pytest.mark.parametrize("sync", [False, True])
def test_g18_hard_distribution_fp64_at_the_benched_size(dev, sync):
    """SURVEY 8(c)'s G11 at its own size -- n = 250, m = round(sqrt(250)) = 16, prob 0.85, seeds 0..127, float64
    (experiments/utils.py:64-131, experiments/experiment_1_hard.py:13-35) -- made by the reference (tests/golden/make_golden.py, G18).
    This is bench.py's `b128_n250_m16_hard_fp64` row: the same module call (pipelined and with the layer's default synchronous calls),
    forward + fixed-point backward.  Iteration count equal, iterates at 1e-9, gradients at rtol 1e-6 of their scale."""
    g = load_golden("g18_hard_f64_n250_m16")
    inp = O.create_hard_qp_data(250, 0.85, list(range(128)))
    assert np.allclose(np.array([float(t.double().sum()) for t in inp]), g["in_sum"].numpy(), rtol=1e-10, atol=0), "generator drifted"
    a = [t.to(dev) for t in inp]
    leaves = [t.clone().requires_grad_(True) for t in a]
    ctl = dict(L.box_qp_control(**TOL))
    if not sync:
        ctl["sync"] = False
    x = L.SolveBoxQP(control=ctl)(*leaves)
    st = SB.last_forward_status(dev)
    x.backward(g["cot"].to(dev))
    L.synchronize()
    assert st["iters"] == int(g["iter"]) == 60, (st["iters"], int(g["iter"]))
    case = f"g18_hard_f64_n250_m16_sync{int(sync)}"
    P.record(case, "x", err(x, g["x"]), max(1.0, float(g["x"].abs().max())))
    assert err(x, g["x"]) < 1e-9
    sol = L.torch_solve_box_qp(*a, dict(L.box_qp_control(**TOL)))
    for k in ("x", "u", "nus", "rho"):
        sc = max(1.0, float(g[k].abs().max()))
        P.record(case, f"functional_{k}", err(sol[k], g[k]), sc)
        assert err(sol[k], g[k]) < 1e-9 * sc, k
    assert torch.equal(sol["x"], x.detach())
    grads = dict(zip(GRADS, [t.grad for t in leaves]))
    for nm in ("dp", "db", "dlb", "dub"):
        sc = float(g[nm].abs().max())
        P.record(case, nm, err(grads[nm], g[nm]), sc)
        assert err(grads[nm], g[nm]) <= 1e-6 * sc + 1e-12, nm
    sc = float(g["dA"].abs().max())
    assert err(grads["dA"][:8], g["dA"]) <= 1e-6 * sc, "dA"
    assert rel(torch.linalg.matrix_norm(grads["dA"]), g["dA_fro"]) < 1e-6 and rel(torch.linalg.matrix_norm(grads["dQ"]), g["dQ_fro"]) < 1e-6
    sb, si, sj = g["sb"].long(), g["si"].long(), g["sj"].long()
    sq = float(g["dQ_samples"].abs().max())
    P.record(case, "dQ_samples", err(grads["dQ"][sb.to(dev), si.to(dev), sj.to(dev)], g["dQ_samples"]), sq)
    assert err(grads["dQ"][sb.to(dev), si.to(dev), sj.to(dev)], g["dQ_samples"]) <= 1e-6 * sq


@pytest.mark.parametrize("case", ["config2_box_n100", "exp1_n330_m2", "hard_f64_n100", "rho_event_n50"])
def test_verbose_prints_the_reference_trace(dev, capsys, case):
    """verbose=True (reference :289-294): `iteration = i`, the largest primal and the largest dual error of the batch, at every
    check.  The loop runs on the device without the host, so the lines come after the solve -- from a trace the loop kernels
    keep when asked (ctrl.reserved2 bit 2, lqp_boxqp_check_trace): the same lines, the same iterations, the numbers within the
    solver's own float32 noise of the oracle's (which restates the prints).  Covers the small-matrix loop, the two-workgroup
    loop, the dense float64 tier and a solve with a refactorisation (the continuation kernel)."""
    if case == "config2_box_n100":
        Q, p, _, _, lb, ub = O.create_qp_data(100, 16, seed=0, with_eq=False)
        inp, kw, tol = (Q, p, None, None, lb, ub), dict(TOL), 2e-5
    elif case == "exp1_n330_m2":
        Q, p, _, _, lb, ub = O.create_qp_data(330, 5, seed=332, with_eq=False)
        g = torch.Generator().manual_seed(330)
        inp, kw, tol = (Q, p, torch.randn(5, 2, 330, generator=g), 0.1 * torch.randn(5, 2, 1, generator=g), lb, ub), dict(TOL), 2e-5
    elif case == "hard_f64_n100":
        inp, kw, tol = O.create_hard_qp_data(100, 0.85, list(range(8))), dict(TOL), 1e-6      # (the trace is kept in float32)
    else:
        g6 = load_golden("g6_adaptive_noscale")
        inp = tuple(g6[k] for k in ("Q", "p", "A", "b", "lb", "ub"))
        kw, tol = dict(rho=100.0, scale=False, **TOL), None

    def lines_of(text):
        out = []
        rows = [ln for ln in text.splitlines() if ln.startswith(("iteration = ", "|| primal_error|| = ", "|| dual_error|| = "))]
        for i in range(0, len(rows) - 2, 3):
            out.append((int(rows[i].split("=")[1]), float(rows[i + 1].split("=")[1]), float(rows[i + 2].split("=")[1])))
        return out

    ref = O.solve_box_qp(*inp, O.make_control(verbose=True, **kw))
    want = lines_of(capsys.readouterr().out)
    sol, _ = solve(dev, inp, O.make_control(verbose=True, **kw))
    got = lines_of(capsys.readouterr().out)
    assert sol["iter"] == ref["iter"] and len(want) == ref["iter"] // (want[1][0] - want[0][0]) + 1
    assert [w[0] for w in want] == [g[0] for g in got], (want, got)
    if tol is not None:
        for w, g in zip(want, got):
            assert abs(w[1] - g[1]) <= tol * max(1.0, w[1]) and abs(w[2] - g[2]) <= tol * max(1.0, w[2]), (w, g)
    else:
        # (rho = 100: from the first refactorisation on the errors follow a rho formed from rounding-level residuals)
        for w, g in zip(want[:5], got[:5]):
            assert abs(w[1] - g[1]) <= 1e-4 * max(1.0, w[1]) and abs(w[2] - g[2]) <= 1e-4 * max(1.0, w[2]), (w, g)


def test_g3_config2(dev):
    g = load_golden("g3_b128_n100_box")
    Q, p, _, _, lb, ub = O.create_qp_data(100, 128, seed=0, with_eq=False)
    sol, _ = solve(dev, (Q, p, None, None, lb, ub), O.make_control(**TOL))
    assert sol["iter"] == g["iter"] == 70
    for k in ("x", "u"):
        P.record("g3_b128_n100_box", k, err(sol[k], g[k]))
        assert err(sol[k], g[k]) < X_TOL, k


_G4_TRUTH = {}


def g4_truth():
    """fp64 solve (CPU oracle) of the headline batch with the reference's iteration count, and its gradients for the
    two cotangents of the golden file -- computed once per session (~10 s on the box's host cores)."""
    if not _G4_TRUTH:
        inp = O.create_qp_data(500, 128, seed=0)
        torch.manual_seed(7)
        cots = {"ones": torch.ones(128, 500, 1), "rand": torch.randn(128, 500, 1)}
        sol, grads = fp64_truth(inp, 60, list(cots.values()))
        _G4_TRUTH.update(inp=inp, cots=cots, sol=sol, grads=dict(zip(cots, grads)))
    return _G4_TRUTH


def check_g4_grads(case, g, gr, tag, t64, rtol=G_RTOL, **tags):
    """dp, dA, db, dlb, dub in full; dQ by its per-problem Frobenius norm and 64 sampled entries (128 MB otherwise)"""
    for idx, nm in enumerate(GRADS):
        if nm == "dQ" or gr[idx] is None:
            continue
        close_or_fp64(case, f"{nm}_{tag}", gr[idx], g[f"{nm}_{tag}"], t64[idx], rtol, **tags)
    dQ = gr[0]
    if dQ is not None:
        fro64 = torch.linalg.matrix_norm(t64[0])
        close_or_fp64(case, f"dQ_fro_{tag}", torch.linalg.matrix_norm(dQ), g[f"dQ_fro_{tag}"], fro64, G_RTOL, **tags)
        sb, si, sj = g["sb"].long(), g["si"].long(), g["sj"].long()
        close_or_fp64(case, f"dQ_samples_{tag}", dQ[sb.to(dQ.device), si.to(dQ.device), sj.to(dQ.device)],
                      g[f"dQ_samples_{tag}"], t64[0][sb, si, sj], G_RTOL, **tags)


@pytest.mark.parametrize("linsolve", ["lu", "spd"])
@pytest.mark.parametrize("mode", [1, 2])
def test_g4_headline_config3(dev, mode, linsolve):
    g = load_golden("g4_b128_n500_eq")
    T = g4_truth()
    sol, a = solve(dev, T["inp"], O.make_control(launch_mode=mode, linsolve=linsolve, **TOL))
    assert sol["iter"] == g["iter"] == 60 and sol["_stats"]["n_check"] == 4 and sol["_stats"]["n_factor"] == 1
    case = "g4_b128_n500_eq"
    for k in ("x", "z", "u", "lams", "nus"):             # (SURVEY 8c lists x, u, lams, nus for G4; z rides along)
        close_or_fp64(case, k, sol[k], g[k], T["sol"][k], X_TOL, linsolve=linsolve, mode=mode)
    for tag, cot in T["cots"].items():
        gr = L.torch_solve_box_qp_grad(cot.to(dev), sol["x"], sol["u"], sol["lams"], sol["nus"], a[0], a[2], a[4], a[5], sol["rho"])
        # (the functional backward = pivoted LU of the bordered system like the reference's linalg.solve (:393), plus
        #  one refinement step: without it dl_dz = ones -- in the row space of A, dv = 0 exactly, all cancellation --
        #  comes out with 1e-4 of fp32 LU noise, as it does from LAPACK's sgetrf/sgetrs on the same matrices:
        #  tests/tools/gpu_lu_accuracy.py)
        check_g4_grads(case, g, gr, tag, T["grads"][tag], linsolve=linsolve, mode=mode, backward="lu")


@pytest.mark.parametrize("sync", [False, True])
def test_the_benched_step_matches_g4(dev, sync):
    """Exactly what bench.py times: SolveBoxQP module call with bench.py's control (sync=False: the whole schedule
    enqueued without waiting, two-workgroup factorisation and loop, Cholesky backward), B=128 n=500 seed 0,
    x.backward(ones) and a random cotangent -- against the reference-made G4 vectors."""
    g = load_golden("g4_b128_n500_eq")
    T = g4_truth()
    Q, p, A, b, lb, ub = (t.to(dev) for t in T["inp"])
    control = L.box_qp_control(eps_rel=1e-5, eps_abs=1e-5, verbose=False, reduce='max')
    control['sync'] = sync
    layer = L.SolveBoxQP(control=control)
    case = "g4_benched_step"
    for tag, cot in T["cots"].items():
        Qg, pg = Q.clone().requires_grad_(True), p.clone().requires_grad_(True)
        Ag, bg = A.clone().requires_grad_(True), b.clone().requires_grad_(True)
        lbg, ubg = lb.clone().requires_grad_(True), ub.clone().requires_grad_(True)
        x = layer(Qg, pg, Ag, bg, lbg, ubg)
        st = SB.last_forward_status(dev)
        assert st["iters"] == 60 and st["linsolve_used"] == 2 and st["mode_used"] == (2 if sync else 3)
        assert st["loop_workgroups_per_qp"] == 2 and st["factor_launches"] > 1      # the small-batch schedules
        x.backward(cot.to(dev))
        L.synchronize()
        close_or_fp64(case, "x", x, g["x"], T["sol"]["x"], X_TOL, sync=sync)
        gr = (Qg.grad, pg.grad, Ag.grad, bg.grad, lbg.grad, ubg.grad)
        check_g4_grads(case, g, gr, tag, T["grads"][tag], sync=sync, backward="cholesky")


def test_g5_config4_n1000(dev):
    """BASELINE configs[3].  x against the reference-made golden at north_star's 1e-5, or -- fp32 rounding at n = 1000 --
    no further from an fp64 solve of the same inputs than the reference's own fp32 result: the fp64 truth is computed for
    16 of the 128 problems (the iteration count is pinned, so the problems are independent)."""
    g = load_golden("g5_b128_n1000_eq")
    inp = O.create_qp_data(1000, 128, seed=0)
    sol, _ = solve(dev, inp, O.make_control(**TOL))
    assert sol["iter"] == g["iter"] == 60 and sol["_stats"]["linsolve_used"] == 2      # symmetric path up to n = 1024
    assert rel(sol["rho"], g["rho"]) < 1e-4
    idx = torch.arange(0, 128, 8)
    t64, _ = fp64_truth([None if t is None else t[idx] for t in inp], g["iter"])
    case = "g5_b128_n1000_eq"
    P.record(case, "x_all_128", err(sol["x"], g["x"]), linsolve=sol["_stats"]["linsolve_used"])
    close_or_fp64(case, "x", sol["x"][idx.to(dev)], g["x"][idx], t64["x"], X_TOL)
    # the other 112 problems: within the band the 16 establish (their own fp32-vs-fp64 distance is not computed)
    band = max(X_TOL, err(g["x"][idx], t64["x"]) + X_TOL)
    assert err(sol["x"], g["x"]) <= band + err(sol["x"][idx.to(dev)], t64["x"]), (err(sol["x"], g["x"]), band)


# ---------------------------------------------------------------- symmetric-inverse x-update (linsolve 'spd')
@pytest.mark.parametrize("n,B", [(1, 2), (10, 3), (64, 2), (65, 2), (300, 3), (512, 2), (513, 2), (700, 2), (1000, 3), (1024, 1)])
def test_spd_inverse_entry(dev, n, B):
    """n <= 512: the panel of a pivot step lives in LDS; 512 < n <= 1024: wg_spd_sweep_big parks it in global scratch."""
    lib = _lib.load()
    torch.manual_seed(n)
    Lm = torch.randn(B, 2 * n + 2, n)
    K = (Lm.transpose(1, 2) @ Lm / (2 * n + 2) + 0.7 * torch.eye(n)).to(dev)
    out = torch.empty_like(K)
    info = torch.full((B,), -1, dtype=torch.int32, device=dev)
    nb = lib.lqp_spd_inverse_workspace_bytes(0, B, n)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    st = lib.lqp_spd_inverse_batched(_lib.stream_ptr(dev), 0, B, n, _lib.ptr(K), _lib.ptr(out), _lib.ptr(info), _lib.ptr(ws), nb)
    assert st == 0 and info.tolist() == [0] * B
    ref = torch.linalg.inv(K.double().cpu())
    assert err(out, ref) < 5e-6 * float(ref.abs().max())
    # not positive definite -> info says where, no exception, no fault
    K2 = K.clone()
    K2[0, n // 2, n // 2] = -1.0
    st = lib.lqp_spd_inverse_batched(_lib.stream_ptr(dev), 0, B, n, _lib.ptr(K2), _lib.ptr(out), _lib.ptr(info), _lib.ptr(ws), nb)
    assert st == 0 and int(info[0]) > 0 and info[1:].tolist() == [0] * (B - 1)
    assert lib.lqp_spd_inverse_batched(_lib.stream_ptr(dev), 1, B, n, _lib.ptr(K), _lib.ptr(out), _lib.ptr(info), _lib.ptr(ws), nb) == 6
    assert lib.lqp_spd_inverse_workspace_bytes(0, 1, 1025) > 0 and \
        lib.lqp_spd_inverse_batched(_lib.stream_ptr(dev), 0, 1, 1025, _lib.ptr(K), _lib.ptr(out), _lib.ptr(info), _lib.ptr(ws), 1 << 40) == 6


@pytest.mark.parametrize("n,m,B", [(10, 0, 5), (64, 1, 3), (100, 3, 4), (448, 16, 2), (512, 2, 2)])
def test_spd_and_lu_paths_agree(dev, n, m, B):
    """Both x-updates solve the same KKT system: same iteration count, same iterates to rounding."""
    torch.manual_seed(n + m)
    Q, p, _, _, lb, ub = O.create_qp_data(n, B, seed=n, with_eq=False)
    A = torch.randn(B, m, n) if m else None
    b = 0.1 * torch.randn(B, m, 1) if m else None
    sols = {}
    for ls in ("lu", "spd"):
        sols[ls], _ = solve(dev, (Q, p, A, b, lb, ub), O.make_control(linsolve=ls, **TOL))
    assert sols["lu"]["iter"] == sols["spd"]["iter"]
    for k in ("x", "z", "u", "lams") + (("nus",) if m else ()):
        assert err(sols["lu"][k], sols["spd"][k]) < 5e-5, k
    res = O.kkt_residuals(Q, p, A, b, lb, ub, {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in sols["spd"].items()})
    assert float(res["stationarity"].max()) < 2e-3 and float(res["box"].max()) < 1e-6
    if m:
        assert float(res["equality"].max()) < 2e-4


def test_spd_path_falls_back_to_lu(dev):
    """A Q that is not symmetric, or Q + rho I that is not positive definite, is outside the symmetric x-update.
    Default (synchronous) calls -- functional AND module -- repeat on the LU path by themselves (same answer as
    linsolve='lu', like the reference, which accepts any nonsingular KKT matrix); an un-synchronised module call
    (control['sync']=False) returns NaN, never plausible numbers, and reports the error late."""
    Q, p, A, b, lb, ub = O.create_qp_data(40, 4, seed=5)
    Qn = Q.clone()
    Qn[:, 3, 17] += 0.05                                   # not symmetric
    Qi = Q - 0.9 * torch.eye(40)                           # indefinite
    for Qx in (Qn, Qi):
        ctl = dict(max_iters=200, **TOL)
        s_auto, _ = solve(dev, (Qx, p, A, b, lb, ub), O.make_control(**ctl))
        s_lu, _ = solve(dev, (Qx, p, A, b, lb, ub), O.make_control(linsolve="lu", **ctl))
        assert s_auto["iter"] == s_lu["iter"] and s_auto["_stats"]["linsolve_used"] == 1
        assert torch.equal(torch.nan_to_num(s_auto["x"]), torch.nan_to_num(s_lu["x"]))   # (indefinite: may diverge)
    args = [t.to(dev) for t in (Qn, p, A, b, lb, ub)]
    # the module's default path: non-symmetric Q solved (LU fallback), gradients through the LU backward
    Qg = args[0].clone().requires_grad_(True)
    x_def = L.SolveBoxQP(control=L.box_qp_control(**TOL))(Qg, *args[1:])
    x_lu = L.SolveBoxQP(control=L.box_qp_control(linsolve="lu", **TOL))(*args)
    assert torch.isfinite(x_def).all() and torch.equal(x_def.detach(), x_lu)
    ref = O.solve_box_qp(Qn, p, A, b, lb, ub, O.make_control(**TOL))
    assert err(x_def, ref["x"]) < 5e-5
    x_def.sum().backward()
    assert torch.isfinite(Qg.grad).all()
    # pipelined mode: NaN + late error, then the queue is clean again
    x_async = L.SolveBoxQP(control=L.box_qp_control(sync=False, **TOL))(*args)
    with pytest.raises(RuntimeError, match="linsolve"):
        L.synchronize()
    assert torch.isnan(x_async).all()
    L.synchronize()
    x = L.SolveBoxQP(control=L.box_qp_control(linsolve="lu", sync=False, **TOL))(*args)
    L.synchronize()
    assert torch.isfinite(x).all() and torch.equal(x, x_lu)


@pytest.mark.parametrize("case", ["g2", "g4", "g6"])
def test_cholesky_backward_matches_goldens_and_lu(dev, case):
    """linsolve 2 of lqp_boxqp_backward_fp (blocked Cholesky of Q_FF + Schur complement of the equality rows)
    against the reference's gradients and against the LU form of the same system."""
    if case == "g2":
        g = load_golden("g2_b8_n50_eq")
        inp = tuple(g[k] for k in ("Q", "p", "A", "b", "lb", "ub")); cot = g["g_rand"]; ctl = O.make_control(**TOL)
        ref = {nm: g[f"{nm}_rand"] for nm in GRADS}
    elif case == "g4":
        g = load_golden("g4_b128_n500_eq")
        inp = O.create_qp_data(500, 128, seed=0); ctl = O.make_control(**TOL)
        torch.manual_seed(7); cot = torch.randn(128, 500, 1)
        ref = {nm: g[f"{nm}_rand"] for nm in GRADS[1:]}
    else:
        g = load_golden("g6_adaptive_scale")
        inp = tuple(g[k] for k in ("Q", "p", "A", "b", "lb", "ub")); cot = g["g"]; ctl = O.make_control(rho=100.0, scale=True, **TOL)
        ref = {nm: g[nm] for nm in GRADS}
    sol, a = solve(dev, inp, ctl)
    want = dict(dQ=True, dp=True, dA=True, db=True, dlb=True, dub=True)
    out = {}
    for ls in (1, 2):
        out[ls] = SB._fp_backward(cot.to(dev), sol["x"], sol["u"], sol["lams"], sol["nus"], a[0], a[2], a[4], a[5], sol["rho"],
                                 want, sync=True, linsolve=ls)
    prof = _lib.profile(enable=True, reset=True)
    SB._fp_backward(cot.to(dev), sol["x"], sol["u"], sol["lams"], sol["nus"], a[0], a[2], a[4], a[5], sol["rho"], want,
                   sync=True, linsolve=2)
    used = _lib.profile(); _lib.profile(enable=False)
    assert used["bwd_cholesky"][1] == 1 and used["lu_factor"][1] == 0          # really the Cholesky path
    # fp64 gradients from the fp64 forward of the same inputs (same iteration count)
    d = [t.double() for t in inp]
    c64 = dict(ctl, eps_abs=1e-12, eps_rel=1e-12, max_iters=sol["iter"] + 1)
    s64 = O.solve_box_qp(*d, c64)
    g64 = O.solve_box_qp_grad(cot.double(), s64["x"], s64["u"], s64["lams"], s64["nus"], d[0], d[2], d[4], d[5], s64["rho"])
    for idx, nm in enumerate(GRADS):
        scale = max(1.0, float(out[1][idx].abs().max()))
        P.record(f"chol_vs_lu_{case}", nm, err(out[2][idx], out[1][idx]), scale)
        assert err(out[2][idx], out[1][idx]) < 2e-4 * scale, nm
        if nm in ref:
            close_or_fp64(f"chol_backward_{case}", nm, out[2][idx], ref[nm], g64[idx], G_RTOL)


def test_cholesky_backward_falls_back(dev):
    """Q_FF not positive definite: the synchronous call repeats on the LU form and returns its answer."""
    Q, p, A, b, lb, ub = O.create_qp_data(40, 3, seed=9)
    Qi = Q - 0.5 * torch.eye(40)
    a = [t.to(dev) for t in (Qi, p, A, b, lb, ub)]
    x = torch.zeros(3, 40, 1, device=dev); u = torch.zeros_like(x)
    lams = torch.zeros(3, 80, 1, device=dev); nus = torch.zeros(3, 1, 1, device=dev)
    cot = torch.randn(3, 40, 1, device=dev)
    want = dict(dQ=True, dp=True, dA=True, db=True, dlb=True, dub=True)
    g1 = SB._fp_backward(cot, x, u, lams, nus, a[0], a[2], a[4], a[5], 1.0, want, sync=True, linsolve=1)
    g2 = SB._fp_backward(cot, x, u, lams, nus, a[0], a[2], a[4], a[5], 1.0, want, sync=True, linsolve=2)
    for t1, t2 in zip(g1[:6], g2[:6]):
        assert torch.equal(t1, t2)


@pytest.mark.parametrize("n,B,m", [(500, 16, 1), (200, 5, 3), (60, 4, 0)])
def test_backward_factorisation_ahead_of_the_cotangent(dev, monkeypatch, n, B, m):
    """lqp_boxqp_backward_fp_prefactor: a synchronous layer call enqueues the free set, Q_FF and its Cholesky factorisation
    right behind its forward; `backward` then only gathers the cotangent, solves and writes the gradients.  The same kernels
    on the same values as the one-call backward: identical bits.  Also: a workspace somebody else has used in between is not
    trusted (the backward runs in full), and a Q_FF that is not positive definite still ends in the pivoted-LU answer."""
    Q, p, _, _, lb, ub = O.create_qp_data(n, B, seed=n + m, with_eq=False)
    g = torch.Generator().manual_seed(n)
    A = torch.randn(B, m, n, generator=g) if m else None
    b = 0.1 * torch.randn(B, m, 1, generator=g) if m else None
    cot = torch.randn(B, n, 1, generator=g).to(dev)

    def run(prefactor, disturb=False):
        monkeypatch.setattr(SB, "_PREFACTOR_BWD", prefactor)
        leaves = [None if t is None else t.clone().to(dev).requires_grad_(True) for t in (Q, p, A, b, lb, ub)]
        prof = _lib.profile(enable=True, reset=True)
        x = L.SolveBoxQP(control=L.box_qp_control(**TOL))(*leaves)
        if disturb:      # another backward on the same stream takes the workspace before ours runs
            other = [None if t is None else t.clone().to(dev).requires_grad_(True) for t in (Q, p, A, b, lb, ub)]
            L.SolveBoxQP(control=L.box_qp_control(**TOL))(*other).sum().backward()
        x.backward(cot)
        used = _lib.profile(); _lib.profile(enable=False)
        return x.detach(), [None if t is None else t.grad for t in leaves], used

    x0, g0, u0 = run(False)
    x1, g1, u1 = run(True)
    x2, g2, u2 = run(True, disturb=True)
    assert u0["bwd_cholesky"][1] == 1 and u0["bwd_build"][1] == 1
    assert u1["bwd_cholesky"][1] == 2 and u1["bwd_build"][1] == 1 and u1["lu_factor"][1] == 0      # factor | solve, one build
    assert u2["bwd_build"][1] == 3 and u2["bwd_cholesky"][1] == 4     # ours ahead | the other call's ahead, then its solve | ours again in full
    assert torch.equal(x0, x1) and torch.equal(x0, x2)
    for a0, a1, a2 in zip(g0, g1, g2):
        if a0 is not None:
            assert torch.equal(a0, a1) and torch.equal(a0, a2)


def test_prefactored_backwards_in_forward_order(dev, monkeypatch):
    """forward A, forward B, backward A, backward B on DIFFERENT data of one shape and one stream: A's backward
    finds the shared workspace taken by B's prefactor and runs in full -- which overwrites B's free set and factor.  B's
    backward must notice (the count of writes moved) and run in full too, never solve with A's factor."""
    n, B = 96, 8
    cot = torch.randn(B, n, 1, generator=torch.Generator().manual_seed(5)).to(dev)
    data = [O.create_qp_data(n, B, seed=s) for s in (11, 12)]

    def leaves_of(d):
        return [t.clone().to(dev).requires_grad_(True) for t in d]

    def alone(d):
        monkeypatch.setattr(SB, "_PREFACTOR_BWD", False)
        lv = leaves_of(d)
        L.SolveBoxQP(control=L.box_qp_control(**TOL))(*lv).backward(cot)
        return [t.grad for t in lv]

    ref = [alone(d) for d in data]
    monkeypatch.setattr(SB, "_PREFACTOR_BWD", True)
    la, lb_ = leaves_of(data[0]), leaves_of(data[1])
    layer = L.SolveBoxQP(control=L.box_qp_control(**TOL))
    xa = layer(*la)
    xb = layer(*lb_)
    xa.backward(cot)
    xb.backward(cot)
    torch.cuda.synchronize()
    for got, want in ((la, ref[0]), (lb_, ref[1])):
        for t, w in zip(got, want):
            assert torch.equal(t.grad, w)


def test_prepared_backward_that_never_runs_and_late_gradient(dev, monkeypatch):
    """A forward with grad enabled prepares its backward (outputs, pinned report buffer, factorisation).  Dropped without a
    backward, the report buffer goes back to the pool (one leaked pinned buffer per such call); and above
    _PREPARE_DQ_MAX_BYTES the large gradient is allocated by `backward` itself -- same numbers either way."""
    import gc
    n, B = 96, 8
    d = O.create_qp_data(n, B, seed=21)
    cot = torch.randn(B, n, 1, generator=torch.Generator().manual_seed(6)).to(dev)
    layer = L.SolveBoxQP(control=L.box_qp_control(**TOL))

    def grads(limit):
        monkeypatch.setattr(SB, "_PREPARE_DQ_MAX_BYTES", limit)
        lv = [t.clone().to(dev).requires_grad_(True) for t in d]
        layer(*lv).backward(cot)
        torch.cuda.synchronize()
        return [t.grad for t in lv]

    early, late = grads(1 << 40), grads(0)
    for a, b_ in zip(early, late):
        assert torch.equal(a, b_)
    monkeypatch.setattr(SB, "_PREPARE_DQ_MAX_BYTES", 1 << 40)
    pool = lambda: sum(len(v) for k, v in _lib._pinned_free.items() if k == B)
    lv = [t.clone().to(dev).requires_grad_(True) for t in d]
    x = layer(*lv)
    torch.cuda.synchronize()
    before = pool()
    del x, lv
    gc.collect()
    assert pool() == before + 1, (before, pool())


def test_backward_factorisation_ahead_of_the_cotangent_falls_back(dev):
    """... and on the C ABI: prefactor + solve-only call == one call; a factorisation that failed ends in the LU retry."""
    lib = _lib.load()
    Q, p, A, b, lb, ub = O.create_qp_data(40, 3, seed=9)
    want = dict(dQ=True, dp=True, dA=True, db=True, dlb=True, dub=True)
    for shift in (0.0, -0.5):          # (-0.5: Q_FF not positive definite)
        a = [t.to(dev) for t in (Q + shift * torch.eye(40), p, A, b, lb, ub)]
        x = torch.zeros(3, 40, 1, device=dev); u = torch.zeros_like(x)
        lams = torch.zeros(3, 80, 1, device=dev); nus = torch.zeros(3, 1, 1, device=dev)
        cot = torch.randn(3, 40, 1, device=dev)
        g1 = SB._fp_backward(cot, x, u, lams, nus, a[0], a[2], a[4], a[5], 1.0, want, sync=True, linsolve=2)
        prep = SB._fp_backward_prepare(x, u, lams, nus, a[0], a[2], a[4], a[5], 1.0, want, sync=True, linsolve=2, prefactor=True)
        assert prep["pref"] is not None
        g2 = SB._fp_backward_run(prep, cot)
        for t1, t2 in zip(g1[:6], g2[:6]):
            assert torch.equal(t1, t2)
    # float64 / linsolve 1, the LU form (ABI 11): free set, reduced system, pivoted LU and packed factor ahead of the cotangent --
    # the same kernels on the same values as the one-call backward
    for dtype, ls in ((torch.float64, 1), (torch.float32, 1), (torch.float64, 2)):
        a = [t.to(dtype).to(dev) for t in (Q, p, A, b, lb, ub)]
        x = (0.3 * torch.randn(3, 40, 1, generator=torch.Generator().manual_seed(3))).to(dtype).to(dev)
        u = torch.zeros_like(x)
        lams = torch.zeros(3, 80, 1, dtype=dtype, device=dev); nus = torch.zeros(3, 1, 1, dtype=dtype, device=dev)
        cot = torch.randn(3, 40, 1, dtype=dtype, device=dev)
        g1 = SB._fp_backward(cot, x, u, lams, nus, a[0], a[2], a[4], a[5], 1.0, want, sync=True, linsolve=ls)
        prep = SB._fp_backward_prepare(x, u, lams, nus, a[0], a[2], a[4], a[5], 1.0, want, sync=True, linsolve=ls, prefactor=True)
        assert prep["pref"] is not None and prep["pref_reported"]
        _lib.profile(enable=True, reset=True)
        g2 = SB._fp_backward_run(prep, cot)
        torch.cuda.synchronize()
        used = _lib.profile(); _lib.profile(enable=False)
        assert used["lu_factor"][1] == 0 and used["pack"][1] == 0, (dtype, ls, used)      # (the solve phase only)
        for t1, t2 in zip(g1[:6], g2[:6]):
            assert torch.isfinite(t1).all() and torch.equal(t1, t2), (dtype, ls)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_shared_lu_that_times_out_degrades(dev, monkeypatch, dtype):
    """The two-workgroup LU with its partner workgroups missing (LQP_DBG_LU2_ABSENT: what a chip whose CUs are held by somebody
    else looks like): the hand-offs give up after their bounded spin, the info words say -7, and the synchronous layer repeats
    the solve -- forward and backward -- with one workgroup per matrix instead of failing: the same numbers as an
    undisturbed run."""
    n, B, m = 200, 3, 2
    Q, p, _, _, lb, ub = O.create_qp_data(n, B, seed=77, with_eq=False)
    g = torch.Generator().manual_seed(78)
    A = torch.randn(B, m, n, generator=g)
    b = A @ (0.5 * (lb + ub))
    cot = torch.randn(B, n, 1, generator=g)
    out = {}
    for absent in ("0", "1"):
        monkeypatch.setenv("LQP_DBG_LU2_ABSENT", absent)
        lv = [t.clone().to(dtype).to(dev).requires_grad_(True) for t in (Q, p, A, b, lb, ub)]
        _lib.profile(enable=True, reset=True)
        x = L.SolveBoxQP(control=L.box_qp_control(linsolve="lu", **TOL))(*lv)
        x.backward(cot.to(dtype).to(dev))
        torch.cuda.synchronize()
        used = _lib.profile(); _lib.profile(enable=False)
        out[absent] = ([x.detach()] + [t.grad for t in lv], used["lu_factor"][1])
    assert out["1"][1] > out["0"][1], (out["0"][1], out["1"][1])          # (the repeated factorisations)
    for a, e in zip(out["1"][0], out["0"][0]):
        assert torch.isfinite(a).all()
        assert err(a, e) <= 1e-5 * max(1.0, float(e.abs().max())), err(a, e)


@pytest.mark.parametrize("B,n", [(130, 320), (141, 450), (100, 500), (30, 500), (7, 400)])
def test_partner_workgroups_for_any_batch_size(dev, monkeypatch, B, n):
    """The kernels that share a problem between workgroups map workgroup ids to (problem, part) so that the partners meet on one
    XCD for EVERY batch size (shared_map: a grid padded to a multiple of 8 problems, surplus workgroups leave at once) -- also the
    turn-taking loop of batches above half the CUs, which used to need a multiple of 8.  Same iteration count and iterates as
    the schedules that share nothing (LQP_LOOP_SPLIT=0, LQP_SPD_SPLIT=0), gradients to float32 rounding."""
    d = O.create_qp_data(n, B, seed=B + n)
    cot = torch.randn(B, n, 1, generator=torch.Generator().manual_seed(B)).to(dev)
    out = {}
    for shared in ("1", "0"):
        monkeypatch.setenv("LQP_LOOP_SPLIT", shared)
        if shared == "0":
            monkeypatch.setenv("LQP_SPD_SPLIT", "0")
        lv = [t.clone().to(dev).requires_grad_(True) for t in d]
        sol = L.torch_solve_box_qp(*[t.detach() for t in lv], dict(L.box_qp_control(**TOL)))
        x = L.SolveBoxQP(control=L.box_qp_control(**TOL))(*lv)
        x.backward(cot)
        torch.cuda.synchronize()
        out[shared] = (sol["iter"], sol["_stats"]["loop_workgroups"], x.detach(), [t.grad for t in lv])
    assert out["1"][1] >= 2 and out["0"][1] == 1, (out["1"][1], out["0"][1])
    assert out["1"][0] == out["0"][0]
    assert err(out["1"][2], out["0"][2]) <= 2e-5
    for a, e in zip(out["1"][3], out["0"][3]):
        assert err(a, e) <= 1e-4 * max(1.0, float(e.abs().max()))


def test_shared_sweep_that_times_out_degrades(dev, monkeypatch):
    """... and the register-resident sweep of the factorisation with its partner workgroups missing (LQP_DBG_LOOP_ABSENT bit 1):
    the step flags time out, the synchronous solve is repeated with one workgroup per matrix."""
    n, B = 500, 3
    d = [t.to(dev) for t in O.create_qp_data(n, B, seed=89)]
    want = L.torch_solve_box_qp(*d, dict(L.box_qp_control(**TOL)))
    monkeypatch.setenv("LQP_DBG_LOOP_ABSENT", "2")
    got = L.torch_solve_box_qp(*d, dict(L.box_qp_control(**TOL)))
    assert got["iter"] == want["iter"] and err(got["x"], want["x"]) <= 2e-5
    assert got["_stats"]["loop_workgroups"] == 1 and want["_stats"]["loop_workgroups"] >= 2


def test_shared_unroll_sweep_that_times_out_leaves_nan(dev, monkeypatch):
    """unroll=True, the reverse sweep on two workgroups per QP with the partners missing (LQP_DBG_LOOP_ABSENT bit 2): nothing
    waits for that launch, so nothing can repeat it -- the gradients of the problems concerned are NaN, never plausible numbers."""
    n, B = 400, 2
    Q, p, A, b, lb, ub = O.create_qp_data(n, B, seed=90)
    cot = torch.randn(B, n, 1, generator=torch.Generator().manual_seed(91)).to(dev)
    grads = {}
    for absent in ("0", "4"):
        monkeypatch.setenv("LQP_DBG_LOOP_ABSENT", absent)
        lv = [t.clone().to(dev).requires_grad_(True) for t in (Q, p, A, b, lb, ub)]
        x = L.SolveBoxQP(control=L.box_qp_control(unroll=True, **TOL))(*lv)
        x.backward(cot)
        torch.cuda.synchronize()
        grads[absent] = [t.grad for t in lv]
    assert all(torch.isfinite(t).all() for t in grads["0"])
    assert not torch.isfinite(grads["4"][1]).any() and not torch.isfinite(grads["4"][0]).all()


@pytest.mark.parametrize("sync", [True, False])
def test_shared_loop_that_times_out_degrades(dev, monkeypatch, sync):
    """The two-workgroup loop of the symmetric path with its partner workgroups missing (LQP_DBG_LOOP_ABSENT): the exchange waits
    give up after their bounded spin and raise the time-out word; a synchronous call -- the split one of the layer and the one-call
    form of torch_solve_box_qp with a strict stop hook alike -- repeats the solve with nothing shared between workgroups and
    returns the undisturbed answer; a pipelined call cannot repeat: it reports the time-out late, its outputs are NaN."""
    n, B = 500, 4
    d = [t.to(dev) for t in O.create_qp_data(n, B, seed=88)]
    ctl = L.box_qp_control(sync=sync, **TOL)
    want = L.torch_solve_box_qp(*d, dict(L.box_qp_control(**TOL)))
    assert want["_stats"]["loop_workgroups"] >= 2
    monkeypatch.setenv("LQP_DBG_LOOP_ABSENT", "1")
    if sync:
        got = L.torch_solve_box_qp(*d, dict(ctl))
        assert got["iter"] == want["iter"] and got["_stats"]["loop_workgroups"] == 1
        assert err(got["x"], want["x"]) <= 2e-5
        x = L.SolveBoxQP(control=dict(ctl))(*d)
        assert err(x, want["x"]) <= 2e-5
    else:
        x = L.SolveBoxQP(control=dict(ctl))(*d)
        with pytest.raises(RuntimeError):
            L.synchronize()
        monkeypatch.setenv("LQP_DBG_LOOP_ABSENT", "0")
        assert not torch.isfinite(x).all()
        x = L.SolveBoxQP(control=dict(ctl))(*d)
        L.synchronize()
        assert err(x, want["x"]) <= 2e-5


@pytest.mark.parametrize("n,B,m", [(576, 3, 2), (700, 2, 0), (1000, 4, 1), (1024, 2, 0), (900, 130, 1)])
def test_streaming_loop_on_two_workgroups(dev, monkeypatch, n, B, m):
    """Above 512 rows (BASELINE configs[3]) the loop of the symmetric path streams H; with 2 B workgroups resident each problem gets
    TWO of them, one range of whole block columns each, the partial products exchanged every iteration (k_admm_loop_np2) -- against
    the one-workgroup loop (LQP_LOOP_NP2=0): the same iteration count, the iterates within float32 summation order (part 0 + part 1
    instead of one running sum), both against the oracle at the tolerance.  B = 130: 2 B workgroups do not fit, one workgroup each."""
    Q, p, _, _, lb, ub = O.create_qp_data(n, B, seed=n + B, with_eq=False)
    gen = torch.Generator().manual_seed(n)
    A = torch.randn(B, m, n, generator=gen) if m else None
    b = 0.1 * torch.randn(B, m, 1, generator=gen) if m else None
    inp = (Q, p, A, b, lb, ub)
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("LQP_LOOP_NP2", flag)
        out[flag], _ = solve(dev, inp, O.make_control(**TOL))
    two = 2 * B <= torch.cuda.get_device_properties(dev).multi_processor_count
    assert out["1"]["_stats"]["linsolve_used"] == 2 and out["1"]["_stats"]["loop_workgroups"] == (2 if two else 1)
    assert out["0"]["_stats"]["loop_workgroups"] == 1 and out["1"]["iter"] == out["0"]["iter"]
    ref = O.solve_box_qp(*inp, O.make_control(**TOL)) if B <= 8 else None
    for k in ("x", "z", "u", "lams") + (("nus",) if m else ()):
        assert err(out["1"][k], out["0"][k]) < 5e-6 * max(1.0, float(out["0"][k].abs().max())), k
        if ref is not None:
            assert err(out["1"][k], ref[k]) < 1e-4 * max(1.0, float(ref[k].abs().max())), k
    if ref is not None:
        assert out["1"]["iter"] == ref["iter"]


@pytest.mark.parametrize("n,B,m", [(577, 3, 1), (704, 2, 0), (1000, 4, 1), (1024, 2, 2), (960, 8, 5)])
def test_sweep_above_512_rows_two_steps_per_pass(dev, monkeypatch, n, B, m):
    """The sweep above 512 rows on two workgroups per matrix takes TWO pivot steps per pass over the tiles (wg_spd_sweep_big: step k's
    update on block column k + 1 alone, then both products into every other tile in one read-modify-write, subtracted one after the
    other): the operations of the two separate passes in their order -- with float32 products (LQP_SPD_BIG_F16=0) the same bits as
    LQP_SPD_BIG_FUSE=0, odd and even numbers of block rows.  The default carries the panel blocks as two-half operands and multiplies
    them on the float16 pipe (lqp_f16x2.hpp): the same iteration count, iterates within float32 rounding of the float32 build, both at
    the tolerance from the oracle."""
    Q, p, _, _, lb, ub = O.create_qp_data(n, B, seed=n + B, with_eq=False)
    gen = torch.Generator().manual_seed(n)
    A = torch.randn(B, m, n, generator=gen) if m else None
    b = 0.1 * torch.randn(B, m, 1, generator=gen) if m else None
    inp = (Q, p, A, b, lb, ub)
    out = {}
    for fuse, f16 in (("1", "0"), ("0", "0"), ("1", "1")):
        monkeypatch.setenv("LQP_SPD_BIG_FUSE", fuse)
        monkeypatch.setenv("LQP_SPD_BIG_F16", f16)
        out[fuse + f16], _ = solve(dev, inp, O.make_control(**TOL))
    assert out["10"]["_stats"]["linsolve_used"] == 2 and out["10"]["iter"] == out["00"]["iter"] == out["11"]["iter"]
    ref = O.solve_box_qp(*inp, O.make_control(**TOL)) if B <= 4 else None
    for k in ("x", "z", "u", "lams") + (("nus",) if m else ()):
        assert torch.equal(out["10"][k], out["00"][k]), k
        assert err(out["11"][k], out["00"][k]) < 3e-6 * max(1.0, float(out["00"][k].abs().max())), k
        if ref is not None:
            assert err(out["11"][k], ref[k]) < 1e-4 * max(1.0, float(ref[k].abs().max())), k


def test_streaming_loop_partner_missing_degrades(dev, monkeypatch):
    """... and with its partner workgroups missing (LQP_DBG_LOOP_ABSENT bit 3): the exchange gives up after its bounded spin, raises the
    time-out word, the synchronous call repeats the solve with nothing shared and returns the undisturbed answer."""
    n, B = 800, 2
    d = [t.to(dev) for t in O.create_qp_data(n, B, seed=5)]
    want = L.torch_solve_box_qp(*d, dict(L.box_qp_control(**TOL)))
    assert want["_stats"]["loop_workgroups"] == 2
    monkeypatch.setenv("LQP_DBG_LOOP_ABSENT", "8")
    got = L.torch_solve_box_qp(*d, dict(L.box_qp_control(**TOL)))
    assert got["iter"] == want["iter"] and got["_stats"]["loop_workgroups"] == 1
    assert err(got["x"], want["x"]) <= 2e-5


def test_report_of_the_factorisation_made_ahead(dev, monkeypatch):
    """ABI 11: the prefactor call stores the factorisation's info words into the report buffer of the backward call, which then
    waits for those words only (LQP_BWD_REPORTED) -- it returns while its solves and the epilogue run.  The same gradients as
    without the early report; a prepared backward that is dropped while its prefactor kernels may still be queued keeps its
    report buffer out of the pool until every word has arrived."""
    import gc
    n, B = 300, 16
    d = O.create_qp_data(n, B, seed=31)
    cot = torch.randn(B, n, 1, generator=torch.Generator().manual_seed(7)).to(dev)
    layer = L.SolveBoxQP(control=L.box_qp_control(**TOL))
    seen = []
    real = SB._fp_backward_run

    def spy(prep, dl_dz):
        seen.append(bool(prep.get('pref_reported')))
        return real(prep, dl_dz)
    monkeypatch.setattr(SB, "_fp_backward_run", spy)
    out = {}
    for early in ("1", "0"):
        monkeypatch.setenv("LQP_BWD_EARLY", early)
        lv = [t.clone().to(dev).requires_grad_(True) for t in d]
        layer(*lv).backward(cot)
        torch.cuda.synchronize()
        out[early] = [t.grad for t in lv]
    assert seen == [True, True]
    for a, b_ in zip(out["1"], out["0"]):
        assert torch.isfinite(a).all() and torch.equal(a, b_)
    # dropped right behind the forward: the buffer may only come back once the prefactor's words are in
    mine = lambda: [r for r in _lib._pinned_quarantine if r.numel() == B]
    for _ in range(4):
        lv = [t.clone().to(dev).requires_grad_(True) for t in d]
        x = layer(*lv)
        del x, lv
        gc.collect()
    torch.cuda.synchronize()
    rep = _lib.host_report(B)                  # (an allocation sweeps the quarantine)
    assert not mine()
    _lib._pinned_free[B].append(rep)
    lv = [t.clone().to(dev).requires_grad_(True) for t in d]
    layer(*lv).backward(cot)
    torch.cuda.synchronize()
    for a, t in zip(out["1"], lv):
        assert torch.equal(a, t.grad)


# ---------------------------------------------------------------- NumPy twin, OptNet (equality only)
_G14_CTL = {"a": dict(scale=False, adaptive_rho=False), "b": dict(), "c": dict(rho=5.0, adaptive_rho_iter=20)}


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_g14_numpy_twin(dev, tag):
    """lqp_py_amd.solve_box_qp_admm.solve_box_qp (NumPy in, NumPy out, float64) vs the reference's NumPy solver."""
    import numpy as np
    from lqp_py_amd.solve_box_qp_admm import solve_box_qp, BoxQP
    g = load_golden("g14_numpy_twin")
    arr = lambda k: g[f"{tag}_{k}"].numpy() if torch.is_tensor(g[f"{tag}_{k}"]) else g[f"{tag}_{k}"]
    has_eq = g[f"{tag}_A"] is not None
    ctl = L.box_qp_control(**TOL)
    ctl.update(_G14_CTL[tag])
    args = (arr("Q"), arr("p"), arr("A") if has_eq else None, arr("b") if has_eq else None, arr("lb"), arr("ub"))
    sol = solve_box_qp(*args, dict(ctl))
    assert sol["iter"] == int(arr("iter")) and sol["x"].dtype == np.float64 and sol["x"].shape == arr("x").shape
    for k in ("x", "z", "u", "lam"):
        assert np.abs(sol[k] - arr(k)).max() < 1e-8, k
    if has_eq:
        assert np.abs(sol["nu"] - arr("nu")).max() < 1e-8
    else:
        assert sol["nu"] is None
    assert abs(sol["rho"] - float(arr("rho"))) < 1e-6 * float(arr("rho"))
    for k in ("primal_error", "dual_error"):
        assert abs(sol[k] - float(arr(k))) < 1e-9 + 1e-5 * float(arr(k)), k
    holder = BoxQP(*args, dict(ctl))
    assert np.array_equal(holder.solve(), sol["x"])


def test_g15_optnet_equality_only(dev):
    g = load_golden("g15_optnet_eq")
    leaves = [g[k].to(dev).requires_grad_(True) for k in ("Q", "p", "A", "b")]
    x = L.OptNet(control=L.optnet_control())(leaves[0], leaves[1], leaves[2], leaves[3], None, None)
    assert err(x, g["x"]) < 1e-4
    x.backward(g["cot"].to(dev))
    for t, nm in zip(leaves, ("dQ", "dp", "dA", "db")):
        assert err(t.grad, g[nm]) < 1e-3 * max(1.0, float(g[nm].abs().max())), nm
    with pytest.raises(NotImplementedError):
        L.OptNet(control=L.optnet_control())(leaves[0], leaves[1], leaves[2], leaves[3],
                                             torch.ones(4, 2, 20, device=dev), torch.ones(4, 2, 1, device=dev))

# ---------------------------------------------------------------- size-independent properties
def test_kkt_conditions_at_full_size(dev):
    """Known-answer check independent of the oracle, at the headline size."""
    inp = O.create_qp_data(500, 128, seed=3)
    sol, a = solve(dev, inp, O.make_control(eps_abs=1e-6, eps_rel=1e-6))
    cpu = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in sol.items()}
    res = O.kkt_residuals(*inp, cpu)
    assert float(res["stationarity"].max()) < 2e-3
    assert float(res["equality"].max()) < 1e-4
    assert float(res["box"].max()) < 1e-6      # z = D * clip(., lb / D): one ulp of slack, as in the reference
    assert float(res["x_minus_z"].max()) < 1e-4
    assert float((cpu["lams"] < 0).sum()) == 0
    # complementarity: a multiplier is non-zero only on an active bound
    n = 500
    lo_gap = (cpu["z"] - inp[4]).abs()
    hi_gap = (inp[5] - cpu["z"]).abs()
    assert float((cpu["lams"][:, :n] * lo_gap).abs().max()) < 1e-3
    assert float((cpu["lams"][:, n:] * hi_gap).abs().max()) < 1e-3


def test_batch_independence_and_determinism(dev):
    """Each QP is independent: solving a sub-batch gives the same answers (up to the shared
    stopping rule, which we neutralise by fixing the iteration count)."""
    inp = O.create_qp_data(200, 12, seed=9)
    ctl = O.make_control(max_iters=40, eps_abs=1e-12, eps_rel=1e-12)
    full, _ = solve(dev, inp, ctl)
    again, _ = solve(dev, inp, ctl)
    assert torch.equal(full["x"], again["x"])                     # bitwise reproducible
    sub, _ = solve(dev, tuple(t[3:7] for t in inp), ctl)
    assert torch.equal(sub["x"], full["x"][3:7])


def test_more_problems_than_resident_workgroups(dev):
    """B > 256 CUs: the persistent grid barrier is impossible, the library must fall back to one launch
    per check segment on its own and still reproduce the global stopping rule."""
    inp = O.create_qp_data(32, 300, seed=11)
    ref = O.solve_box_qp(*inp, O.make_control(**TOL))
    sol, _ = solve(dev, inp, O.make_control(**TOL))
    assert sol["_stats"]["mode_used"] == 1 and sol["iter"] == ref["iter"]
    assert err(sol["x"], ref["x"]) < 5e-5 and err(sol["lams"], ref["lams"]) < 5e-5


def test_autograd_module_matches_functional(dev):
    inp = O.create_qp_data(64, 5, seed=4)
    Q, p, A, b, lb, ub = (t.to(dev) for t in inp)
    Qg, pg = Q.clone().requires_grad_(True), p.clone().requires_grad_(True)
    ctl = L.box_qp_control(**TOL)
    x = L.SolveBoxQP(control=ctl)(Qg, pg, A, b, lb, ub)
    cot = torch.randn_like(x)
    x.backward(cot)
    sol = L.torch_solve_box_qp(Q, p, A, b, lb, ub, L.box_qp_control(**TOL))
    gr = L.torch_solve_box_qp_grad(cot, sol["x"], sol["u"], sol["lams"], sol["nus"], Q, A, lb, ub, sol["rho"])
    assert torch.equal(x.detach(), sol["x"])
    # (the module's backward runs the Cholesky form of the reduced system, the functional one the reference's LU)
    assert rel(Qg.grad, gr[0]) < 1e-4 and rel(pg.grad, gr[1]) < 1e-4
    xl = L.SolveBoxQP(control=L.box_qp_control(linsolve='lu', **TOL))(Qg, pg, A, b, lb, ub)
    Qg.grad = None; pg.grad = None
    xl.backward(cot)
    sl = L.torch_solve_box_qp(Q, p, A, b, lb, ub, L.box_qp_control(linsolve='lu', **TOL))
    gl = L.torch_solve_box_qp_grad(cot, sl["x"], sl["u"], sl["lams"], sl["nus"], Q, A, lb, ub, sl["rho"])
    assert torch.equal(xl.detach(), sl["x"]) and torch.equal(Qg.grad, gl[0]) and torch.equal(pg.grad, gl[1])


def test_module_path_does_not_sync_and_defers_errors(dev):
    """SolveBoxQP (autograd path) enqueues the whole schedule without waiting for the GPU; results are
    the same as the synchronous path, errors surface at a later call / lqp_py_amd.synchronize()."""
    inp = [t.to(dev) for t in O.create_qp_data(48, 6, seed=2)]
    x_async = L.SolveBoxQP(control=L.box_qp_control(sync=False, **TOL))(*inp)
    x_sync = L.SolveBoxQP(control=L.box_qp_control(**TOL))(*inp)            # default: waits, raises at the call
    L.synchronize()
    assert torch.equal(x_async, x_sync)
    # adaptive rho firing inside the speculative schedule (G6): still the reference's answer
    g = load_golden("g6_adaptive_scale")
    a = [g[k].to(dev) for k in ("Q", "p", "A", "b", "lb", "ub")]
    xa = L.SolveBoxQP(control=L.box_qp_control(rho=100.0, scale=True, sync=False, **TOL))(*a)
    L.synchronize()
    assert err(xa, g["x"]) < 5e-5
    # singular KKT.  Default: RuntimeError at the call, like torch.linalg.lu_factor in the reference (:215) ...
    Qz = torch.zeros(2, 6, 6, device=dev)
    pz = torch.ones(2, 6, 1, device=dev)
    lbz, ubz = -torch.ones(2, 6, 1, device=dev), torch.ones(2, 6, 1, device=dev)
    with pytest.raises(RuntimeError, match="singular"):
        L.SolveBoxQP(control={"rho": 0.0, "scale": False})(Qz, pz, None, None, lbz, ubz)
    # ... pipelined: no exception at the call, NaN results, RuntimeError when the status arrives
    xz = L.SolveBoxQP(control={"rho": 0.0, "scale": False, "sync": False})(Qz, pz, None, None, lbz, ubz)
    with pytest.raises(RuntimeError, match="singular"):
        L.synchronize()
    assert torch.isnan(xz).all()
    L.synchronize()          # queue drained


def test_singular_kkt_raises(dev):
    Q = torch.zeros(2, 6, 6, device=dev)
    p = torch.ones(2, 6, 1, device=dev)
    lb, ub = -torch.ones(2, 6, 1, device=dev), torch.ones(2, 6, 1, device=dev)
    with pytest.raises(RuntimeError, match="singular|zero pivot"):
        L.torch_solve_box_qp(Q, p, None, None, lb, ub, {"rho": 0.0, "scale": False})


@pytest.mark.parametrize("native", [True, False])
def test_g12_kkt_backward_mode(dev, monkeypatch, native):
    """backward='kkt' (SURVEY 8f rank 2): reference solves the (3n+m) system, we its exact (n+m) reduction -- native: one
    library call on the fixed-point backward's kernels (lqp_boxqp_backward_kkt: Cholesky form behind the symmetric
    forward, gradients in the epilogue); otherwise composed from lqp_kkt_solve and torch ops."""
    monkeypatch.setattr(SB, "_KKT_NATIVE", native)
    _lib.profile(enable=True, reset=True)
    g = load_golden("g12_kkt_backward")
    leaves = [g[k].to(dev).requires_grad_(True) for k in ("Q", "p", "A", "b", "lb", "ub")]
    x = L.SolveBoxQP(control=L.box_qp_control(backward='kkt', **TOL))(*leaves)
    x.backward(g["cot"].to(dev))
    assert err(x, g["x"]) < 2e-5
    for nm, t in zip(GRADS, leaves):
        assert err(t.grad, g[nm]) < 1e-4 * max(1.0, float(g[nm].abs().max())), nm
    lv = [g[k].to(dev).requires_grad_(True) for k in ("Q", "p", "lb", "ub")]
    x2 = L.SolveBoxQP(control=L.box_qp_control(backward='kkt', **TOL))(lv[0], lv[1], None, None, lv[2], lv[3])
    x2.backward(g["cot"].to(dev))
    for nm, t in zip(("dQ_box", "dp_box", "dlb_box", "dub_box"), lv):
        assert err(t.grad, g[nm]) < 1e-4 * max(1.0, float(g[nm].abs().max())), nm
    used = _lib.profile(); _lib.profile(enable=False)
    assert (used["bwd_cholesky"][1] == 2) == native and (used["lu_factor"][1] == 0) == native, used
    # the functional entry (pivoted LU of the reduced system, it cannot know that Q is symmetric) and a one-sided batch
    # (upper bounds only: the reference's bookkeeping of dl_dh, :576-584, stays with the composed path)
    sol, a = solve(dev, [g[k] for k in ("Q", "p", "A", "b", "lb", "ub")], O.make_control(**TOL))
    gk = L.torch_solve_box_qp_grad_kkt(g["cot"].to(dev), sol["x"], sol["lams"], sol["nus"], a[0], a[2], a[4], a[5])
    for nm, t in zip(GRADS, gk):
        assert err(t, g[nm]) < 1e-4 * max(1.0, float(g[nm].abs().max())), nm
    inf = torch.full_like(a[4], float("inf"))
    s1 = L.torch_solve_box_qp(a[0], a[1], a[2], a[3], -inf, a[5], O.make_control(**TOL))
    g1 = L.torch_solve_box_qp_grad_kkt(g["cot"].to(dev), s1["x"], s1["lams"], s1["nus"], a[0], a[2], -inf, a[5])
    assert g1[4] is None and g1[5] is not None and all(torch.isfinite(t).all() for t in g1 if t is not None)


@pytest.mark.parametrize("sides", ["ub", "lb", "none"])
def test_kkt_backward_one_sided_native(dev, monkeypatch, sides):
    """backward='kkt' with upper bounds only / lower bounds only / no bounds on the library's kernels (round 5; the reference keeps
    both halves of G whenever a bound is finite, an infinite slack takes its rows out: lam / inf = 0) against the composition of
    lqp_kkt_solve and torch ops that restates the reference (:435-584), including its bookkeeping quirk in the ub-only case."""
    g = load_golden("g12_kkt_backward")
    a = [g[k].to(dev) for k in ("Q", "p", "A", "b", "lb", "ub")]
    inf = torch.full_like(a[4], float("inf"))
    lb = a[4] if sides == "lb" else -inf
    ub = a[5] if sides == "ub" else inf
    sol = L.torch_solve_box_qp(a[0], a[1], a[2], a[3], lb, ub, O.make_control(**TOL))
    cot = g["cot"].to(dev)
    out = {}
    for native in (False, True):
        monkeypatch.setattr(SB, "_KKT_NATIVE", native)
        _lib.profile(enable=True, reset=True)
        out[native] = L.torch_solve_box_qp_grad_kkt(cot, sol["x"], sol["lams"], sol["nus"], a[0], a[2], lb, ub)
        used = _lib.profile(); _lib.profile(enable=False)
        assert (used["bwd_epilogue"][1] == 1) == native, used
    for i, (t0, t1) in enumerate(zip(out[False], out[True])):
        assert (t0 is None) == (t1 is None), i
        if t0 is not None:
            assert err(t1, t0) < 1e-4 * max(1.0, float(t0.abs().max())), i
    # ... and through the module (Cholesky form of the reduced system behind the symmetric forward)
    monkeypatch.setattr(SB, "_KKT_NATIVE", True)
    lv = [t.clone().requires_grad_(True) for t in (a[0], a[1], a[2], a[3])]
    lbg, ubg = lb.clone().requires_grad_(sides == "lb"), ub.clone().requires_grad_(sides == "ub")
    x = L.SolveBoxQP(control=L.box_qp_control(backward='kkt', **TOL))(lv[0], lv[1], lv[2], lv[3], lbg, ubg)
    x.backward(cot)
    for t, r in zip(lv, out[False][:4]):
        assert err(t.grad, r) < 2e-4 * max(1.0, float(r.abs().max()))


class _CpuLU(torch.nn.Module):
    """CPU stand-in for the taped solve of the unrolled loop (lqp_py/lu_layer.py:5-58 restated with torch.linalg): lets
    lqp_py_amd.unrolled._eager_unrolled run in float64 on the host as the truth of the same taped computation."""

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


def _unroll_truth64(g, ctl):
    from lqp_py_amd.unrolled import _eager_unrolled
    leaves = [g[k].double().requires_grad_(True) for k in ("Q", "p", "A", "b", "lb", "ub")]
    x = _eager_unrolled(*leaves, SB.resolve_control(ctl, leaves[1].shape[1]), True, True, solver_cls=_CpuLU)
    x.backward(g["cot"].double())
    return x.detach(), [t.grad for t in leaves]


@pytest.mark.parametrize("name,tol", [("g13_unroll", 1e-6), ("g17_unroll_n100", 1e-5)])
@pytest.mark.parametrize("native", ["1", "0"])
def test_unroll_mode_goldens(dev, monkeypatch, name, tol, native):
    """unroll=True against the reference-made goldens G13 (n = 20) and G17 (n = 100): the solution at
    1e-5, all six gradients at rtol 1e-4 -- or no further from a float64 run of the same taped computation than the
    reference's own float32 gradients.  native 1: the HIP reverse sweep (lqp_boxqp_unroll_backward); 0: the taped loop."""
    monkeypatch.setenv("LQP_UNROLL_NATIVE", native)
    g = load_golden(name)
    leaves = [g[k].to(dev).requires_grad_(True) for k in ("Q", "p", "A", "b", "lb", "ub")]
    ctl = L.box_qp_control(unroll=True, eps_abs=tol, eps_rel=tol)
    _lib.profile(enable=True, reset=True)
    x = L.SolveBoxQP(control=ctl)(*leaves)
    assert torch.is_tensor(x)
    x.backward(g["cot"].to(dev))
    used = _lib.profile(); _lib.profile(enable=False)
    assert (used["unroll_backward"][1] == 2) == (native == "1"), used["unroll_backward"]
    x64, g64 = _unroll_truth64(g, ctl)
    close_or_fp64(name, "x", x, g["x"], x64, X_TOL, native=native)
    for nm, t, t64 in zip(GRADS, leaves, g64):
        close_or_fp64(name, nm, t.grad, g[nm], t64, G_RTOL, native=native)


@pytest.mark.parametrize("events", ["1", "0"])
def test_g22_unroll_through_a_rho_event(dev, monkeypatch, events):
    """unroll=True through ONE adaptive-rho refactorisation, against the reference-made golden G22 (Q x 50, rho = 100 given, no
    auto-scaling, tol 1e-6: the reference adapts rho at iteration 100 and stops there; its autograd runs through the adaptation
    itself, lqp_py/solve_box_qp_admm_torch.py:237-256).  The factor is not constant along this tape: events 1 = walked epoch by epoch
    in the library (lqp_boxqp_unroll_tape_segment; the adaptation by autograd on its own small graph), 0 = the taped loop of torch
    ops; solution at 1e-5, gradients at rtol 1e-4 or the float64 criterion."""
    monkeypatch.setenv("LQP_UNROLL_EVENTS", events)
    g = load_golden("g22_unroll_rho_event")
    leaves = [g[k].to(dev).requires_grad_(True) for k in ("Q", "p", "A", "b", "lb", "ub")]
    ctl = L.box_qp_control(unroll=True, rho=100.0, scale=False, eps_abs=1e-6, eps_rel=1e-6)
    _lib.profile(enable=True, reset=True)
    x = L.SolveBoxQP(control=dict(ctl))(*leaves)
    x.backward(g["cot"].to(dev))
    used = _lib.profile(); _lib.profile(enable=False)
    assert (used["unroll_backward"][1] > 0) == (events == "1") and used["lu_factor"][1] >= 2, used      # (the tape refactorised)
    x64, g64 = _unroll_truth64(g, dict(ctl))
    close_or_fp64("g22_unroll_rho_event", "x", x, g["x"], x64, X_TOL, events=events)
    for nm, t, t64 in zip(GRADS, leaves, g64):
        close_or_fp64("g22_unroll_rho_event", nm, t.grad, g[nm], t64, G_RTOL, events=events)
    plain = L.torch_solve_box_qp(*[t.detach() for t in leaves], dict(ctl, unroll=False))
    assert plain["iter"] == g["iter"] == 100
    # (the adapted rho is 100 x sqrt of a ratio of residual norms that are ~1e-6 of scale at iteration 90: float32 noise of a percent)
    assert float(((plain["rho"].reshape(-1).cpu() - g["rho"].reshape(-1)) / g["rho"].reshape(-1)).abs().max()) < 2e-2


@pytest.mark.parametrize("case", ["f64_given_rho", "f64_three_factors", "f64_scaled_small_rho", "f32_given_rho", "f64_box_only", "f32_wide",
                                  "f64_five_factors_auto_rho", "f32_five_factors_auto_rho", "f64_small_rho_every_20"])
def test_unroll_tape_with_rho_events_against_the_taped_loop(dev, monkeypatch, case):
    """Tapes along which rho is adapted once or several times (different given rho, adaptive_rho_iter 20 / 100, auto-scaling on and
    off, with and without equality rows, float32 and float64): the epoch-by-epoch walk in the library (lqp_boxqp_unroll_tape_segment +
    the adaptation by autograd on its own graph) against the taped loop of torch ops it replaces (LQP_UNROLL_EVENTS=0: the
    reference's tape as restated, lqp_py/solve_box_qp_admm_torch.py:235-313 under unroll).  float64: 1e-9 of the gradient's own
    largest entry (measured: 1e-15 ... 1e-13, two to five factorisations, active bounds); float32: rtol 1e-4 or no further from the
    float64 tape on the host than the torch-op tape."""
    cfg = {"f64_given_rho": (30, 5, 2, torch.float64, dict(rho=100.0, scale=False, eps_abs=1e-8, eps_rel=1e-8), 50.0),
           "f64_three_factors": (40, 3, 1, torch.float64, dict(rho=1e3, scale=False, eps_abs=1e-8, eps_rel=1e-8), 1.0),
           "f64_five_factors_auto_rho": (50, 4, 5, torch.float64, dict(adaptive_rho_iter=10, eps_abs=1e-9, eps_rel=1e-9), 1.0),
           "f32_five_factors_auto_rho": (50, 4, 5, torch.float32, dict(adaptive_rho_iter=10, eps_abs=1e-6, eps_rel=1e-6), 1.0),
           "f64_small_rho_every_20": (40, 3, 1, torch.float64, dict(rho=1e-2, scale=False, adaptive_rho_iter=20, eps_abs=1e-9, eps_rel=1e-9), 1.0),
           "f64_scaled_small_rho": (40, 3, 3, torch.float64, dict(rho=1e-3, eps_abs=1e-8, eps_rel=1e-8), 1.0),
           "f32_given_rho": (20, 4, 1, torch.float32, dict(rho=100.0, scale=False, eps_abs=1e-6, eps_rel=1e-6), 50.0),
           "f64_box_only": (35, 6, 0, torch.float64, dict(rho=200.0, scale=False, adaptive_rho_iter=50, eps_abs=1e-8, eps_rel=1e-8), 20.0),
           "f32_wide": (150, 3, 2, torch.float32, dict(rho=100.0, scale=False, eps_abs=1e-6, eps_rel=1e-6), 50.0)}[case]
    n, B, m, dtype, kw, qmul = cfg
    Q, p, _, _, lb, ub = O.create_qp_data(n, B, seed=n + B, with_eq=False)
    gen = torch.Generator().manual_seed(n + m)
    Q = Q * qmul
    A = torch.randn(B, m, n, generator=gen) if m else None
    b = 0.1 * torch.randn(B, m, 1, generator=gen) if m else None
    cot = torch.randn(B, n, 1, generator=gen)
    names = ("Q", "p", "A", "b", "lb", "ub")
    data = dict(zip(names, (Q, p, A, b, lb, ub)))
    ctl = L.box_qp_control(unroll=True, **kw)
    plain = L.torch_solve_box_qp(*[None if data[k] is None else data[k].to(dtype).to(dev) for k in names], dict(ctl, unroll=False))
    assert plain["_stats"]["n_factor"] >= 2, plain["_stats"]            # (the case must adapt rho at least once)
    grads = {}
    for events in ("1", "0"):
        monkeypatch.setenv("LQP_UNROLL_EVENTS", events)
        leaves = [None if data[k] is None else data[k].to(dtype).to(dev).requires_grad_(True) for k in names]
        _lib.profile(enable=True, reset=True)
        x = L.SolveBoxQP(control=dict(ctl))(*leaves)
        x.backward(cot.to(dtype).to(dev))
        used = _lib.profile(); _lib.profile(enable=False)
        assert (used["unroll_backward"][1] > 0) == (events == "1"), (events, used["unroll_backward"])
        grads[events] = (x.detach(), [None if t is None else t.grad for t in leaves])
    tag = f"unroll_rho_events_{case}"
    if dtype == torch.float64:
        assert err(grads["1"][0], grads["0"][0]) < 1e-8
        for nm, g1, g0 in zip(GRADS, grads["1"][1], grads["0"][1]):
            if g1 is None:
                assert g0 is None
                continue
            e = err(g1, g0) / (float(g0.abs().max()) + 1e-300)
            P.record(tag, nm, e, n_factor=plain["_stats"]["n_factor"], scale_of_gradient=float(g0.abs().max()))
            assert e < 1e-9, (nm, e)
        return
    from lqp_py_amd.unrolled import _eager_unrolled
    l64 = [None if data[k] is None else data[k].double().requires_grad_(True) for k in names]
    x64 = _eager_unrolled(*l64, SB.resolve_control(dict(ctl), n), True, True, solver_cls=_CpuLU)
    x64.backward(cot.double())
    close_or_fp64(tag, "x", grads["1"][0], grads["0"][0], x64.detach(), X_TOL)
    for nm, g1, g0, t64 in zip(GRADS, grads["1"][1], grads["0"][1], l64):
        if g1 is None:
            assert g0 is None
            continue
        close_or_fp64(tag, nm, g1, g0, t64.grad, G_RTOL)


@pytest.mark.parametrize("native", ["1", "0"])
def test_g21_unroll_float64_lu_tape(dev, monkeypatch, native):
    """unroll=True in float64 with three equality rows against the reference-made golden G21 (autograd through the reference's own
    loop, every node TorchLULayer on the pivoted LU, lqp_py/lu_layer.py:25-58; tol 1e-8): native 1 = the reverse sweep on the packed
    LU factor (lqp_boxqp_unroll_backward_lu), 0 = the taped loop of torch ops.  float64: solution and all six gradients at 1e-7
    of scale (the solve stops at 1e-8)."""
    monkeypatch.setenv("LQP_UNROLL_NATIVE", native)
    g = load_golden("g21_unroll_f64_m3")
    leaves = [g[k].to(dev).requires_grad_(True) for k in ("Q", "p", "A", "b", "lb", "ub")]
    assert leaves[0].dtype == torch.float64
    ctl = L.box_qp_control(unroll=True, eps_abs=1e-8, eps_rel=1e-8)
    _lib.profile(enable=True, reset=True)
    x = L.SolveBoxQP(control=ctl)(*leaves)
    x.backward(g["cot"].to(dev))
    used = _lib.profile(); _lib.profile(enable=False)
    assert (used["unroll_backward"][1] == 3) == (native == "1"), used["unroll_backward"]
    assert err(x, g["x"]) < 1e-7 * max(1.0, float(g["x"].abs().max()))
    for nm, t in zip(GRADS, leaves):
        e = err(t.grad, g[nm]) / max(1.0, float(g[nm].abs().max()))
        P.record("g21_unroll_f64_m3", nm, e, native=native)
        assert e < 1e-7, (nm, e)


@pytest.mark.parametrize("case", ["default", "ties", "no_scale", "rho_given", "beta_given", "box_only", "ragged", "many_rows"])
def test_unroll_scaling_chain_native_vs_autograd(dev, monkeypatch, case):
    """unroll=True: the scaling (:160-203) behind the reverse sweep with its Q-sized nodes on the library's one-pass kernels
    (lqp_unroll_scale_*; the n-sized rest by autograd) against the whole chain as eager torch ops (LQP_UNROLL_SCALE_NATIVE=0):
    the same sweep output goes through both, so the gradients agree to float32 summation order.  `ties`: two rows attain a
    column's maximum of |Q| -- amax shares that column's gradient between them, signs included."""
    n, B = (61, 3) if case == "ragged" else (96, 4)
    Q, p, A, b, lb, ub = O.create_qp_data(n, B, seed=41)
    ctl = L.box_qp_control(unroll=True, **TOL)
    if case == "ties":
        Q = Q.clone()
        big = float(Q.abs().max()) * 1.5
        Q[:, 3, 7] = big; Q[:, 7, 3] = big; Q[:, 11, 7] = -big; Q[:, 7, 11] = -big      # (symmetric; rows 3 and 11 tie in column 7)
        Q[:, 7, 7] += 4 * big; Q[:, 3, 3] += 4 * big; Q[:, 11, 11] += 4 * big            # (still positive definite)
    if case == "no_scale":
        ctl["scale"] = False
    if case == "rho_given":
        ctl["rho"] = 0.7
    if case == "beta_given":
        ctl["beta"] = 0.3
    if case == "box_only":
        A = b = None
    if case == "many_rows":                                # five equality rows through a feasible point
        A = torch.randn(B, 5, n, generator=torch.Generator().manual_seed(43))
        b = A @ (0.5 * (lb + ub))
    cot = torch.randn(B, n, 1, generator=torch.Generator().manual_seed(42)).to(dev)
    got = {}
    for flag in ("1", "2", "0"):         # 1: five kernels, no autograd; 2: the Q-sized nodes native, the n-sized ones by autograd; 0: all eager
        monkeypatch.setenv("LQP_UNROLL_SCALE_NATIVE", flag)
        leaves = [None if t is None else t.clone().to(dev).requires_grad_(True) for t in (Q, p, A, b, lb, ub)]
        _lib.profile(enable=True, reset=True)
        x = L.SolveBoxQP(control=dict(ctl))(*leaves)
        x.backward(cot)
        used = _lib.profile(); _lib.profile(enable=False)
        assert used["unroll_backward"][1] == 2, used
        assert (used["unroll_scaling"][1] > 0) == (flag != "0"), used["unroll_scaling"]
        got[flag] = [None if t is None else t.grad for t in leaves]
    for flag in ("1", "2"):
        for nm, a, e in zip(GRADS, got[flag], got["0"]):
            if e is None:
                assert a is None, (flag, nm)
                continue
            scale = max(1e-3, float(e.abs().max()))
            assert err(a, e) <= 2e-5 * scale, (case, flag, nm, err(a, e), scale)


@pytest.mark.parametrize("n,B,m", [(300, 4, 1), (500, 6, 1), (400, 3, 3), (512, 2, 0)])
def test_unroll_sweep_on_two_workgroups(dev, monkeypatch, n, B, m):
    """unroll=True: the reverse sweep shared by two workgroups per QP (k_unroll_sweep_split: the split loop's products, partial
    vectors exchanged per product) against the one-workgroup sweep (LQP_UNROLL_SPLIT=0): the same recurrence, products summed in
    another order -- gradients to float32 rounding of their scale."""
    Q, p, A, b, lb, ub = O.create_qp_data(n, B, seed=50 + n)
    if m == 0:
        A = b = None
    elif m > 1:
        A = torch.randn(B, m, n, generator=torch.Generator().manual_seed(51))
        b = A @ (0.5 * (lb + ub))
    cot = torch.randn(B, n, 1, generator=torch.Generator().manual_seed(52)).to(dev)
    got = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("LQP_UNROLL_SPLIT", flag)
        leaves = [None if t is None else t.clone().to(dev).requires_grad_(True) for t in (Q, p, A, b, lb, ub)]
        x = L.SolveBoxQP(control=L.box_qp_control(unroll=True, **TOL))(*leaves)
        x.backward(cot)
        got[flag] = [None if t is None else t.grad for t in leaves]
    for nm, a, e in zip(GRADS, got["1"], got["0"]):
        if e is None:
            assert a is None, nm
            continue
        scale = max(1e-3, float(e.abs().max()))
        assert err(a, e) <= 5e-5 * scale, (n, m, nm, err(a, e), scale)


def test_unroll_native_falls_back(dev):
    """Which tape runs where: float64 and the cached-LU x-update take the reverse sweep on the packed LU factor (round 6:
    lqp_boxqp_unroll_backward_lu -- sweep, equality rows, outer product: three launches); a solve in which rho was adapted (the
    factor is no longer constant along the tape) is walked epoch by epoch (lqp_boxqp_unroll_tape_segment: a replay and a reverse
    launch per segment + the two of the finish).  Nothing takes the taped loop of torch ops by itself any more."""
    g = load_golden("g13_unroll")
    for kind in ("f64", "lu", "adapted"):
        cast = (lambda t: t.double()) if kind == "f64" else (lambda t: t)
        leaves = [cast(g[k]).to(dev).requires_grad_(True) for k in ("Q", "p", "A", "b", "lb", "ub")]
        ctl = L.box_qp_control(unroll=True, eps_abs=1e-6, eps_rel=1e-6)
        if kind == "lu":
            ctl["linsolve"] = "lu"
        if kind == "adapted":
            leaves[0] = (g["Q"] * 50).to(dev).requires_grad_(True)
            ctl.update(rho=100.0, scale=False)
        _lib.profile(enable=True, reset=True)
        x = L.SolveBoxQP(control=ctl)(*leaves)
        x.backward(cast(g["cot"]).to(dev))
        used = _lib.profile(); _lib.profile(enable=False)
        assert (used["unroll_backward"][1] >= 4 if kind == "adapted" else used["unroll_backward"][1] == 3) and used["lu_factor"][1] >= 1, (kind, used)
        assert all(torch.isfinite(t.grad).all() for t in leaves), kind
        if kind != "adapted":
            assert err(x, cast(g["x"])) < 2e-5, kind


def test_training_loop_matches_cpu_oracle(dev):
    """Linear -> SolveBoxQP -> QP loss -> SGD (experiments/experiment_2.py:57-99).
    The GPU layer and an autograd wrapper around the CPU oracle must produce the same loss trajectory."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    import experiment_2 as E

    class OracleLayer(torch.autograd.Function):
        @staticmethod
        def forward(ctx, Q, p, A, b, lb, ub):
            sol = O.solve_box_qp(Q, p, A, b, lb, ub, O.make_control(**TOL))
            ctx.save_for_backward(sol["x"], sol["u"], sol["lams"], sol["nus"], Q, A, lb, ub, sol["rho"])
            return sol["x"]

        @staticmethod
        def backward(ctx, g):
            x, u, lams, nus, Q, A, lb, ub, rho = ctx.saved_tensors
            return O.solve_box_qp_grad(g, x, u, lams, nus, Q, A, lb, ub, rho)[:6]

    kw = dict(n_x=40, n_batch=16, n_mini=8, n_epochs=6, lr=1e-2, verbose=False)
    gpu_losses, _, _ = E.train(dev=dev, **kw)
    cpu_losses, _, _ = E.train(dev=torch.device("cpu"), layer=lambda *a: OracleLayer.apply(*a), **kw)
    L.synchronize()
    assert len(gpu_losses) == 6
    for a, b in zip(gpu_losses, cpu_losses):
        assert abs(a - b) < 1e-3 * max(1.0, abs(b)), (gpu_losses, cpu_losses)


def test_unsupported_sizes_fail_loudly(dev):
    # (n + m <= 4096 in float32 since round 6 -- four panel rows per thread --, 2048 in float64: the LDS holds 2 * PB * N elements of a panel)
    for n, dt in ((4097, torch.float32), (2049, torch.float64)):
        with pytest.raises(RuntimeError, match="unsupported"):
            L.torch_solve_box_qp(torch.zeros(1, n, n, device=dev, dtype=dt), torch.zeros(1, n, 1, device=dev, dtype=dt), None, None,
                                 -torch.ones(1, n, 1, device=dev, dtype=dt), torch.ones(1, n, 1, device=dev, dtype=dt), {})


def test_g16_n1500_above_the_on_chip_tiers(dev):
    """Reference-made golden at n = 1500, m = 1, B = 8 (the reference's LAPACK calls take any size; README.md:49 discusses
    n_x > 500): forward iterates + all six fixed-point gradients.  n + m > 1024 runs the reference's own algorithm, the
    pivoted LU of the KKT matrix with cached triangular solves, on the HBM-resident tier."""
    g = load_golden("g16_b8_n1500_eq")
    inp = O.create_qp_data(1500, 8, seed=0)
    assert abs(float(inp[0].double().sum()) - float(g["in_sum"][0])) < 1e-6 * abs(float(g["in_sum"][0]))
    sol, a = solve(dev, inp, O.make_control(**TOL))
    assert sol["iter"] == g["iter"] and sol["_stats"]["linsolve_used"] == 1
    t64, g64 = fp64_truth(inp, int(g["iter"]), cots=(g["cot"],))
    case = "g16_b8_n1500_eq"
    for k in ("x", "z", "u", "lams", "nus"):
        close_or_fp64(case, k, sol[k], g[k], t64[k], X_TOL)
    # (rho = ||Qs||_F / sqrt(n), a float32 sum of 2.25 M squares: within 1e-5, or no further from float64 than the reference's)
    close_or_fp64(case, "rho", sol["rho"], g["rho"], t64["rho"], 1e-5)
    gr = L.torch_solve_box_qp_grad(g["cot"].to(dev), sol["x"], sol["u"], sol["lams"], sol["nus"], a[0], a[2], a[4], a[5], sol["rho"])
    for idx, nm in enumerate(GRADS):
        if nm == "dQ":
            continue
        close_or_fp64(case, nm, gr[idx], g[nm], g64[0][idx], G_RTOL)
    dQ = gr[0]
    close_or_fp64(case, "dQ_fro", torch.linalg.matrix_norm(dQ), g["dQ_fro"], torch.linalg.matrix_norm(g64[0][0]), G_RTOL)
    sb, si, sj = (g[k].long() for k in ("sb", "si", "sj"))
    close_or_fp64(case, "dQ_samples", dQ[sb.to(dev), si.to(dev), sj.to(dev)], g["dQ_samples"], g64[0][0][sb, si, sj], G_RTOL)
    # the module path (autograd) at the same size, float64 inputs: the same kernels in double
    d = [t.double().to(dev) for t in inp]
    Qg = d[0].clone().requires_grad_(True)
    x64 = L.SolveBoxQP(control=L.box_qp_control(**TOL))(Qg, *d[1:])
    x64.backward(g["cot"].double().to(dev))
    assert err(x64, t64["x"]) < 1e-4 and torch.isfinite(Qg.grad).all()      # (its own stopping point: the tolerance level)


def test_g20_n3000_above_2048_rows(dev):
    """Reference-made golden at n = 3000, m = 1, B = 2, float32 (the reference's LAPACK calls at solve_box_qp_admm_torch.py:205-215
    take any size; README.md:49 discusses large n_x): forward iterates, iteration count and the fixed-point gradients.  Above 2048
    rows the pivoted LU holds four panel rows per thread; loop and backward are the HBM-resident tier's."""
    g = load_golden("g20_b2_n3000_eq")
    inp = O.create_qp_data(3000, 2, seed=20)
    assert abs(float(inp[0].double().sum()) - float(g["in_sum"][0])) < 1e-6 * abs(float(g["in_sum"][0]))
    sol, a = solve(dev, inp, O.make_control(**TOL))
    assert sol["iter"] == g["iter"] and sol["_stats"]["linsolve_used"] == 1
    t64, g64 = fp64_truth(inp, int(g["iter"]), cots=(g["cot"],))
    case = "g20_b2_n3000_eq"
    for k in ("x", "z", "u", "lams", "nus"):
        close_or_fp64(case, k, sol[k], g[k], t64[k], X_TOL)
    close_or_fp64(case, "rho", sol["rho"], g["rho"], t64["rho"], 1e-5)
    gr = L.torch_solve_box_qp_grad(g["cot"].to(dev), sol["x"], sol["u"], sol["lams"], sol["nus"], a[0], a[2], a[4], a[5], sol["rho"])
    for idx, nm in enumerate(GRADS):
        if nm == "dQ":
            continue
        close_or_fp64(case, nm, gr[idx], g[nm], g64[0][idx], G_RTOL)
    dQ = gr[0]
    close_or_fp64(case, "dQ_fro", torch.linalg.matrix_norm(dQ), g["dQ_fro"], torch.linalg.matrix_norm(g64[0][0]), G_RTOL)
    sb, si, sj = (g[k].long() for k in ("sb", "si", "sj"))
    close_or_fp64(case, "dQ_samples", dQ[sb.to(dev), si.to(dev), sj.to(dev)], g["dQ_samples"], g64[0][0][sb, si, sj], G_RTOL)
    # the module path (autograd) at the same size
    leaves = [t.to(dev).clone().requires_grad_(True) for t in inp]
    xm = L.SolveBoxQP(control=L.box_qp_control(**TOL))(*leaves)
    xm.backward(g["cot"].to(dev))
    assert err(xm, sol["x"]) < 1e-6 and all(torch.isfinite(t.grad).all() for t in leaves)
    assert err(leaves[1].grad, gr[1]) < 1e-6 * max(1.0, float(gr[1].abs().max()))


# ---------------------------------------------------------------- bench.py contract
def test_bench_json_contract(dev):
    """bench.py prints ONE JSON line with the keys the driver reads, the roofline objects and (on request) the CPU
    baseline; values are sane for a short run."""
    import json
    import subprocess
    import sys as _sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([_sys.executable, os.path.join(repo, "bench.py"), "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline", "--no-other-configs"], capture_output=True, text=True, timeout=400)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["unit"] == "QPs/sec" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["value"] > 1000 and abs(d["value"] - 128 / (d["ms_per_step"] * 1e-3)) < 0.01 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["peak"] in (8000.0, 157.3) and r["frac"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert d["roofline_factorisation"]["bound"] == "mfma" and 0 < d["roofline_factorisation"]["frac"] < 1.0
    assert d["roofline_loop"]["bound"] == "hbm" and d["roofline_loop"]["iterations"] == 61
    assert d["config"]["iters"] == 60 and d["config"]["checks"] == 4 and d["config"]["launch_mode"] == 3
    e1 = d["experiment_1_protocol"]
    assert e1["simulations"] == 10 and e1["QPs_per_sec_median"] > 1000


@pytest.mark.parametrize("seed", range(6))
def test_random_small_problems_vs_cpu_oracle(dev, seed):
    """Randomised sizes / constraint counts / controls against the CPU oracle, on both x-updates.  The stopping
    check is a floating-point threshold: a borderline problem may stop one check later on one side, so iteration
    counts may differ by one check interval (then only the KKT conditions are compared)."""
    g = torch.Generator().manual_seed(1000 + seed)
    n = int(torch.randint(2, 90, (1,), generator=g))
    m = int(torch.randint(0, 4, (1,), generator=g))
    B = int(torch.randint(1, 7, (1,), generator=g))
    Q, p, _, _, lb, ub = O.create_qp_data(n, B, seed=seed, with_eq=False)
    A = torch.randn(B, m, n, generator=g) if m else None
    b = 0.1 * torch.randn(B, m, 1, generator=g) if m else None
    extra = [dict(), dict(scale=False), dict(adaptive_rho=False), dict(rho=1.5), dict(scale=False, rho=0.8),
             dict(check_solved=3)][seed % 6]
    ctl = O.make_control(**TOL)
    ctl.update(extra)
    ref = O.solve_box_qp(Q, p, A, b, lb, ub, dict(ctl))
    interval = O.resolve_control(ctl, n).check_solved
    for ls in ("lu", "spd"):
        c2 = dict(ctl)
        c2["linsolve"] = ls
        sol, _ = solve(dev, (Q, p, A, b, lb, ub), c2)
        assert abs(sol["iter"] - ref["iter"]) <= interval, (ls, sol["iter"], ref["iter"])
        if sol["iter"] == ref["iter"]:
            for k in ("x", "z", "u", "lams") + (("nus",) if m else ()):
                assert err(sol[k], ref[k]) < 5e-5, (ls, k)
        res = O.kkt_residuals(Q, p, A, b, lb, ub, {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in sol.items()})
        assert float(res["box"].max()) < 1e-6 and float(res["x_minus_z"].max()) < 1e-3


@pytest.mark.parametrize("n,B,m", [(200, 3, 1), (330, 5, 2), (512, 2, 0)])
@pytest.mark.parametrize("mode", [1, 2])
def test_split_factorisation_is_bit_identical_and_tracks_the_oracle(dev, monkeypatch, n, B, m, mode):
    """Small batches share each matrix between two workgroups (one launch per pivot step, lookahead pivot block).
    The arithmetic per tile is the same as in the single-launch sweep -> identical bits; rho = 100 forces the
    adaptive-rho refactorisation (gated launches in mode 1, in-kernel in mode 2); the CPU oracle is the checker."""
    Q, p, _, _, lb, ub = O.create_qp_data(n, B, seed=n + m, with_eq=False)
    g = torch.Generator().manual_seed(n)
    A = torch.randn(B, m, n, generator=g) if m else None
    b = 0.1 * torch.randn(B, m, 1, generator=g) if m else None
    sols = {}
    # (the resident sweep normally takes rho = ||Qs||_F / sqrt(n) from sums k_spd_begin leaves -- another summation
    #  order than the setup pass, i.e. another last bit of rho; test_late_rho_matches_the_setup_pass covers that)
    monkeypatch.setenv("LQP_RHO_LATE", "0")
    # (... and applies the equality correction inside the loop kernel, with the summation order of its own products:
    #  test_equality_correction_in_the_loop_kernel)
    monkeypatch.setenv("LQP_EQ_IN_LOOP", "0")
    # (... and the one-workgroup sweep behind k_spd_prep sums ||Qs||_F from the prepared blocks: the same last-bit matter)
    monkeypatch.setenv("LQP_PREP_ONE", "0")
    # (... and, since round 6, multiplies its panels on the float16 matrix pipe with two-half operands: float32-grade products
    #  in another rounding -- test_sweep_on_the_float16_pipe_matches_the_float32_sweep; LQP_SPD_F16=0 is the exact-float32 build
    #  of the same schedule, which is what is bit-identical to the multi-launch sweep)
    monkeypatch.setenv("LQP_SPD_F16", "0")
    for split in ("1", "0"):
        monkeypatch.setenv("LQP_SPD_SPLIT", split)
        for rho in (None, 100.0):
            ctl = O.make_control(rho=rho, linsolve="spd", launch_mode=mode, **TOL)
            sols[split, rho], _ = solve(dev, (Q, p, A, b, lb, ub), ctl)
            assert sols[split, rho]["_stats"]["linsolve_used"] == 2
    for rho in (None, 100.0):
        s1, s0 = sols["1", rho], sols["0", rho]
        assert s1["iter"] == s0["iter"] and s1["_stats"]["n_factor"] == s0["_stats"]["n_factor"]
        assert s1["_stats"]["n_launch"] > s0["_stats"]["n_launch"]        # really the multi-launch path
        for k in ("x", "z", "u", "lams") + (("nus",) if m else ()):
            assert torch.equal(s1[k], s0[k]), (rho, k)
        ref = O.solve_box_qp(Q, p, A, b, lb, ub, O.make_control(rho=rho, **TOL))
        assert abs(s1["iter"] - ref["iter"]) <= ref["iter"] // 4 + 20
        assert err(s1["x"], ref["x"]) < 5e-4 * max(1.0, float(ref["x"].abs().max()))
    assert sols["1", 100.0]["_stats"]["n_factor"] >= 2


@pytest.mark.parametrize("n,B,m", [(500, 4, 1), (333, 3, 2), (448, 2, 0), (449, 70, 1), (300, 5, 3), (512, 2, 16), (250, 5, 2), (150, 6, 1), (129, 3, 0)])
@pytest.mark.parametrize("rho", [None, 0.7])
def test_pass_over_q_inside_the_resident_sweep(dev, monkeypatch, n, B, m, rho):
    """FwdParams::prep_fused == 3: no k_spd_prep launch -- the workgroups that keep the matrix in their registers read Q
    themselves (tiles, mirrors for the column maxima and the symmetry verdict), swap the maxima, form the scaling vector
    with the setup kernel's code and do what that kernel deferred (D, ps, As / E / bs, lbs / ubs).  The same operations on
    the same values as behind k_spd_prep: identical bits, one launch less."""
    Q, p, _, _, lb, ub = O.create_qp_data(n, B, seed=n + m, with_eq=False)
    g = torch.Generator().manual_seed(n)
    A = torch.randn(B, m, n, generator=g) if m else None
    b = 0.1 * torch.randn(B, m, 1, generator=g) if m else None
    lb[0, : n // 3] = -float("inf")
    ub[B - 1, n // 2:] = float("inf")
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("LQP_QPASS", flag)
        ctl = O.make_control(rho=rho, linsolve="spd", **TOL)
        out[flag], _ = solve(dev, (Q, p, A, b, lb, ub), ctl)
        assert out[flag]["_stats"]["linsolve_used"] == 2
    s1, s0 = out["1"], out["0"]
    assert s1["_stats"]["n_launch"] == s0["_stats"]["n_launch"] - 1
    assert s1["iter"] == s0["iter"] and s1["_stats"]["n_factor"] == s0["_stats"]["n_factor"]
    for k in ("x", "z", "u", "lams") + (("rho",) if rho is None else ()) + (("nus",) if m else ()):
        assert torch.equal(s1[k], s0[k]), k
    ref = O.solve_box_qp(Q, p, A, b, lb, ub, O.make_control(rho=rho, **TOL))
    assert abs(s1["iter"] - ref["iter"]) <= ref["iter"] // 4 + 20
    assert err(s1["x"], ref["x"]) < 5e-4 * max(1.0, float(ref["x"].abs().max()))


def test_pass_over_q_inside_the_resident_sweep_sees_an_unsymmetric_matrix(dev):
    """... and its verdict on symmetry is the one of k_spd_prep: a Q that is not symmetric goes to the pivoted LU."""
    n, B = 500, 4
    Q, p, A, b, lb, ub = O.create_qp_data(n, B, seed=5)
    Q = Q.clone()
    Q[2, 400, 17] += 0.05                      # one entry of one matrix, far from the diagonal, in the last block row
    ctl = O.make_control(**TOL)
    sol, _ = solve(dev, (Q, p, A, b, lb, ub), ctl)
    assert sol["_stats"]["linsolve_used"] == 1
    ref = O.solve_box_qp(Q, p, A, b, lb, ub, O.make_control(**TOL))
    assert abs(sol["iter"] - ref["iter"]) <= O.resolve_control(ctl, n).check_solved
    if sol["iter"] == ref["iter"]:
        assert err(sol["x"], ref["x"]) < 5e-5


@pytest.mark.parametrize("n,B,scale", [(500, 4, True), (330, 3, True), (448, 2, False)])
def test_late_rho_matches_the_setup_pass(dev, monkeypatch, n, B, scale):
    """rho = ||Qs||_F / sqrt(n) (reference :200-203): with the resident sweep the squares are summed by k_spd_begin
    (which reads Q anyway) and rho is added by the sweep, instead of a second pass over Q in the setup kernel.  Same
    number up to the summation order; fixed iteration count so that the comparison is not about the stopping rule."""
    inp = O.create_qp_data(n, B, seed=n)
    out = {}
    for late in ("1", "0"):
        monkeypatch.setenv("LQP_RHO_LATE", late)
        ctl = O.make_control(max_iters=41, eps_abs=1e-12, eps_rel=1e-12, linsolve="spd", scale=scale)
        out[late], _ = solve(dev, inp, ctl)
        assert out[late]["_stats"]["linsolve_used"] == 2 and out[late]["_stats"]["factor_launches"] == 3
    r1, r0 = out["1"]["rho"].flatten().double().cpu(), out["0"]["rho"].flatten().double().cpu()
    assert float(((r1 - r0).abs() / r0).max()) < 1e-6
    ref = O.solve_box_qp(*inp, O.make_control(max_iters=41, eps_abs=1e-12, eps_rel=1e-12, scale=scale))
    assert float((r1 - ref["rho"].flatten().double()).abs().max() / ref["rho"].max()) < 1e-5
    t64 = O.solve_box_qp(*[None if t is None else t.double() for t in inp],
                         O.make_control(max_iters=41, eps_abs=1e-12, eps_rel=1e-12, scale=scale))
    for k in ("x", "u", "lams"):
        assert err(out["1"][k], out["0"][k]) < 1e-5, k
        close_or_fp64(f"late_rho_n{n}", k, out["1"][k], ref[k], t64[k], X_TOL, scale_on=scale)


@pytest.mark.parametrize("family,n,B,m", [("exp1", 500, 128, 1), ("exp1", 330, 5, 2), ("exp1", 200, 3, 0), ("exp1", 448, 40, 5),
                                          ("hard", 250, 16, 16), ("hard_ill", 250, 8, 16), ("noscale", 500, 8, 1)])
def test_sweep_on_the_float16_pipe_matches_the_float32_sweep(dev, monkeypatch, family, n, B, m):
    """Round 6: the resident sweep forms Y = P W^T and every tile update on the float16 matrix pipe, each float32 operand
    carried as two halves (csrc/lqp_f16x2.hpp: 22-24 significant bits, a power-of-two scale per 32-row block, three
    v_mfma_f32_32x32x16_f16 per 16-deep slice; reference: the LAPACK factorisation of :214-215).  Against the exact-float32
    build of the same schedule (LQP_SPD_F16=0) at a pinned iteration count, both measured from a float64 solve of the same
    inputs: the float16-pipe sweep must be inside the north-star tolerance AND no further from float64 than twice the float32
    sweep + a tenth of the tolerance.  Families: the benchmark's distribution (two and four workgroups per matrix, K = 4 ... 8),
    the hard sparse distribution with sixteen equality rows, the same with Q = G^T G + 1e-6 I (cond ~ 1e7: the auto-scaled
    Qs + rho I stays well conditioned), and scale=False with a given rho = 1e-2 (a badly scaled K: entries of its inverse up to 100)."""
    if family == "exp1":
        inp = O.create_qp_data(n, B, seed=n + m, with_eq=False)
        g = torch.Generator().manual_seed(n)
        A = torch.randn(B, m, n, generator=g) if m else None
        b = 0.1 * torch.randn(B, m, 1, generator=g) if m else None
        inp = (inp[0], inp[1], A, b, inp[4], inp[5])
        kw = {}
    elif family == "noscale":
        inp = O.create_qp_data(n, B, seed=5)
        kw = dict(scale=False, rho=1e-2, adaptive_rho=False)
    else:
        inp = [t.float() for t in O.create_hard_qp_data(n, 0.85, list(range(B)))]
        if family == "hard_ill":
            # (the generator's Q carries a ridge; here: a rank-deficient Gram matrix + 1e-6 I)
            g = torch.Generator().manual_seed(3)
            G = torch.randn(B, n // 2, n, generator=g) * (torch.rand(B, n // 2, n, generator=g) < 0.15)
            inp[0] = (G.transpose(1, 2) @ G + 1e-6 * torch.eye(n)).float()
        kw = {}
    iters = 60
    ctl = O.make_control(eps_abs=1e-12, eps_rel=1e-12, max_iters=iters + 1, linsolve="spd", **kw)
    t64 = O.solve_box_qp(*[None if t is None else t.double() for t in inp], O.make_control(eps_abs=1e-12, eps_rel=1e-12, max_iters=iters + 1, **kw))
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("LQP_SPD_F16", flag)
        out[flag], _ = solve(dev, inp, ctl)
        st = out[flag]["_stats"]
        assert st["linsolve_used"] == 2 and out[flag]["iter"] == iters, st
    scale = max(1.0, float(t64["x"].abs().max()))
    for k in ("x", "u") + (("nus",) if m else ()):
        e16, e32 = err(out["1"][k], t64[k]), err(out["0"][k], t64[k])
        sc = max(1.0, float(t64[k].abs().max()))
        P.record(f"f16_sweep_{family}_n{n}_B{B}_m{m}", k, e16, sc, hip_vs_fp64=e16, float32_sweep_vs_fp64=e32, tol=X_TOL * sc)
        assert e16 <= X_TOL * sc and e16 <= 2 * e32 + 0.1 * X_TOL * sc, (family, k, e16, e32)


@pytest.mark.parametrize("n,B,m", [(500, 4, 1), (330, 3, 2), (448, 2, 5)])
def test_equality_correction_in_the_loop_kernel(dev, monkeypatch, n, B, m):
    """With two workgroups per QP the equality correction H + T G^T of the first factorisation is applied to the blocks
    in the loop kernel's registers (no k_spd_end launch, no three passes over H in global memory).  Same mathematics as
    wg_eq_correct, another summation order: compared at a fixed iteration count with that path and with the oracle, and
    through the constraint itself.  rho = 100 forces refactorisations: the continuation launches must find corrected
    blocks in global memory."""
    Q, p, _, _, lb, ub = O.create_qp_data(n, B, seed=n + m, with_eq=False)
    g = torch.Generator().manual_seed(n)
    A = torch.randn(B, m, n, generator=g)
    b = 0.1 * torch.randn(B, m, 1, generator=g)
    for kw in (dict(max_iters=41, eps_abs=1e-12, eps_rel=1e-12), dict(rho=100.0, **TOL)):
        out = {}
        for flag in ("1", "0"):
            monkeypatch.setenv("LQP_EQ_IN_LOOP", flag)
            out[flag], _ = solve(dev, (Q, p, A, b, lb, ub), O.make_control(linsolve="spd", **kw))
            assert out[flag]["_stats"]["linsolve_used"] == 2 and out[flag]["_stats"]["loop_workgroups"] in (2, 4)
        ref = O.solve_box_qp(Q, p, A, b, lb, ub, O.make_control(**kw))
        scale = max(1.0, float(ref["x"].abs().max()))
        if "rho" in kw:
            # (with the stopping rule live -- and rho adapted from a ratio of residuals -- the solution is only as good
            #  as the tolerance: compare the primal solution, at the accuracy the stopping rule gives)
            # (the iteration count of this run is decided by a ratio of two residuals at rounding level: the reference's float32
            #  arithmetic stops at 240 at n = 330, its float64 run at 340 -- either is a legitimate answer)
            ref64 = O.solve_box_qp(*[t.double() for t in (Q, p, A, b, lb, ub)], O.make_control(**kw))
            assert out["1"]["_stats"]["n_factor"] >= 2
            assert min(abs(out["1"]["iter"] - r["iter"]) - (r["iter"] // 4 + 20) for r in (ref, ref64)) <= 0, (out["1"]["iter"], ref["iter"], ref64["iter"])
            assert err(out["1"]["x"], out["0"]["x"]) < 5e-4 * scale and err(out["1"]["x"], ref["x"]) < 5e-4 * scale, kw
        else:
            t64 = O.solve_box_qp(*[t.double() for t in (Q, p, A, b, lb, ub)], O.make_control(**kw))
            for k in ("x", "u", "lams", "nus"):
                close_or_fp64(f"eq_in_loop_n{n}_m{m}", k, out["1"][k], ref[k], t64[k], X_TOL)
                close_or_fp64(f"eq_in_loop_n{n}_m{m}_off", k, out["0"][k], ref[k], t64[k], X_TOL)
        assert float((A.to(dev) @ out["1"]["x"] - b.to(dev)).abs().max()) < 1e-4


@pytest.mark.parametrize("n,B,m,rho,qscale", [(330, 16, 1, 0.01, 1.0), (500, 8, 1, 0.01, 1.0), (200, 6, 2, 100.0, 50.0)])
def test_hot_loop_goes_on_behind_rho_events(dev, monkeypatch, n, B, m, rho, qscale):
    """Round 6: with a GIVEN rho (the case in which the reference's adaptation, :237-256, does fire) the library enqueues rounds of
    {the event on the gated kernels of the refactorisation, the register-resident two-workgroup loop again from where it stopped}
    behind the first hot launch (LQP_HOT_ROUNDS, default 2), so that the iterations behind an event no longer run on the one-workgroup
    continuation kernel.  Against LQP_HOT_ROUNDS=0 (everything behind iteration 100 on the continuation kernel) and the oracle: the same
    number of factorisations, the same iteration count (the stop is decided at 1e-5; two summation orders), x within the tolerance."""
    Q, p, _, _, lb, ub = O.create_qp_data(n, B, seed=n + m, with_eq=False)
    Q = Q * qscale
    g = torch.Generator().manual_seed(n)
    A = torch.randn(B, m, n, generator=g)
    b = 0.1 * torch.randn(B, m, 1, generator=g)
    kw = dict(rho=rho, scale=(qscale == 1.0), **TOL)
    out = {}
    for rounds in ("2", "0"):
        monkeypatch.setenv("LQP_HOT_ROUNDS", rounds)
        out[rounds], _ = solve(dev, (Q, p, A, b, lb, ub), O.make_control(linsolve="spd", **kw))
        st = out[rounds]["_stats"]
        assert st["linsolve_used"] == 2 and st["loop_workgroups"] in (2, 4), st
    ref = O.solve_box_qp(Q, p, A, b, lb, ub, O.make_control(**kw))
    ref64 = O.solve_box_qp(*[t.double() for t in (Q, p, A, b, lb, ub)], O.make_control(**kw))
    s2, s0 = out["2"], out["0"]
    assert s2["_stats"]["n_factor"] >= 2 and s2["_stats"]["n_factor"] == s0["_stats"]["n_factor"], (s2["_stats"], s0["_stats"])
    assert s2["_stats"]["n_launch"] > s0["_stats"]["n_launch"]                   # (the rounds were enqueued)
    assert s2["iter"] == s0["iter"] or min(abs(s2["iter"] - r["iter"]) for r in (ref, ref64)) <= 20, (s2["iter"], s0["iter"], ref["iter"], ref64["iter"])
    scale_x = max(1.0, float(ref64["x"].abs().max()))
    P.record(f"hot_rounds_n{n}_B{B}", "x", err(s2["x"], ref64["x"]), scale_x, one_workgroup_tail_vs_fp64=err(s0["x"], ref64["x"]))
    assert err(s2["x"], s0["x"]) < 5e-4 * scale_x and err(s2["x"], ref64["x"]) < 5e-4 * scale_x
    assert float((A.to(dev) @ s2["x"] - b.to(dev)).abs().max()) < 1e-3


@pytest.mark.parametrize("n,B,m", [(333, 64, 1), (500, 17, 1), (129, 17, 3), (320, 8, 16)])
def test_backward_on_a_badly_scaled_q(dev, monkeypatch, n, B, m):
    """The fixed-point backward factorises the free-set block of the UNSCALED Q (reference :378-393: no pre-conditioning there).  Round 6
    equilibrates it symmetrically by powers of two (exact: the float32 Cholesky sees the same significands) so that the float16-pipe
    tile products may scale a 32-row block by one factor: on Q := D Q D, d = 10^U(-1.5, 1.5) -- rows a thousand times apart in
    magnitude -- dp is held to the float64 oracle's at 5e-6 of its scale (without the equilibration the float16-pipe build measured
    5e-5, the float32 build 1e-6), and LQP_BWD_EQUIL=0 / LQP_BWD_F16=0 stay within rtol 1e-4."""
    Q, p, _, _, lb, ub = O.create_qp_data(n, B, seed=2000 + n, with_eq=False)
    g = torch.Generator().manual_seed(n + m)
    d = 10.0 ** (3 * torch.rand(B, n, generator=g) - 1.5)
    Q = d.unsqueeze(2) * Q * d.unsqueeze(1)
    A = torch.randn(B, m, n, generator=g)
    b = 0.1 * torch.randn(B, m, 1, generator=g)
    cot = torch.randn(B, n, 1, generator=g)
    iters = 40
    ctl = O.make_control(eps_abs=1e-12, eps_rel=1e-12, max_iters=iters + 1)
    inp = (Q, p, A, b, lb, ub)
    t64 = O.solve_box_qp(*[t.double() for t in inp], dict(ctl))
    g64 = O.solve_box_qp_grad(cot.double(), t64["x"], t64["u"], t64["lams"], t64["nus"], Q.double(), A.double(), lb.double(), ub.double(), t64["rho"])
    sp, sq = float(g64[1].abs().max()), float(g64[0].abs().max())
    args = [t.to(dev) for t in inp]
    for env, tol in (({}, 5e-6), ({"LQP_BWD_EQUIL": "0"}, G_RTOL), ({"LQP_BWD_F16": "0"}, 5e-6)):
        for k_, v_ in (("LQP_BWD_EQUIL", "1"), ("LQP_BWD_F16", "1")):
            monkeypatch.setenv(k_, env.get(k_, v_))
        Ql, pl = args[0].clone().requires_grad_(True), args[1].clone().requires_grad_(True)
        x = L.SolveBoxQP(control=dict(ctl))(Ql, pl, *args[2:])
        x.backward(cot.to(dev))
        ep, eq = err(pl.grad, g64[1]) / sp, err(Ql.grad, g64[0]) / sq
        P.record(f"bwd_badly_scaled_n{n}_B{B}_m{m}", "dp_" + ("default" if not env else "_".join(f"{a}={c}" for a, c in env.items())), ep * sp, sp, tol=tol * sp)
        assert ep <= tol and eq <= G_RTOL, (env, ep, eq)


@pytest.mark.parametrize("n,B,m", [(330, 600, 3), (500, 520, 16), (200, 300, 2)])
def test_large_batch_sweep_on_pairs_taking_turns(dev, monkeypatch, n, B, m):
    """Round 6: more matrices than half the CUs run the register-resident sweep too, its PAIRS of workgroups taking turns on the chip
    (LQP_SPD_TURNS; before: k_spd_prep + the one-workgroup sweep, every tile through L2 / HBM in every pivot step); from two turns
    on (B >= 2 #CUs) the first check segment of the loop applies the equality correction.  Against LQP_SPD_TURNS=0: the same iteration
    count (the stop is decided by all problems), x within the tolerance of two schedules; eight problems against the float64 oracle
    at the same count; the constraint itself; the backward on top."""
    Q, p, _, _, lb, ub = O.create_qp_data(n, B, seed=n + B, with_eq=False)
    g = torch.Generator().manual_seed(n)
    A = torch.randn(B, m, n, generator=g)
    b = 0.1 * torch.randn(B, m, 1, generator=g)
    cot = torch.randn(B, n, 1, generator=g)
    args = [t.to(dev) for t in (Q, p, A, b, lb, ub)]
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("LQP_SPD_TURNS", flag)
        Ql, pl = args[0].clone().requires_grad_(True), args[1].clone().requires_grad_(True)
        x = L.SolveBoxQP(control=L.box_qp_control(**TOL))(Ql, pl, *args[2:])
        st = SB.last_forward_status(dev)
        x.backward(cot.to(dev))
        out[flag] = (x.detach(), pl.grad, st)
        assert st["linsolve_used"] == 2 and st["loop_workgroups_per_qp"] == 2, st
        assert st["factor_launches"] == (3 if flag == "1" else 1), st          # (turns: begin / resident sweep / end; else the one-workgroup sweep)
    (x1, dp1, st1), (x0, dp0, st0) = out["1"], out["0"]
    assert st1["iters"] == st0["iters"]
    idx = torch.arange(0, B, B // 8)[:8]
    sub = [t[idx].double() for t in (Q, p, A, b, lb, ub)]
    t64 = O.solve_box_qp(*sub, O.make_control(eps_abs=1e-12, eps_rel=1e-12, max_iters=st1["iters"] + 1))
    g64 = O.solve_box_qp_grad(cot[idx].double(), t64["x"], t64["u"], t64["lams"], t64["nus"], sub[0], sub[2], sub[4], sub[5], t64["rho"])
    sx = max(1.0, float(t64["x"].abs().max()))
    P.record(f"sweep_in_turns_n{n}_B{B}_m{m}", "x", err(x1[idx.to(dev)], t64["x"]), sx, one_workgroup_sweep_vs_fp64=err(x0[idx.to(dev)], t64["x"]))
    assert err(x1[idx.to(dev)], t64["x"]) <= X_TOL * sx and err(x1, x0) <= 2 * X_TOL * sx
    assert rel(dp1[idx.to(dev)], g64[1]) <= G_RTOL
    assert float((args[2] @ x1 - args[3]).abs().max()) < 2e-4


def test_continuation_launch_finds_corrected_blocks(dev):
    """The two-workgroup loop kernel applies the equality correction to its register blocks; when the loop has to go on in
    a continuation launch (here: tolerances that are never met, more iterations than one launch may hold, no adaptive rho
    so that nothing is refactorised) the blocks must have been written back WITH the correction: the continuation reads
    them from global memory."""
    n, B, m = 300, 2, 2
    Q, p, _, _, lb, ub = O.create_qp_data(n, B, seed=77, with_eq=False)
    g = torch.Generator().manual_seed(77)
    A = torch.randn(B, m, n, generator=g)
    b = 0.1 * torch.randn(B, m, 1, generator=g)
    kw = dict(max_iters=5135, eps_abs=1e-30, eps_rel=1e-30, adaptive_rho=False)
    sol, _ = solve(dev, (Q, p, A, b, lb, ub), O.make_control(**kw))
    st = sol["_stats"]
    assert st["linsolve_used"] == 2 and st["loop_workgroups"] in (2, 4) and st["n_factor"] == 1 and sol["iter"] == 5134
    ref = O.solve_box_qp(Q, p, A, b, lb, ub, O.make_control(max_iters=400, eps_abs=1e-30, eps_rel=1e-30, adaptive_rho=False))
    for k in ("x", "u", "nus"):
        assert err(sol[k], ref[k]) < 5e-5 * max(1.0, float(ref[k].abs().max())), k        # (both sit at the fixed point)
    assert float((A.to(dev) @ sol["x"] - b.to(dev)).abs().max()) < 1e-5


@pytest.mark.parametrize("n,B,m", [(576, 3, 2), (1000, 2, 1), (1024, 2, 0)])
@pytest.mark.parametrize("split", ["1", "0"])
def test_symmetric_path_above_512(dev, monkeypatch, n, B, m, split):
    """512 < n <= 1024: the sweep parks its panel in L2 (wg_spd_sweep_big) -- one workgroup per matrix in one launch, or
    two per matrix with a launch per phase of a pivot step.  Against the oracle, and against the LU path of the same
    call; rho = 100 forces adaptive-rho refactorisations (in-kernel in the persistent mode)."""
    monkeypatch.setenv("LQP_SPD_SPLIT", split)
    Q, p, _, _, lb, ub = O.create_qp_data(n, B, seed=n + m, with_eq=False)
    g = torch.Generator().manual_seed(n)
    A = torch.randn(B, m, n, generator=g) if m else None
    b = 0.1 * torch.randn(B, m, 1, generator=g) if m else None
    for kw in (dict(max_iters=41, eps_abs=1e-12, eps_rel=1e-12), dict(rho=100.0, **TOL)):
        sol, _ = solve(dev, (Q, p, A, b, lb, ub), O.make_control(**kw))
        assert sol["_stats"]["linsolve_used"] == 2
        assert sol["_stats"]["factor_launches"] == (2 * ((n + 63) // 64) + 2 if split == "1" else 1)
        lu, _ = solve(dev, (Q, p, A, b, lb, ub), O.make_control(linsolve="lu", **kw))
        ref = O.solve_box_qp(Q, p, A, b, lb, ub, O.make_control(**kw))
        dbl = [None if t is None else t.double() for t in (Q, p, A, b, lb, ub)]
        t64 = O.solve_box_qp(*dbl, O.make_control(**kw))          # the same solve in float64 (same control, same stop rule)
        case = f"symmetric_above_512_n{n}_split{split}" + ("_adaptive" if "rho" in kw else "")
        if "rho" in kw:
            # rho after a refactorisation is the ratio of two 1e-6-level residuals: float32 implementations legitimately take
            # different numbers of checks -- no further from the reference's count than the float64 solve is (+ one check)
            chk = max(round((n ** 0.5) / 10) * 10, 1)
            assert sol["_stats"]["n_factor"] >= 2
            assert abs(sol["iter"] - ref["iter"]) <= abs(t64["iter"] - ref["iter"]) + chk, (sol["iter"], ref["iter"], t64["iter"])
            close_or_fp64(case, "x", sol["x"], ref["x"], t64["x"], X_TOL)
        else:
            # 41 iterations from a cold start: north-star tolerance, or no further from float64 than the reference's own float32
            for k in ("x", "u", "lams") + (("nus",) if m else ()):
                close_or_fp64(case, k, sol[k], ref[k], t64[k], X_TOL, lu_path=err(lu[k], ref[k]))
                close_or_fp64(case + "_lu", k, lu[k], ref[k], t64[k], X_TOL)


# ---------------------------------------------------------------- config 5: the per-GPU shard of B=8192 over 8 GPUs
def test_config5_shard_b1024_n500(dev):
    """BASELINE configs[4]: batch 8192 dz 500 sharded over 8 GPUs = 1024 QPs per GPU, forward + backward through the
    module.  B > 256 CUs: one launch per check segment; since round 4 the loop's launches are those of the two-workgroup kernel,
    its pairs taking turns on the chip (FwdParams::split_seg).  Checked against the CPU oracle on 16
    problems (its iteration count pinned to the GPU's: the stop is decided by ALL 1024 problems) and, at full size,
    through the KKT conditions."""
    B, n = 1024, 500
    inp = O.create_qp_data(n, B, seed=0)
    Q, p, A, b, lb, ub = (t.to(dev) for t in inp)
    Qg, pg = Q.clone().requires_grad_(True), p.clone().requires_grad_(True)
    x = L.SolveBoxQP(control=L.box_qp_control(**TOL))(Qg, pg, A, b, lb, ub)
    st = SB.last_forward_status(dev)
    assert st["mode_used"] == 1 and st["linsolve_used"] == 2 and st["loop_workgroups_per_qp"] == 2
    # the reference's own run of this shard (tests/golden/make_golden.py, G19: B = 1024, n = 500, seed 0): iteration count -- decided by
    # ALL 1024 problems, :312 -- and x
    g19 = load_golden("g19_b1024_n500_eq")
    # (checksums of the inputs: Q = L^T L / 2n is a float32 matrix product of the HOST -- its last bits follow the host's BLAS
    #  kernels, 2e-9 of the sum between the build container and this box; p, lb, ub are draws: exact)
    sums = np.array([float(t.double().sum()) for t in (inp[0], inp[1], inp[4], inp[5])])
    assert np.allclose(sums[0], g19["in_sum"].numpy()[0], rtol=1e-7, atol=0) and np.array_equal(sums[1:], g19["in_sum"].numpy()[1:]), "generator drifted"
    assert st["iters"] == int(g19["iter"]) == 60, (st["iters"], int(g19["iter"]))
    torch.manual_seed(3)
    cot = torch.randn(B, n, 1)
    x.backward(cot.to(dev))
    sol = L.torch_solve_box_qp(Q, p, A, b, lb, ub, L.box_qp_control(**TOL))
    assert sol["iter"] == st["iters"] and torch.equal(sol["x"], x.detach())
    # ---- 16 problems against the oracle at the same iteration count ----
    idx = torch.arange(0, B, B // 16)
    sub = [t[idx] for t in inp]
    ref = O.solve_box_qp(*sub, O.make_control(eps_abs=1e-12, eps_rel=1e-12, max_iters=st["iters"] + 1))
    t64, g64 = fp64_truth(sub, st["iters"], cots=(cot[idx],))
    case = "config5_shard_b1024_n500"
    for k in ("x", "u", "nus", "lams"):
        close_or_fp64(case, k, sol[k][idx.to(dev)], ref[k], t64[k], X_TOL)
    # ... all 1024 problems against the reference-made vector at 1 x the tolerance (rho: the reference's float32 sum of 250 000 squares)
    for k, tol in (("x", X_TOL), ("u", 2 * X_TOL), ("rho", X_TOL)):
        sc = max(1.0, float(g19[k].abs().max()))
        P.record(case, f"g19_{k}", err(sol[k], g19[k]), sc, tol=tol * sc)
        assert err(sol[k], g19[k]) <= tol * sc, (k, err(sol[k], g19[k]))
    gref = O.solve_box_qp_grad(cot[idx], ref["x"], ref["u"], ref["lams"], ref["nus"], sub[0], sub[2], sub[4], sub[5], ref["rho"])
    for nm, t, r, r64 in (("dp", pg.grad, gref[1], g64[0][1]), ("dQ", Qg.grad, gref[0], g64[0][0])):
        close_or_fp64(case, nm, t[idx.to(dev)], r, r64, G_RTOL)
    # ---- full size: KKT conditions ----
    cpu = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in sol.items()}
    res = O.kkt_residuals(*inp, cpu)
    assert float(res["stationarity"].max()) < 2e-3 and float(res["equality"].max()) < 2e-4
    assert float(res["box"].max()) < 1e-6 and float(res["x_minus_z"].max()) < 2e-4
    assert float((cpu["lams"] < 0).sum()) == 0 and torch.isfinite(Qg.grad).all()


@pytest.mark.parametrize("B,n,m,rho", [(264, 500, 1, None), (272, 330, 2, 100.0), (520, 448, 0, None), (136, 330, 2, None)])
def test_large_batch_loop_on_pairs_taking_turns(dev, monkeypatch, B, n, m, rho):
    """More problems than half the CUs: the loop is launched once per check segment, and since round 4 those launches are
    the two-workgroup kernel's (a pair holds its whole matrix in registers for the segment), the pairs taking their turns on
    the chip.  Against the one-workgroup launches (LQP_LOOP_SPLIT_SEG=0): the same iteration count (the stop is decided by
    all problems), iterates within the tolerance of two summation orders; rho = 100 forces refactorisations between the
    segments; the CPU oracle on a few problems at the same iteration count."""
    Q, p, _, _, lb, ub = O.create_qp_data(n, B, seed=B + m, with_eq=False)
    g = torch.Generator().manual_seed(n)
    A = torch.randn(B, m, n, generator=g) if m else None
    b = 0.1 * torch.randn(B, m, 1, generator=g) if m else None
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("LQP_LOOP_SPLIT_SEG", flag)
        out[flag], _ = solve(dev, (Q, p, A, b, lb, ub), O.make_control(rho=rho, linsolve="spd", **TOL))
        st = out[flag]["_stats"]
        # (B = 136: all workgroups of the one-workgroup kernel are resident -- without the pairs it runs persistently)
        assert st["linsolve_used"] == 2 and st["loop_workgroups"] == (2 if flag == "1" else 1) and (st["mode_used"] == 1 or flag == "0")
    s1, s0 = out["1"], out["0"]
    assert s1["iter"] == s0["iter"] and s1["_stats"]["n_factor"] == s0["_stats"]["n_factor"]
    if rho is not None:
        assert s1["_stats"]["n_factor"] >= 2
    # (rho = 100 is a hundred times too large: at the checks the adaptation reads, the primal residual is at rounding level, the
    #  ratio sqrt(r / s) it multiplies rho with moves by 0.5 % between two summation orders -- 0.06 ... 0.7 % between the CPU
    #  oracle and either of them -- and u = lam / rho with it: only what does not depend on rho is compared there)
    for k in (("x", "z") if rho is not None else ("x", "z", "u", "lams") + (("nus",) if m else ())):
        assert err(s1[k], s0[k]) < 2e-5 * max(1.0, float(s0[k].abs().max())), k
    idx = torch.arange(0, B, B // 4)
    sub = [None if t is None else t[idx] for t in (Q, p, A, b, lb, ub)]
    if rho is None:      # (with refactorisations the oracle's rho history depends on the whole batch)
        ref = O.solve_box_qp(*sub, O.make_control(eps_abs=1e-12, eps_rel=1e-12, max_iters=s1["iter"] + 1))
        assert err(s1["x"][idx.to(dev)], ref["x"]) < 5e-5


# ---------------------------------------------------------------- the reference's "hard" distribution in float32
@pytest.mark.parametrize("shift", [1e-2, 1e-3, 1e-4, 0.0])
def test_hard_distribution_fp32_and_conditioning_sweep(dev, shift):
    """experiments/utils.py:64-131 (sparse G, Q = G^T G + shift I, sqrt(n) equality rows) in FLOAT32.  shift = 1e-2
    is the reference's generator; smaller shifts push the condition number of Q up.  Both x-updates are checked
    against an fp64 solve of the same inputs, next to the fp32 CPU oracle (= the reference's arithmetic): the explicit
    inverse of the default path must not be further from fp64 than 4x the pivoted LU's error (+1e-4), or it must
    have fallen back to LU by itself.  Conditioning and errors go to the parity report."""
    n, seeds = 100, list(range(8))
    inp64 = list(O.create_hard_qp_data(n, 0.85, seeds))
    inp64[0] = inp64[0] + (shift - 1e-2) * torch.eye(n, dtype=torch.float64)
    inp32 = [t.float() for t in inp64]
    cond = float(torch.linalg.cond(inp64[0]).max())
    ctl = O.make_control(**TOL)
    o32 = O.solve_box_qp(*inp32, dict(ctl))
    s64 = O.solve_box_qp(*inp64, dict(ctl))
    scale = max(1.0, float(s64["x"].abs().max()))
    e_o32 = err(o32["x"], s64["x"])
    errs, used = {}, {}
    for ls in ("lu", "auto"):
        sol, _ = solve(dev, inp32, dict(ctl, linsolve=ls))
        assert torch.isfinite(sol["x"]).all()
        errs[ls], used[ls] = err(sol["x"], s64["x"]), sol["_stats"]["linsolve_used"]
        res = O.kkt_residuals(*inp32, {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in sol.items()})
        P.record("hard_fp32", "x", errs[ls], scale, linsolve=ls, linsolve_used=used[ls], shift=shift, cond_Q=cond,
                 iters=sol["iter"], iters_oracle_fp32=o32["iter"], iters_fp64=s64["iter"], oracle_fp32_vs_fp64=e_o32,
                 stationarity=float(res["stationarity"].max()), equality=float(res["equality"].max()))
        assert float(res["box"].max()) < 1e-5
    assert used["lu"] == 1
    # the stopping test is an fp32 threshold: all three fp32 runs land within the tolerance band of fp64
    band = 50 * TOL["eps_abs"] * scale + 4 * e_o32
    assert errs["lu"] < band, (errs, e_o32)
    assert used["auto"] == 1 or errs["auto"] < 4 * errs["lu"] + 1e-4 * scale, (errs, used, cond)


def test_per_problem_beta_tensor(dev):
    """control['beta'] may be a (B,1) tensor (reference :171-175 broadcasts it against D)."""
    inp = O.create_qp_data(30, 5, seed=2)
    beta = torch.tensor([[0.1], [0.3], [0.5], [0.7], [0.9]])
    ref = O.solve_box_qp(*inp, O.make_control(beta=beta, **TOL))
    for ls in ("lu", "spd"):
        sol, _ = solve(dev, inp, O.make_control(beta=beta.to(dev), linsolve=ls, **TOL))
        assert sol["iter"] == ref["iter"]
        for k in ("x", "u", "lams", "nus", "rho"):
            assert err(sol[k], ref[k]) < 2e-5, (ls, k)
    one = O.solve_box_qp(*inp, O.make_control(beta=0.3, **TOL))
    sol, _ = solve(dev, inp, O.make_control(beta=torch.tensor(0.3), **TOL))
    assert err(sol["x"], one["x"]) < 2e-5
    with pytest.raises(ValueError, match="beta"):
        solve(dev, inp, O.make_control(beta=torch.ones(3, 1), **TOL))


def test_two_streams_do_not_share_a_workspace(dev):
    """Workspaces are keyed by (device, stream, tag): two forwards in flight on two streams keep their own factors."""
    a = [t.to(dev) for t in O.create_qp_data(300, 16, seed=1)]
    b = [t.to(dev) for t in O.create_qp_data(300, 16, seed=2)]
    ctl = dict(L.box_qp_control(**TOL), sync=False)
    ref_a = L.SolveBoxQP(control=dict(ctl))(*a)
    ref_b = L.SolveBoxQP(control=dict(ctl))(*b)
    L.synchronize()
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    torch.cuda.synchronize(dev)
    with torch.cuda.stream(s1):
        xa = L.SolveBoxQP(control=dict(ctl))(*a)
    with torch.cuda.stream(s2):
        xb = L.SolveBoxQP(control=dict(ctl))(*b)
    torch.cuda.synchronize(dev)
    L.synchronize()
    assert torch.equal(xa, ref_a) and torch.equal(xb, ref_b)


def test_lqp_py_alias_resolves_to_the_hip_layer(dev):
    from lqp_py.solve_box_qp_admm_torch import SolveBoxQP, torch_solve_box_qp
    from lqp_py.control import box_qp_control
    assert SolveBoxQP is L.SolveBoxQP and torch_solve_box_qp is L.torch_solve_box_qp
    inp = [t.to(dev) for t in O.create_qp_data(20, 3, seed=0)]
    x = SolveBoxQP(control=box_qp_control(**TOL))(*inp)
    assert x.is_cuda and err(x, O.solve_box_qp(*O.create_qp_data(20, 3, seed=0), O.make_control(**TOL))["x"]) < 2e-5


@pytest.mark.parametrize("n,m", [(1000, 10), (760, 16), (960, 8)])
def test_many_equality_rows_at_large_n_take_the_lu_path(dev, n, m):
    """The rank-m equality correction keeps G and T (2 m rows of 64 Ks floats) in LDS next to the product's scratch:
    above 160 KB (n > 960 at m >= 8, n > 896 at m >= 9, n > 704 at m >= 15) 'auto' must stay on the pivoted LU instead
    of failing with 'unsupported size'.  Results against the CPU oracle."""
    B = 2
    torch.manual_seed(n + m)
    Q, p, _, _, lb, ub = O.create_qp_data(n, B, seed=n + m, with_eq=False)
    A = torch.randn(B, m, n)
    b = 0.1 * torch.randn(B, m, 1)
    sol, _ = solve(dev, (Q, p, A, b, lb, ub), O.make_control(**TOL))
    ref = O.solve_box_qp(Q, p, A, b, lb, ub, O.make_control(**TOL))
    assert sol["iter"] == ref["iter"]
    assert sol["_stats"]["linsolve_used"] in (1, 2)
    if n * 1 > 960 or m >= 15:
        assert sol["_stats"]["linsolve_used"] == 1
    P.record("lds_budget", "x", err(sol["x"], ref["x"]), 1.0, n=n, m=m)
    assert err(sol["x"], ref["x"]) < 3e-5
    assert err(sol["nus"], ref["nus"]) < 2e-3 * max(1.0, float(ref["nus"].abs().max()))


def test_bound_flags_come_from_the_device(dev):
    """Whether any bound is finite (:33-38, :129-131) is found by the setup kernel from the DATA of every call: fresh
    tensors cost no host look, a tensor rewritten through .data is seen, and a batch without any finite bound takes the
    reference's rho = 0 one-shot solve (with the dict side effect) whatever the previous call looked like."""
    n, B = 40, 5
    Q, p, A, b, lb, ub = (t.to(dev) for t in O.create_qp_data(n, B, seed=9))
    ref = O.solve_box_qp(*(t.cpu() for t in (Q, p, A, b, lb, ub)), O.make_control(**TOL))
    inf = float("inf")
    lb_none, ub_none = torch.full_like(lb, -inf), torch.full_like(ub, inf)
    ref0 = O.solve_box_qp(Q.cpu(), p.cpu(), A.cpu(), b.cpu(), lb_none.cpu(), ub_none.cpu(), O.make_control(**TOL))
    ctl = L.box_qp_control(**TOL)
    layer = L.SolveBoxQP(control=ctl)
    x1 = layer(Q, p, A, b, lb, ub)
    assert "rho" not in ctl or ctl["rho"] is None
    assert err(x1, ref["x"]) < 2e-5
    st = SB.last_forward_status(dev)
    assert st["any_lb"] == 1 and st["any_ub"] == 1 and st["iters"] == ref["iter"]
    # the same dict, now a batch without bounds: one-shot solve, dict mutated -- although the layer assumed bounds
    x0 = layer(Q, p, A, b, lb_none, ub_none)
    assert ctl["rho"] == 0
    assert SB.last_forward_status(dev)["iters"] == ref0["iter"] == 0
    assert err(x0, ref0["x"]) < 2e-5
    # only one side finite: the other clamp is an exact no-op
    ref_lb = O.solve_box_qp(Q.cpu(), p.cpu(), A.cpu(), b.cpu(), lb.cpu(), ub_none.cpu(), O.make_control(**TOL))
    sol_lb = L.torch_solve_box_qp(Q, p, A, b, lb, ub_none, L.box_qp_control(**TOL))
    assert sol_lb["iter"] == ref_lb["iter"] and err(sol_lb["x"], ref_lb["x"]) < 2e-5
    assert sol_lb["_stats"]["any_lb"] == 1 and sol_lb["_stats"]["any_ub"] == 0
    # a tensor rewritten through .data (same object, same _version) is looked at again
    ctl2 = L.box_qp_control(**TOL)
    lbm, ubm = lb.clone(), ub.clone()
    xa = L.SolveBoxQP(control=ctl2)(Q, p, A, b, lbm, ubm)
    lbm.data.fill_(-inf)
    ubm.data.fill_(inf)
    xb = L.SolveBoxQP(control=ctl2)(Q, p, A, b, lbm, ubm)
    assert ctl2["rho"] == 0 and err(xa, ref["x"]) < 2e-5 and err(xb, ref0["x"]) < 2e-5


def test_the_callers_control_dict_stays_clean(dev, monkeypatch):
    """what the layer remembers between calls (did the last batch hold a finite bound?) lives in a side table
    keyed by the module, never in the caller's dict: the dict keeps its keys, and two modules sharing ONE dict -- one fed
    bounded batches, one unbounded ones -- do not make each other repeat solves."""
    Q, p, A, b, lb, ub = (t.to(dev) for t in O.create_qp_data(40, 3, seed=5))
    inf = torch.full_like(lb, float("inf"))
    control = L.box_qp_control(**TOL)
    keys = set(control)
    bounded, free = L.SolveBoxQP(control=control), L.SolveBoxQP(control=control)
    calls = []
    inner = SB._forward_solve
    monkeypatch.setattr(SB, "_forward_solve", lambda *a, **k: (calls.append(1), inner(*a, **k))[1])
    per_round = []
    for _ in range(3):
        calls.clear()
        xb = bounded(Q, p, A, b, lb, ub)
        xf = free(Q, p, A, b, -inf, inf)
        assert control["rho"] == 0                       # the reference's own side effect (:37-38) ...
        control["rho"] = None                            # ... undone by the caller between the rounds
        per_round.append(len(calls))
    assert set(control) == keys, set(control) ^ keys
    assert SB._seen_by_module[bounded] is True and SB._seen_by_module[free] is False
    assert per_round[0] == 3 and per_round[1:] == [2, 2], per_round      # (only `free`'s first call repeated itself)
    ref = O.solve_box_qp(*[t.cpu() for t in (Q, p, A, b, lb, ub)], O.make_control(**TOL))
    assert err(xb, ref["x"]) < 2e-5 and torch.isfinite(xf).all()


def test_workgroups_can_ask_which_xcd_they_run_on(dev):
    """HW_REG_XCC_ID, the question behind the XCD-aware exchange: ids are 0..7, an MI355X in its default mode shows more than
    one of them over a 256-workgroup launch, and -- recorded, not asserted: placement is the dispatcher's -- how often
    workgroups b and b + 128 (the two that share a matrix at B = 128) report the same one."""
    lib = _lib.load()
    out = torch.full((256,), -1, dtype=torch.int32, device=dev)
    _lib.check(lib.lqp_debug_xcd(_lib.stream_ptr(dev), 256, _lib.ptr(out)), "debug_xcd")
    ids = out.cpu()
    assert int(ids.min()) >= 0 and int(ids.max()) <= 7
    together = float((ids[:128] == ids[128:]).float().mean())
    P.record("xcd_placement", "observation", 0.0, 1.0, pairs_on_one_xcd=together, distinct_ids=int(ids.unique().numel()))
    assert ids.unique().numel() >= 1


@pytest.mark.parametrize("n,B,m", [(500, 8, 1), (500, 16, 0), (448, 5, 2), (512, 3, 0)])
def test_xcd_local_exchange_is_only_a_transport(dev, monkeypatch, n, B, m):
    """Workgroups that share a matrix and find themselves on one XCD (B a multiple of 8 under round-robin placement) exchange
    tiles and granules through that XCD's L2 (workgroup-scope stores); otherwise, and with LQP_XCD_LOCAL=0, through
    write-through stores.  Only the transport differs: bit-identical outputs, same iteration count."""
    torch.manual_seed(n + 3 * B)
    Q, p, _, _, lb, ub = O.create_qp_data(n, B, seed=5 * n + B, with_eq=False)
    A = torch.randn(B, m, n) if m else None
    b = 0.1 * torch.randn(B, m, 1) if m else None
    ctl = O.make_control(**TOL)
    sols = {}
    for knob in ("1", "0"):
        monkeypatch.setenv("LQP_XCD_LOCAL", knob)
        sols[knob], _ = solve(dev, (Q, p, A, b, lb, ub), ctl)
        assert sols[knob]["_stats"]["factor_launches"] == 3 and sols[knob]["_stats"]["loop_workgroups"] >= 2
    assert sols["1"]["iter"] == sols["0"]["iter"]
    for k in ("x", "z", "u", "lams") + (("nus",) if m else ()):
        assert torch.equal(sols["1"][k], sols["0"][k]), k
    ref = O.solve_box_qp(Q, p, A, b, lb, ub, ctl)
    assert sols["1"]["iter"] == ref["iter"] and err(sols["1"]["x"], ref["x"]) < 2e-5 * max(1.0, float(ref["x"].abs().max()))


def test_control_struct_cache_follows_the_dict(dev):
    """The resolved control struct of the last call with the same settings is reused (host time in front of the first
    launch); a changed value -- in place or in another dict -- must reach the library, tensors are never cached."""
    inp = [t.to(dev) for t in O.create_qp_data(40, 4, seed=11)]
    ctl = dict(L.box_qp_control(eps_abs=1e-3, eps_rel=1e-3))
    loose = L.torch_solve_box_qp(*inp, ctl)
    again = L.torch_solve_box_qp(*inp, ctl)                       # served from the cache
    assert loose["iter"] == again["iter"] and torch.equal(loose["x"], again["x"])
    ctl["eps_abs"] = ctl["eps_rel"] = 1e-6                          # the same dict, mutated in place
    tight = L.torch_solve_box_qp(*inp, ctl)
    assert tight["iter"] > loose["iter"]
    ref = O.solve_box_qp(*[t.cpu() for t in inp], O.make_control(eps_abs=1e-6, eps_rel=1e-6))
    assert tight["iter"] == ref["iter"] and err(tight["x"], ref["x"]) < 2e-5
    ctl["max_iters"] = 3
    assert L.torch_solve_box_qp(*inp, ctl)["iter"] == 2
    rho_t = torch.full((4, 1, 1), 0.7, device=dev)                 # per-problem rho: resolved afresh every call
    a = L.torch_solve_box_qp(*inp, dict(ctl, max_iters=200, rho=rho_t))
    rho_t.fill_(5.0)
    b = L.torch_solve_box_qp(*inp, dict(ctl, max_iters=200, rho=rho_t))
    assert not torch.equal(a["x"], b["x"]) or a["iter"] != b["iter"]


def test_pipelined_calls_verify_the_bound_assumption_late(dev):
    """control['sync'] = False cannot repeat a solve: a batch whose 'any finite bound?' differs from what the previous
    solve with the same control saw is reported late; the next call then runs the right schedule."""
    n, B = 40, 5
    Q, p, A, b, lb, ub = (t.to(dev) for t in O.create_qp_data(n, B, seed=9))
    inf = float("inf")
    lb_none, ub_none = torch.full_like(lb, -inf), torch.full_like(ub, inf)
    ref = O.solve_box_qp(*(t.cpu() for t in (Q, p, A, b, lb, ub)), O.make_control(**TOL))
    ref0 = O.solve_box_qp(Q.cpu(), p.cpu(), A.cpu(), b.cpu(), lb_none.cpu(), ub_none.cpu(), O.make_control(**TOL))
    ctl = L.box_qp_control(sync=False, **TOL)
    layer = L.SolveBoxQP(control=ctl)
    for _ in range(3):                                   # fresh clones every call: nothing to remember, nothing to wait for
        x = layer(Q, p, A, b, lb.clone(), ub.clone())
    L.synchronize()
    assert err(x, ref["x"]) < 2e-5
    layer(Q, p, A, b, lb_none, ub_none)                  # enqueued for the ADMM loop; the device finds no bound
    with pytest.raises(RuntimeError, match="finite bound"):
        L.synchronize()
    assert ctl["rho"] == 0                               # the reference's side effect, applied when the status arrived
    x0 = layer(Q, p, A, b, lb_none, ub_none)             # now assumed right
    L.synchronize()
    assert err(x0, ref0["x"]) < 2e-5
    L.synchronize()


def test_two_workgroup_schedules_beside_a_foreign_kernel(dev):
    """B = 128, n = 500: the register-resident sweep and the two-workgroup loop hold 256 workgroups that wait for each
    other inside a launch.  A kernel of another stream that sits on 48 CUs for 8 ms (its workgroups take the whole LDS of
    a CU) delays some of them; the schedules must come through with the SAME bits and no timeout."""
    lib = _lib.load()
    B, n = 128, 500
    inp = [t.to(dev) for t in O.create_qp_data(n, B, seed=1)]
    ctl = L.box_qp_control(**TOL)
    ref = L.torch_solve_box_qp(*inp, dict(ctl))
    assert ref["_stats"]["loop_workgroups"] == 2 and ref["_stats"]["factor_launches"] == 3
    side = torch.cuda.Stream(device=dev)
    spin = lambda: _lib.check(lib.lqp_debug_spin(_lib.stream_ptr(dev), 48, 8000, 150 * 1024), "debug_spin")
    # (1) the foreign kernel is there first: 48 of the 256 workgroups of every two-workgroup launch start 8 ms late
    torch.cuda.synchronize(dev)
    with torch.cuda.stream(side):
        spin()
    sol = L.torch_solve_box_qp(*inp, dict(ctl))
    torch.cuda.synchronize(dev)
    assert sol["iter"] == ref["iter"]
    for k in ("x", "z", "u", "lams", "nus"):
        assert torch.equal(sol[k], ref[k]), k
    # (2) it arrives while the (pipelined) solve is in flight
    layer = L.SolveBoxQP(control=L.box_qp_control(sync=False, **TOL))
    x0 = layer(*inp)
    L.synchronize()
    for _ in range(3):
        x1 = layer(*inp)
        with torch.cuda.stream(side):
            spin()
        x2 = layer(*inp)
        torch.cuda.synchronize(dev)
        L.synchronize()
        assert torch.equal(x1, x0) and torch.equal(x2, x0) and torch.equal(x0, ref["x"])


@pytest.mark.parametrize("n,m,B", [(1000, 1, 3), (700, 2, 2), (1024, 0, 2)])
def test_cholesky_backward_above_512(dev, n, m, B):
    """The Cholesky form of the backward for free sets above 512 variables (config-4 size: the reference's own layer demo
    differentiates at n_x = 1000, demo/demo_solve_box_qp_torch_layer.py:25-40): wg_chol_factor_big (panel in chunks,
    trailing updates by group pairs) + block solves over up to 16 block columns.  All six gradients against the fp64
    oracle at the GPU's iteration count, and against the LU form of the same system."""
    torch.manual_seed(n + m)
    Q, p, _, _, lb, ub = O.create_qp_data(n, B, seed=n + 3, with_eq=False)
    A = torch.randn(B, m, n) if m else None
    b = 0.1 * torch.randn(B, m, 1) if m else None
    inp = (Q, p, A, b, lb, ub)
    ctl = O.make_control(**TOL)
    sol, a = solve(dev, inp, ctl)
    assert sol["_stats"]["linsolve_used"] == 2
    cot = torch.randn(B, n, 1)
    want = dict(dQ=True, dp=True, dA=m > 0, db=m > 0, dlb=True, dub=True)
    out = {}
    for ls in (1, 2):
        _lib.profile(enable=True, reset=True)
        out[ls] = SB._fp_backward(cot.to(dev), sol["x"], sol["u"], sol["lams"], sol["nus"], a[0], a[2], a[4], a[5], sol["rho"],
                                 want, sync=True, linsolve=ls)
        used = _lib.profile(); _lib.profile(enable=False)
        assert (used["bwd_cholesky"][1] == 1) == (ls == 2), used
    # the reference's arithmetic on the same inputs (fp32 oracle at the GPU's iteration count) and the fp64 truth: within
    # rtol 1e-4 of the former, or no further from the latter than the former is (fp32 forward iterates differ from the
    # fp64 ones at the 1e-5 level, which the gradients inherit)
    pinned = dict(ctl, eps_abs=1e-12, eps_rel=1e-12, max_iters=sol["iter"] + 1)
    s32 = O.solve_box_qp(*inp, dict(pinned))
    g32 = O.solve_box_qp_grad(cot, s32["x"], s32["u"], s32["lams"], s32["nus"], inp[0], inp[2], inp[4], inp[5], s32["rho"])
    d = [None if t is None else t.double() for t in inp]
    s64 = O.solve_box_qp(*d, dict(pinned))
    g64 = O.solve_box_qp_grad(cot.double(), s64["x"], s64["u"], s64["lams"], s64["nus"], d[0], d[2], d[4], d[5], s64["rho"])
    for idx, nm in enumerate(GRADS):
        if out[2][idx] is None:
            continue
        close_or_fp64(f"chol_backward_big_n{n}", nm, out[2][idx], g32[idx], g64[idx], G_RTOL, lu_form_vs_fp64=err(out[1][idx], g64[idx]))


@pytest.mark.parametrize("n,B,m", [(500, 32, 1), (448, 3, 2), (512, 5, 0)])
def test_four_workgroups_per_qp_loop(dev, monkeypatch, n, B, m):
    """Batches up to a quarter of the CUs share every product between FOUR workgroups (one column pair each).  Same
    iterates as the two-workgroup loop up to the order of the four partial sums: same iteration count, x within rounding,
    and both against the CPU oracle."""
    torch.manual_seed(n + B)
    Q, p, _, _, lb, ub = O.create_qp_data(n, B, seed=n + B, with_eq=False)
    A = torch.randn(B, m, n) if m else None
    b = 0.1 * torch.randn(B, m, 1) if m else None
    ctl = O.make_control(**TOL)
    sols = {}
    for np4 in ("1", "0"):
        monkeypatch.setenv("LQP_LOOP_SPLIT4", np4)
        sols[np4], _ = solve(dev, (Q, p, A, b, lb, ub), ctl)
    assert sols["1"]["_stats"]["loop_workgroups"] == 4 and sols["0"]["_stats"]["loop_workgroups"] == 2
    assert sols["1"]["iter"] == sols["0"]["iter"]
    ref = O.solve_box_qp(Q, p, A, b, lb, ub, ctl)
    assert sols["1"]["iter"] == ref["iter"]
    for k in ("x", "z", "u", "lams") + (("nus",) if m else ()):
        P.record("loop_np4", k, err(sols["1"][k], ref[k]), 1.0, n=n, B=B, m=m)
        assert err(sols["1"][k], sols["0"][k]) < 1e-5 * max(1.0, float(ref[k].abs().max())), k
        assert err(sols["1"][k], ref[k]) < 2e-5 * max(1.0, float(ref[k].abs().max())), k


@pytest.mark.gpu
@pytest.mark.parametrize("n,B,m,scale", [(500, 8, 1, True), (448, 3, 0, True), (512, 5, 2, False), (449, 2, 0, True)])
def test_four_workgroups_per_qp_sweep(dev, monkeypatch, n, B, m, scale):
    """Batches up to a quarter of the CUs share the register-resident factorisation between FOUR workgroups per matrix (one
    column pair each; the halves of ||Qs||_F -- four of them -- travel in the step-0 flag granules).  Same tile arithmetic
    as with two: same iteration count, x within rounding of the two-workgroup sweep, and both against the CPU oracle."""
    torch.manual_seed(7 * n + B)
    Q, p, _, _, lb, ub = O.create_qp_data(n, B, seed=3 * n + B, with_eq=False)
    A = torch.randn(B, m, n) if m else None
    b = 0.1 * torch.randn(B, m, 1) if m else None
    ctl = O.make_control(**TOL)
    ctl["scale"] = scale
    sols = {}
    for np4 in ("1", "0"):
        monkeypatch.setenv("LQP_SPD_RESIDENT4", np4)
        sols[np4], _ = solve(dev, (Q, p, A, b, lb, ub), ctl)
        assert sols[np4]["_stats"]["linsolve_used"] == 2 and sols[np4]["_stats"]["factor_launches"] == 3
    assert sols["1"]["iter"] == sols["0"]["iter"]
    ref = O.solve_box_qp(Q, p, A, b, lb, ub, ctl)
    assert sols["1"]["iter"] == ref["iter"]
    for k in ("x", "z", "u", "lams") + (("nus",) if m else ()):
        P.record("sweep_np4", k, err(sols["1"][k], ref[k]), 1.0, n=n, B=B, m=m)
        assert err(sols["1"][k], sols["0"][k]) < 1e-5 * max(1.0, float(ref[k].abs().max())), k
        assert err(sols["1"][k], ref[k]) < 2e-5 * max(1.0, float(ref[k].abs().max())), k


# ---------------------------------------------------------------- round 4: hardening of the new tiers
@pytest.mark.parametrize("n,B,m", [(30, 5, 0), (64, 4, 2), (100, 16, 1), (128, 3, 3)])
def test_small_loop_kernel_against_the_general_one(dev, monkeypatch, n, B, m):
    """k_admm_loop_small (n <= 128: the whole matrix in the registers of 256 threads) against the 1024-thread loop kernel
    it stands in for (LQP_LOOP_SMALL=0) and against the oracle.  At a pinned iteration count: iterates within rounding of
    each other and within the bar of the oracle; with the stopping rule live: the same stop up to one check interval (the
    check is a floating-point threshold: a borderline problem may stop one check later on one side)."""
    Q, p, _, _, lb, ub = O.create_qp_data(n, B, seed=n + m, with_eq=False)
    g = torch.Generator().manual_seed(n)
    A = torch.randn(B, m, n, generator=g) if m else None
    b = 0.1 * torch.randn(B, m, 1, generator=g) if m else None
    pinned = dict(eps_abs=1e-12, eps_rel=1e-12, max_iters=41)
    out, live = {}, {}
    for flag in ("1", "0"):
        monkeypatch.setenv("LQP_LOOP_SMALL", flag)
        out[flag], _ = solve(dev, (Q, p, A, b, lb, ub), O.make_control(linsolve="spd", **pinned))
        live[flag], _ = solve(dev, (Q, p, A, b, lb, ub), O.make_control(linsolve="spd", **TOL))
        assert out[flag]["_stats"]["linsolve_used"] == 2 and out[flag]["_stats"]["mode_used"] == 2 and out[flag]["iter"] == 40
    ref = O.solve_box_qp(Q, p, A, b, lb, ub, O.make_control(**pinned))
    t64 = O.solve_box_qp(*[None if t is None else t.double() for t in (Q, p, A, b, lb, ub)], O.make_control(**pinned))
    for k in ("x", "z", "u", "lams") + (("nus",) if m else ()):
        assert err(out["1"][k], out["0"][k]) < 5e-6, k
        close_or_fp64(f"small_loop_n{n}_m{m}", k, out["1"][k], ref[k], t64[k], X_TOL)
    ref_live = O.solve_box_qp(Q, p, A, b, lb, ub, O.make_control(**TOL))
    interval = O.resolve_control(O.make_control(**TOL), n).check_solved
    assert abs(live["1"]["iter"] - live["0"]["iter"]) <= interval and abs(live["1"]["iter"] - ref_live["iter"]) <= interval
    assert err(live["1"]["x"], ref_live["x"]) < 1e-3 * max(1.0, float(ref_live["x"].abs().max()))


@pytest.mark.parametrize("n,B,m", [(60, 3, 0), (96, 4, 3), (200, 2, 1)])
def test_unroll_native_against_the_taped_loop(dev, monkeypatch, n, B, m):
    """The reverse sweep (lqp_boxqp_unroll_backward) against the taped loop it replaces, box-only and with several equality
    rows: the same solution, all gradients within rtol 1e-4 of scale, and the float64 tape as arbiter."""
    Q, p, _, _, lb, ub = O.create_qp_data(n, B, seed=n + 7 * m, with_eq=False)
    gen = torch.Generator().manual_seed(n + m)
    A = torch.randn(B, m, n, generator=gen) if m else None
    b = 0.1 * torch.randn(B, m, 1, generator=gen) if m else None
    cot = torch.randn(B, n, 1, generator=gen)
    names = ("Q", "p", "A", "b", "lb", "ub")
    data = dict(zip(names, (Q, p, A, b, lb, ub)))
    grads = {}
    for native in ("1", "0"):
        monkeypatch.setenv("LQP_UNROLL_NATIVE", native)
        leaves = [None if data[k] is None else data[k].to(dev).requires_grad_(True) for k in names]
        x = L.SolveBoxQP(control=L.box_qp_control(unroll=True, **TOL))(*leaves)
        x.backward(cot.to(dev))
        grads[native] = (x.detach(), [None if t is None else t.grad for t in leaves])
    from lqp_py_amd.unrolled import _eager_unrolled
    l64 = [None if data[k] is None else data[k].double().requires_grad_(True) for k in names]
    x64 = _eager_unrolled(*l64, SB.resolve_control(L.box_qp_control(unroll=True, **TOL), n), True, True, solver_cls=_CpuLU)
    x64.backward(cot.double())
    close_or_fp64(f"unroll_vs_tape_n{n}_m{m}", "x", grads["1"][0], grads["0"][0], x64.detach(), X_TOL)
    for nm, g1, g0, t64 in zip(GRADS, grads["1"][1], grads["0"][1], l64):
        if g1 is None:
            assert g0 is None
            continue
        close_or_fp64(f"unroll_vs_tape_n{n}_m{m}", nm, g1, g0, t64.grad, G_RTOL)


@pytest.mark.parametrize("n,B,m,dtype,kind", [(60, 3, 0, torch.float64, "auto"), (96, 4, 3, torch.float64, "auto"), (120, 2, 20, torch.float32, "auto"),
                                              (200, 2, 1, torch.float32, "lu"), (150, 2, 2, torch.float32, "nonsym"), (250, 4, 16, torch.float64, "auto"),
                                              (1100, 1, 2, torch.float32, "auto")])
def test_unroll_on_the_lu_tape_against_the_taped_loop(dev, monkeypatch, n, B, m, dtype, kind):
    """unroll=True where the x-update is the pivoted LU -- float64, more than 16 equality rows, linsolve='lu', a non-symmetric Q, more
    than 1024 rows: the reverse sweep on the packed factor (lqp_boxqp_unroll_backward_lu; the tape's node is TorchLULayer,
    lqp_py/lu_layer.py:25-58) against the taped loop of torch ops it replaces (lqp_py/solve_box_qp_admm_torch.py:235-313 under
    unroll): the same solution, all gradients within rtol 1e-4 of scale in float32 (the float64 tape on the host as arbiter), 1e-8
    in float64."""
    Q, p, _, _, lb, ub = O.create_qp_data(n, B, seed=n + 7 * m, with_eq=False)
    gen = torch.Generator().manual_seed(n + m)
    if kind == "nonsym":
        Q = Q + 0.02 * torch.triu(torch.randn(B, n, n, generator=gen), 1)
    A = torch.randn(B, m, n, generator=gen) if m else None
    b = 0.1 * torch.randn(B, m, 1, generator=gen) if m else None
    cot = torch.randn(B, n, 1, generator=gen)
    names = ("Q", "p", "A", "b", "lb", "ub")
    data = dict(zip(names, (Q, p, A, b, lb, ub)))
    ctl = L.box_qp_control(unroll=True, **TOL)
    if kind == "lu":
        ctl["linsolve"] = "lu"
    grads = {}
    for native in ("1", "0"):
        monkeypatch.setenv("LQP_UNROLL_NATIVE", native)
        leaves = [None if data[k] is None else data[k].to(dtype).to(dev).requires_grad_(True) for k in names]
        _lib.profile(enable=True, reset=True)
        x = L.SolveBoxQP(control=dict(ctl))(*leaves)
        x.backward(cot.to(dtype).to(dev))
        used = _lib.profile(); _lib.profile(enable=False)
        assert used["unroll_backward"][1] == ((3 if m else 2) if native == "1" else 0), (native, used["unroll_backward"])
        grads[native] = (x.detach(), [None if t is None else t.grad for t in leaves])
    case = f"unroll_lu_tape_n{n}_m{m}_{kind}_{'f32' if dtype == torch.float32 else 'f64'}"
    if dtype == torch.float64:
        assert err(grads["1"][0], grads["0"][0]) < 1e-9
        for nm, g1, g0 in zip(GRADS, grads["1"][1], grads["0"][1]):
            if g1 is None:
                assert g0 is None
                continue
            e = err(g1, g0) / max(1.0, float(g0.abs().max()))
            P.record(case, nm, e)
            assert e < 1e-8, (nm, e)
        return
    from lqp_py_amd.unrolled import _eager_unrolled
    l64 = [None if data[k] is None else data[k].double().requires_grad_(True) for k in names]
    x64 = _eager_unrolled(*l64, SB.resolve_control(dict(ctl), n), True, True, solver_cls=_CpuLU)
    x64.backward(cot.double())
    close_or_fp64(case, "x", grads["1"][0], grads["0"][0], x64.detach(), X_TOL)
    for nm, g1, g0, t64 in zip(GRADS, grads["1"][1], grads["0"][1], l64):
        if g1 is None:
            assert g0 is None
            continue
        close_or_fp64(case, nm, g1, g0, t64.grad, G_RTOL)


@pytest.mark.parametrize("n,B,m,dtype", [(1100, 2, 0, torch.float32), (1030, 2, 20, torch.float32), (1200, 1, 2, torch.float64)])
def test_lu_tier_above_1024_rows(dev, n, B, m, dtype):
    """n + m > 1024 with no / many equality rows and in float64: forward against the oracle (iteration count, iterates), the
    fixed-point gradients against the oracle's for the same solution."""
    Q, p, _, _, lb, ub = O.create_qp_data(n, B, seed=n, with_eq=False, dtype=dtype)
    gen = torch.Generator().manual_seed(n)
    A = torch.randn(B, m, n, generator=gen, dtype=dtype) if m else None
    b = (0.1 * torch.randn(B, m, 1, generator=gen, dtype=dtype)) if m else None
    inp = (Q, p, A, b, lb, ub)
    sol, a = solve(dev, inp, O.make_control(**TOL))
    ref = O.solve_box_qp(*inp, O.make_control(**TOL))
    assert sol["_stats"]["linsolve_used"] == 1 and sol["iter"] == ref["iter"]
    t64 = ref if dtype == torch.float64 else O.solve_box_qp(*[None if t is None else t.double() for t in inp],
                                                          O.make_control(eps_abs=1e-12, eps_rel=1e-12, max_iters=ref["iter"] + 1))
    tol = X_TOL if dtype == torch.float32 else 1e-9
    for k in ("x", "u", "lams") + (("nus",) if m else ()):
        close_or_fp64(f"lu_tier_n{n}_m{m}", k, sol[k], ref[k], t64[k], tol)
    cot = torch.randn(B, n, 1, generator=gen, dtype=dtype)
    gr = L.torch_solve_box_qp_grad(cot.to(dev), sol["x"], sol["u"], sol["lams"], sol["nus"], a[0], a[2], a[4], a[5], sol["rho"])
    gref = O.solve_box_qp_grad(cot, ref["x"], ref["u"], ref["lams"], ref["nus"], Q, A, lb, ub, ref["rho"])
    d = [None if t is None else t.double() for t in inp]
    g64 = O.solve_box_qp_grad(cot.double(), t64["x"], t64["u"], t64["lams"], t64["nus"], d[0], d[2], d[4], d[5], t64["rho"])
    for idx, nm in enumerate(GRADS):
        if gr[idx] is None:
            continue
        close_or_fp64(f"lu_tier_n{n}_m{m}", nm, gr[idx], gref[idx], g64[idx], G_RTOL if dtype == torch.float32 else 1e-8)


def test_kkt_backward_native_in_float64(dev, monkeypatch):
    """backward='kkt' in float64 (the reference itself fails there, SURVEY 8c: a dtype bug at :447): native (LU form of the
    reduced system: float64 has no Cholesky form) against the composed path."""
    g = load_golden("g12_kkt_backward")
    out = {}
    for native in (True, False):
        monkeypatch.setattr(SB, "_KKT_NATIVE", native)
        leaves = [g[k].double().to(dev).requires_grad_(True) for k in ("Q", "p", "A", "b", "lb", "ub")]
        x = L.SolveBoxQP(control=L.box_qp_control(backward='kkt', **TOL))(*leaves)
        x.backward(g["cot"].double().to(dev))
        out[native] = [t.grad for t in leaves]
    for nm, a_, b_ in zip(GRADS, out[True], out[False]):
        assert err(a_, b_) < 1e-8 * max(1.0, float(b_.abs().max())), nm
        assert err(a_, g[nm]) < 1e-4 * max(1.0, float(g[nm].abs().max())), nm
