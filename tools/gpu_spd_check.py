#!/usr/bin/env python3
"""SPD inverse kernel vs torch (f64), and the symmetric-inverse forward path vs the LU path."""
import os, sys, ctypes, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from lqp_py_amd import _lib
import lqp_py_amd.solve_box_qp_admm_torch as L
from lqp_py_amd.synthetic import create_qp_data
from lqp_py_amd.control import box_qp_control

dev = torch.device("cuda:0")
lib = _lib.load()
for n, B in ((10, 3), (64, 2), (65, 2), (100, 4), (300, 3), (500, 8), (512, 4)):
    torch.manual_seed(n)
    Lm = torch.randn(B, 2 * n, n)
    K = (Lm.transpose(1, 2) @ Lm / (2 * n) + 1.2 * torch.eye(n)).to(dev)
    out = torch.empty_like(K)
    info = torch.full((B,), -1, dtype=torch.int32, device=dev)
    nb = lib.lqp_spd_inverse_workspace_bytes(0, B, n)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    st = lib.lqp_spd_inverse_batched(_lib.stream_ptr(dev), 0, B, n, _lib.ptr(K), _lib.ptr(out), _lib.ptr(info), _lib.ptr(ws), nb)
    torch.cuda.synchronize()
    ref = torch.linalg.inv(K.double().cpu())
    err = float((out.cpu().double() - ref).abs().max())
    print(f"spd_inverse n={n} B={B}: status {st} info {info.tolist()} max err {err:.2e} (|inv| max {float(ref.abs().max()):.2f})", flush=True)

TOL = dict(eps_abs=1e-5, eps_rel=1e-5)
for (n, B, eq) in ((10, 32, False), (50, 8, True), (100, 128, False), (500, 128, True)):
    inp = create_qp_data(n, B, seed=0, with_eq=eq)
    a = [None if t is None else t.to(dev) for t in inp]
    sols = {}
    for ls in ("lu", "spd"):
        sols[ls] = L.torch_solve_box_qp(*a, box_qp_control(linsolve=ls, **TOL))
    s0, s1 = sols["lu"], sols["spd"]
    d = {k: float((s0[k] - s1[k]).abs().max()) for k in ("x", "z", "u", "lams") }
    if eq: d["nus"] = float((s0["nus"] - s1["nus"]).abs().max())
    print(f"forward n={n} B={B} eq={eq}: iters lu {s0['iter']} spd {s1['iter']} | diffs {d}", flush=True)
