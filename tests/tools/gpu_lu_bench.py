#!/usr/bin/env python3
"""LU kernel timing + in-kernel phase breakdown (debug counters) on KKT-like matrices."""
import os, sys, time
os.environ.setdefault("LQP_ENV_NOCACHE", "1")      # (this tool flips library knobs between solves)
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from lqp_py_amd import _lib, lu_layer
from oracle import boxqp_oracle as O

dev = torch.device("cuda:0")
lib = _lib.load()
B, n = 128, int(os.environ.get("N_X", "500"))
Q, p, A, b, lb, ub = O.create_qp_data(n, B, seed=0)
M = O.kkt_matrix(Q + 1.2 * torch.eye(n), A).to(dev)
N = M.shape[1]
dbg = torch.zeros(B * 4, dtype=torch.int64, device=dev)
for (nt, pb, mfma, la) in ((512, 32, 1, 0), (512, 16, 1, 0), (1024, 16, 1, 0)):
    if True:
        os.environ["LQP_LU2"] = "0"            # (the one-workgroup kernel; csrc/lqp_lu2.hpp has its own tool: gpu_lu2_check.py)
        os.environ["LQP_LU_NT"] = str(nt)
        os.environ["LQP_LU_PB"] = str(pb)
        os.environ["LQP_LU_MFMA"] = str(mfma)
        lib.lqp_debug_set_lu_counters(None)
        LU, P = lu_layer.lu_factor(M)
        torch.cuda.synchronize()
        _lib.profile(enable=True, reset=True)
        for _ in range(5):
            LU, P = lu_layer.lu_factor(M)
        torch.cuda.synchronize()
        pr = _lib.profile()
        _lib.profile(enable=False)
        ms = pr["lu_factor"][0] / pr["lu_factor"][1]
        lib.lqp_debug_set_lu_counters(_lib.ptr(dbg))
        LU, P = lu_layer.lu_factor(M)
        torch.cuda.synchronize()
        lib.lqp_debug_set_lu_counters(None)
        c = dbg.view(B, 4).double().mean(0).tolist()
        Pm, Lm, Um = torch.lu_unpack(LU.cpu(), P.cpu())
        err = float((Pm @ Lm @ Um - M.cpu()).abs().max())
        print(f"N={N} LA={la} NT={nt} PB={pb} mfma={mfma}: {ms*1e3:8.1f} us/launch | cycles panel(P) {c[0]:.0f} swaps+U12(serial) {c[1]:.0f} "
              f"trailing {c[2]:.0f} total {c[3]:.0f} | recon err {err:.2e}", flush=True)
