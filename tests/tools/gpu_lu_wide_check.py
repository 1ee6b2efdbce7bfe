#!/usr/bin/env python3
"""Wide LU (csrc/lqp_lu_wide.hpp) against the one-workgroup kernel above 1024 rows: bits, pivots, time.  Usage: [f32|f64] [B] [N ...]"""
import os, sys
os.environ.setdefault("LQP_ENV_NOCACHE", "1")
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from lqp_py_amd import _lib, lu_layer
dev = torch.device("cuda:0")
dt = torch.float64 if (len(sys.argv) > 1 and sys.argv[1] == "f64") else torch.float32
args = [a for a in sys.argv[1:] if a not in ("f32", "f64")]
B = int(args[0]) if args else 8
sizes = [int(a) for a in args[1:]] or [1100, 1500, 2048]
def timed(M, reps=3):
    lu_layer.lu_factor(M); torch.cuda.synchronize()
    _lib.profile(enable=True, reset=True)
    for _ in range(reps):
        LU, P = lu_layer.lu_factor(M)
    torch.cuda.synchronize()
    pr = _lib.profile(); _lib.profile(enable=False)
    return LU, P, pr["lu_factor"][0] / pr["lu_factor"][1]
for N in sizes:
    torch.manual_seed(N)
    M = torch.randn(B, N, N, dtype=dt).to(dev)
    os.environ["LQP_LU_WIDE"] = "0"; LU0, P0, ms0 = timed(M)
    os.environ["LQP_LU_WIDE"] = "1"; LU1, P1, ms1 = timed(M)        # (LQP_LU_WIDE_MIN=512: also between 512 and 1024 rows)
    lib = _lib.load()
    dbg = torch.zeros(B * 16, dtype=torch.int64, device=dev)
    lib.lqp_debug_set_lu_counters(_lib.ptr(dbg))
    lu_layer.lu_factor(M); torch.cuda.synchronize()
    lib.lqp_debug_set_lu_counters(None)
    c = dbg.view(B, 16)[:, :8].double().mean(0).tolist()
    names = ["panel", "slot-wait", "publish", "msg-wait", "msg-copy", "swap+U12", "trailing", "total"]
    print("   workgroup 1, k cycles: " + "  ".join(f"{n} {v/1e3:.0f}" for n, v in zip(names, c)))
    same_p = bool(torch.equal(P0, P1)); d = float((LU0 - LU1).abs().max())
    nbad = int((LU0 != LU1).sum())
    print(f"N={N} B={B}: one-wg {ms0:.2f} ms  wide {ms1:.2f} ms | pivots equal {same_p}  max|diff| {d:.2e}  differing entries {nbad}", flush=True)
    if nbad:
        bad = (LU0 != LU1).nonzero()
        print("  first differing (b, i, j):", bad[:5].tolist(), " min col", int(bad[:, 2].min()), " min row", int(bad[:, 1].min()))
