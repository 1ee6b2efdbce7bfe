#!/usr/bin/env python3
"""In-kernel cycle breakdown of the register-resident sweep k_spd_resident (thread 0 of workgroup 0 of every QP)."""
import os, sys
os.environ.setdefault("LQP_ENV_NOCACHE", "1")
WAVE = int(os.environ.get("LQP_DBG_WAVE", "0"))      # the wave of workgroup 0 whose stamps are recorded (4 .. 7: a staging wave)
os.environ["LQP_DBG_QPASS"] = str(1 + (WAVE << 8))      # (the stamps of the pass over Q go to a second half of the debug buffer)      # (this tool flips library knobs between solves)
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import lqp_py_amd as L
from lqp_py_amd import _lib
from lqp_py_amd.synthetic import create_qp_data
dev = torch.device("cuda:0")
lib = _lib.load()
B, n = 128, 500
inp = [t.to(dev) for t in create_qp_data(n, B, seed=0)]
ctl = L.box_qp_control(eps_abs=1e-5, eps_rel=1e-5, max_iters=1)
os.environ["LQP_LOOP_SPLIT"] = "0"          # (its debug build would add its own counters to the same words)
dbg = torch.zeros(2 * B * 8, dtype=torch.int64, device=dev)      # (second half: the stamps of the pass over Q)
L.torch_solve_box_qp(*inp, dict(ctl))
lib.lqp_debug_set_lu_counters(_lib.ptr(dbg))
L.torch_solve_box_qp(*inp, dict(ctl))
torch.cuda.synchronize()
lib.lqp_debug_set_lu_counters(None)
c = dbg[:B * 8].view(B, 8).double()
q = dbg[B * 8:].view(B, 8).double().mean(0).tolist()
if os.environ.get("LQP_RS_FORM", "2") == "4":
    # the look-ahead sweep (wg_spd_sweep_resident_v4) built with -DLQP_RS4_STAMPS=1 (wave 0, a chain wave) or 2 (wave 4, a staging wave)
    names = ["Y = P W^T + barrier", "tiles of rows/columns k, k+1 + publish", "the other tiles", "waits", "pivot block k+1 / staging of panel k+1",
             "closing barrier"]
    tot = c[:, :6].sum(1).mean()
    for i, nm in enumerate(names):
        print("%-40s %9.0f cycles (%.0f per step)" % (nm, c[:, i].mean(), c[:, i].mean() / 8))
    print("total of the 8 steps %.0f cycles" % tot)
else:
    names = ["publish + wait for the partner", "pivot block + staging", "W -> halves, panel cells", "Y = P W^T", "tile updates", "publish of the next step"]
    tot = c[:, :6].sum(1).mean()
    for i, nm in enumerate(names):
        print("%-32s %9.0f cycles  (%4.1f %%)" % (nm, c[:, i].mean(), 100 * c[:, i].mean() / tot))
    print("total of the 8 steps %.0f cycles" % tot)
    if WAVE:
        print("wave %d: its own staging work inside the pivot phase %.0f cycles" % (WAVE, c[:, 6].mean()))
    print("pass over Q inside the sweep: tiles + mirrors %.0f cycles, up to the scaling vector %.0f" % (c[:, 7].mean(), c[:, 6].mean()))
    print("  cumulative: tiles + mirrors %.0f | maxima stored and acknowledged %.0f | partner announced %.0f | partner's maxima read %.0f | scaling vector %.0f" % tuple(q[:5]))
