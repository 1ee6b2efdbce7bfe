import os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import lqp_py_amd as L
from lqp_py_amd.synthetic import create_qp_data
dev = torch.device("cuda:0")
B, n, m = 128, int(os.environ.get("N", 100)), int(os.environ.get("M", 3))
dtype = torch.float64
Q, p, _, _, lb, ub = create_qp_data(n, B, seed=3, with_eq=False)
g = torch.Generator().manual_seed(4)
A = torch.randn(B, m, n, generator=g) if m else None
b = (A @ (0.5 * (lb + ub))) if m else None
inp = [None if t is None else t.to(dtype).to(dev) for t in (Q, p, A, b, lb, ub)]
layer = L.SolveBoxQP(control=dict(L.box_qp_control(eps_abs=1e-5, eps_rel=1e-5), sync=False))
cot = torch.ones(B, n, 1, dtype=dtype, device=dev)
def step():
    Qg = inp[0].detach().requires_grad_(True); pg = inp[1].detach().requires_grad_(True)
    layer(Qg, pg, *inp[2:]).backward(cot)
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize(); L.synchronize()
print("ms/step", (time.perf_counter() - t0) / 20 * 1e3, L.solve_box_qp_admm_torch.last_forward_status(dev))
import cProfile, pstats, io
pr = cProfile.Profile(); pr.enable()
for _ in range(50): step()
pr.disable(); torch.cuda.synchronize(); L.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue()[:3500])
