import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import boxqp_oracle as O
from lqp_py_amd import lu_layer
dev = torch.device("cuda:0")
B, n = 8, 500
Q, p, A, b, lb, ub = O.create_qp_data(n, B, seed=0)
sol = O.solve_box_qp(Q, p, A, b, lb, ub, O.make_control(eps_abs=1e-5, eps_rel=1e-5))
x, u = sol["x"], sol["u"]
w = x + u
keep = ~((w > ub) | (w < lb))
mats, rhss = [], []
for i in range(B):
    f = keep[i, :, 0].nonzero()[:, 0]
    nf = len(f)
    M = torch.zeros(nf + 1, nf + 1)
    M[:nf, :nf] = Q[i][f][:, f]
    M[:nf, nf] = A[i, 0, f]; M[nf, :nf] = A[i, 0, f]
    M += 1e-8 * torch.eye(nf + 1)
    r = torch.zeros(nf + 1, 1); r[:nf, 0] = -1.0
    mats.append(M); rhss.append(r)
for i in range(2):
    M, r = mats[i], rhss[i]
    x64 = torch.linalg.solve(M.double(), r.double())
    LU, P = torch.linalg.lu_factor(M.unsqueeze(0))
    xt = torch.linalg.lu_solve(LU, P, r.unsqueeze(0))[0]
    xh = lu_layer.lu_solve(LU.to(dev), P.to(dev), r.unsqueeze(0).to(dev))[0].cpu()
    LUh, Ph = lu_layer.lu_factor(M.unsqueeze(0).to(dev))
    xhh = lu_layer.lu_solve(LUh, Ph, r.unsqueeze(0).to(dev))[0].cpu()
    xth = torch.linalg.lu_solve(LUh.cpu(), Ph.cpu(), r.unsqueeze(0))[0]
    e = lambda a: float((a.double() - x64).abs().max())
    print(f"nf={M.shape[0]-1} cond={float(torch.linalg.cond(M.double())):.2e}  torch/torch {e(xt):.2e}  torchLU/hipsolve {e(xh):.2e}  hipLU/hipsolve {e(xhh):.2e}  hipLU/torchsolve {e(xth):.2e}  pivots equal {torch.equal(P, Ph.cpu())}")
