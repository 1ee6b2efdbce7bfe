#!/usr/bin/env python3
"""Learning-p training loop on the GPU layer (the reference's experiments/experiment_2.py:57-99):
Linear(5 -> n_x) -> SolveBoxQP -> QP loss -> SGD, minibatch 32, only p requires grad (so dQ is never
formed: the layer honours needs_input_grad).

    python examples/experiment_2.py [--n 500] [--epochs 100] [--batch 128] [--mini 32]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lqp_py_amd as L                      # noqa: E402


def make_data(n_x, n_batch, seed, dev):
    """Same draw order as the reference's create_qp_data (experiments/utils.py:41-61)."""
    torch.manual_seed(seed)
    Lm = torch.randn(n_batch, 2 * n_x, n_x)
    Q = torch.matmul(Lm.transpose(1, 2), Lm) / (2 * n_x)
    _ = torch.randn(n_batch, n_x, 1)
    A, b = torch.ones(n_batch, 1, n_x), torch.ones(n_batch, 1, 1)
    lb = -(torch.rand(n_batch, n_x, 1) + 1)
    ub = torch.rand(n_batch, n_x, 1) + 1
    return [t.to(dev) for t in (Q, A, b, lb, ub)]


def train(n_x=500, n_batch=128, n_mini=32, n_epochs=100, n_features=5, lr=1e-3, tol=1e-5, seed=0, dev=None,
          layer=None, verbose=True):
    dev = dev or torch.device("cuda:0")
    Q, A, b, lb, ub = make_data(n_x, n_batch, seed, dev)
    x = torch.normal(mean=0, std=1, size=(n_batch, n_features)).to(dev)
    beta = torch.normal(mean=0, std=1, size=(n_features, n_x)).to(dev)
    p = torch.matmul(x, beta).unsqueeze(2)
    torch.manual_seed(seed + 1)
    model = torch.nn.Linear(n_features, n_x).to(dev)
    opt = torch.optim.SGD(model.parameters(), lr=lr)
    layer = layer or L.SolveBoxQP(control=L.box_qp_control(eps_rel=tol, eps_abs=tol, verbose=False, reduce='max'))
    rs = np.random.RandomState(seed)
    losses, t_fwd, t_bwd = [], 0.0, 0.0
    for epoch in range(n_epochs):
        idx = torch.as_tensor(rs.randint(low=0, high=n_batch, size=n_mini), device=dev)
        p_hat = model(x[idx, :]).unsqueeze(2)
        t0 = time.perf_counter()
        z = layer(Q[idx], p_hat, A[idx], b[idx], lb[idx], ub[idx])
        t1 = time.perf_counter()
        loss = 0.5 * torch.matmul(torch.matmul(z.transpose(1, 2), Q[idx]), z).sum() + (p[idx] * z).sum()
        opt.zero_grad()
        loss.backward()
        t2 = time.perf_counter()
        opt.step()
        losses.append(float(loss.detach()))
        t_fwd += t1 - t0
        t_bwd += t2 - t1
        if verbose and (epoch % 10 == 0 or epoch == n_epochs - 1):
            print(f"epoch {epoch:4d}  loss {losses[-1]:.6f}")
    return losses, t_fwd, t_bwd


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=500)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--mini", type=int, default=32)
    ap.add_argument("--epochs", type=int, default=100)
    a = ap.parse_args()
    losses, tf, tb = train(a.n, a.batch, a.mini, a.epochs)
    torch.cuda.synchronize()
    L.synchronize()
    print(f"n_x={a.n}: {a.epochs} epochs x minibatch {a.mini}: host time in forward {tf:.3f}s, backward {tb:.3f}s "
          f"(reference, 6-core i7: 19.6 s + 5.8 s at n_x=500, paper variant)")
