"""Synthetic box-QP batches for the benchmark and the examples.

The distribution and, more importantly, the ORDER of the random draws follow the
reference's experiment harness (experiments/utils.py:41-61 and the fixed unit box of
demo/demo_solve_box_qp_torch.py:19-20), so that a given seed yields the very batch the
reference's experiment_1 times.  Tensors are drawn with the CPU generator (identical on
every machine) and moved to `device` afterwards.
"""
import torch


def create_qp_data(n_x, n_batch, n_samples=None, seed=0, with_eq=True, unit_box=False, dtype=torch.float32,
                   device=None):
    """(Q, p, A, b, lb, ub): Q = LᵀL / n_samples with L ~ N(0,1) of shape (B, n_samples, n_x) [n_samples
    defaults to 2 n_x], p ~ N(0,1), A = 1ᵀ, b = 1 (or None, None), lb = -U(1,2), ub = U(1,2)."""
    n_samples = 2 * n_x if n_samples is None else n_samples
    torch.manual_seed(seed)
    factor = torch.randn(n_batch, n_samples, n_x)
    Q = factor.transpose(1, 2) @ factor / n_samples
    p = torch.randn(n_batch, n_x, 1)
    if unit_box:
        lb, ub = -torch.ones(n_batch, n_x, 1), torch.ones(n_batch, n_x, 1)
    else:
        lb = -(torch.rand(n_batch, n_x, 1) * (2 - 1) + 1)
        ub = torch.rand(n_batch, n_x, 1) * (2 - 1) + 1
    A = torch.ones(n_batch, 1, n_x) if with_eq else None
    b = torch.ones(n_batch, 1, 1) if with_eq else None
    return tuple(None if t is None else t.to(dtype=dtype, device=device) for t in (Q, p, A, b, lb, ub))


def create_hard_qp_data(n_x, prob, seeds, dtype=torch.float64, device=None):
    """The "hard" distribution of experiments/experiment_1_hard.py (experiments/utils.py:64-131): sparse factor
    Q = GᵀG + 0.01 I with G ~ N(0,1) masked by Bernoulli(prob), m = round(sqrt(n_x)) sparse equality rows through a
    feasible point x0, bounds x0 - U(0,1) .. x0 + U(0,1); one problem per seed, NumPy's legacy global generator (so the
    draws -- and their order -- are those of the reference on every machine).  -> (Q, p, A, b, lb, ub)."""
    import numpy as np
    m = round(n_x ** 0.5)
    out = [[] for _ in range(6)]
    for seed in seeds:
        np.random.seed(seed)
        G = np.random.normal(size=(n_x, n_x)) * np.random.binomial(1, prob, size=(n_x, n_x))
        Q = G.T.dot(G) + 1e-2 * np.eye(n_x)
        p = np.random.normal(size=(n_x, 1))
        x0 = np.random.normal(size=(n_x, 1))
        below = np.random.uniform(size=(n_x, 1))
        above = np.random.uniform(size=(n_x, 1))
        A = np.zeros((m, n_x))
        for r in range(m):
            row = np.random.normal(size=(1, n_x))
            keep = np.zeros(1)
            while keep.sum() == 0:                       # (an all-zero row is drawn again)
                keep = np.random.binomial(1, prob, size=(1, n_x))
            A[r] = row * keep
        for k, v in enumerate((Q, p, A, A.dot(x0), x0 - below, x0 + above)):
            out[k].append(v)
    return tuple(torch.tensor(np.stack(v), dtype=dtype, device=device) for v in out)
