#!/usr/bin/env python3
"""One named workload of bench.py, alone, for the profilers (tools/profile_session.sh): K steps after W warm-up steps.

    python3 tools/profile_workload.py <workload> [--steps K] [--warmup W]

workloads (BASELINE.json configs and the rows of bench.py's `other_*` tables; inputs drawn on the device like bench.py's):
  config2       batch=128 dz=100 box-only, forward                       (BASELINE configs[1])
  headline      batch=128 dz=500 m=1, forward + backward                 (configs[2], the metric's configuration)
  lu            the same with control['linsolve']='lu' (pivoted LU + cached triangular solves: the north-star-named algorithm)
  config4       batch=128 dz=1000 m=1, forward                           (configs[3])
  config5shard  batch=1024 dz=500 m=1, forward + backward                (the per-GPU shard of configs[4])
  b16 / b32 / b64  the per-GPU shard of batch=128 at 8 / 4 / 2 GPUs, forward + backward
  hard64        experiments/experiment_1_hard.py: n=250, m=16, float64, forward + backward
  unroll        batch=128 dz=500 m=1 with control['unroll']=True (the reverse sweep + the native scaling chain), forward + backward
  n1500         batch=8 dz=1500 m=1: the LU tier above 1024 rows (several workgroups per matrix in the factorisation)
"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lqp_py_amd as L
from lqp_py_amd.synthetic import create_hard_qp_data

TOL = 1e-5


def device_batch(dev, B, n, seed, with_eq=True):
    gen = torch.Generator(device=dev).manual_seed(seed)
    Lm = torch.randn(B, 2 * n, n, device=dev, generator=gen)
    Q = torch.matmul(Lm.transpose(1, 2), Lm) / (2 * n)
    del Lm
    p = torch.randn(B, n, 1, device=dev, generator=gen)
    A = torch.ones(B, 1, n, device=dev) if with_eq else None
    b = torch.ones(B, 1, 1, device=dev) if with_eq else None
    lb = -(torch.rand(B, n, 1, device=dev, generator=gen) + 1)
    ub = torch.rand(B, n, 1, device=dev, generator=gen) + 1
    return Q, p, A, b, lb, ub


WORKLOADS = {   # name: (B, n, with_eq, backward, control extras, dtype)
    "config2": (128, 100, False, False, {}, "f32"),
    "headline": (128, 500, True, True, {}, "f32"),
    "lu": (128, 500, True, True, {"linsolve": "lu"}, "f32"),
    "config4": (128, 1000, True, False, {}, "f32"),
    "config5shard": (1024, 500, True, True, {}, "f32"),
    "b16": (16, 500, True, True, {}, "f32"),
    "b32": (32, 500, True, True, {}, "f32"),
    "b64": (64, 500, True, True, {}, "f32"),
    "hard64": (128, 250, True, True, {}, "f64"),
    "unroll": (128, 500, True, True, {"unroll": True, "sync": True}, "f32"),
    "n1500": (8, 1500, True, True, {}, "f32"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload", choices=sorted(WORKLOADS))
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=3)
    a = ap.parse_args()
    B, n, with_eq, backward, extra, dt = WORKLOADS[a.workload]
    dev = torch.device("cuda:0")
    if a.workload == "hard64":
        batch = create_hard_qp_data(n, 0.85, range(B), dtype=torch.float64, device=dev)
    else:
        batch = device_batch(dev, B, n, 4242 + n + B, with_eq)
    ctl = dict(dict(L.box_qp_control(eps_rel=TOL, eps_abs=TOL, verbose=False), sync=False), **extra)
    layer = L.SolveBoxQP(control=ctl)
    cot = torch.ones_like(batch[1])

    def step():
        if backward:
            Q = batch[0].detach().requires_grad_(True)
            p = batch[1].detach().requires_grad_(True)
            layer(Q, p, *batch[2:]).backward(cot)
        else:
            layer(*batch)
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize(dev)
    L.synchronize()
    dt_s = (time.perf_counter() - t0) / a.steps
    print(f'{{"workload": "{a.workload}", "batch": {B}, "n": {n}, "dtype": "{dt}", "backward": {str(backward).lower()}, '
          f'"steps": {a.steps}, "warmup": {a.warmup}, "ms_per_step": {dt_s * 1e3:.4f}, "QPs_per_sec": {B / dt_s:.1f}}}', flush=True)


if __name__ == "__main__":
    main()
