#!/usr/bin/env python3
"""What the sharded layer adds to a step on ONE rank (world size 1, RCCL): its collectives are launched like on any world size
(flags all-reduce, all-gather of the solutions), only the links are missing.  Pipelined calls, B = 128, n = 500."""
import os, sys, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lqp_py_amd as L
from lqp_py_amd.dist import ShardedBoxQP
from lqp_py_amd.synthetic import create_qp_data
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
B, n = 128, 500
data = [[t.to(dev) for t in create_qp_data(n, B, seed=s)] for s in range(4)]
ones = torch.ones(B, n, 1, device=dev)
def run(layer, sharded, K=40):
    def step(i):
        Q, p, A, b, lb, ub = data[i % 4]
        Q = Q.detach().requires_grad_(True); p = p.detach().requires_grad_(True)
        out = layer(Q, p, A, b, lb, ub)
        (out[0] if sharded else out).backward(ones)
    for i in range(10): step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(K): step(i)
    torch.cuda.synchronize(); L.synchronize()
    return (time.perf_counter() - t0) / K * 1e3
ctl = lambda: dict(L.box_qp_control(eps_abs=1e-5, eps_rel=1e-5), sync=False)
for rep in range(2):
    a = run(L.SolveBoxQP(control=ctl()), False)
    b_ = run(ShardedBoxQP(ctl(), shard_sizes=[B]), True)
    print(f"plain layer {a:.4f} ms/step   sharded layer (world 1) {b_:.4f} ms/step   +{(b_ - a) * 1e3:.0f} us")
dist.destroy_process_group()
