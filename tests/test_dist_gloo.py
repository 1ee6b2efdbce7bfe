"""world_size-2 gloo test of the batch-sharded path (host logic only): the
sharding helper, the global any_lb/any_ub reduction and the single all-gather.
The per-rank solver is injected (the CPU oracle stands in for the HIP layer,
which needs a GPU)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import boxqp_oracle as O


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _oracle_layer(Q, p, A, b, lb, ub, control):
    return O.layer_forward(Q, p, A, b, lb, ub, control)["x"]


def _worker(rank, world, port, B, n, out_dir, no_bounds_rank):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from lqp_py_amd.dist import ShardedBoxQP, shard_slice
    Q, p, A, b, lb, ub = O.create_qp_data(n, B, seed=5)
    if no_bounds_rank is not None:        # one shard without any finite bound: flags must still be global
        lo, hi = shard_slice(B, no_bounds_rank, world)
        lb[lo:hi] = -float("inf")
        ub[lo:hi] = float("inf")
    lo, hi = shard_slice(B, rank, world)
    ctl = O.make_control(eps_abs=1e-5, eps_rel=1e-5)
    layer = ShardedBoxQP(ctl, layer_apply=_oracle_layer)
    x_local, x_all = layer(Q[lo:hi], p[lo:hi], A[lo:hi], b[lo:hi], lb[lo:hi], ub[lo:hi])
    assert x_all.shape == (B, n, 1)
    assert torch.equal(x_all[lo:hi], x_local)
    assert ctl.get("rho") is None, "caller's dict must not be touched when any rank has a finite bound"
    if rank == 0:
        torch.save(x_all, os.path.join(out_dir, "x_all.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("no_bounds_rank", [None, 1])
def test_sharded_equals_single_process(tmp_path, no_bounds_rank):
    B, n, world = 6, 24, 2
    mp.spawn(_worker, args=(world, _free_port(), B, n, str(tmp_path), no_bounds_rank), nprocs=world, join=True)
    x_all = torch.load(os.path.join(tmp_path, "x_all.pt"))
    Q, p, A, b, lb, ub = O.create_qp_data(n, B, seed=5)
    if no_bounds_rank is not None:
        from lqp_py_amd.dist import shard_slice
        lo, hi = shard_slice(B, no_bounds_rank, world)
        lb[lo:hi] = -float("inf")
        ub[lo:hi] = float("inf")
    ref = O.solve_box_qp(Q, p, A, b, lb, ub, O.make_control(eps_abs=1e-5, eps_rel=1e-5))
    # per-shard stopping may run a shard a few checks longer/shorter than the global rule: tolerance, not equality
    torch.testing.assert_close(x_all, ref["x"], atol=2e-4, rtol=2e-4)


def test_shard_slices_cover_the_batch():
    from lqp_py_amd.dist import shard_slice
    for total in (1, 7, 128, 130):
        for world in (1, 2, 3, 8):
            parts = [shard_slice(total, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
