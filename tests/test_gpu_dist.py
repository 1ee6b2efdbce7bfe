"""Two ranks drive the REAL ``ShardedBoxQP`` (HIP layer, C ABI) on one GPU box.

The box has one GPU, and RCCL refuses two ranks on one device, so the process
group is gloo (``lqp_py_amd.dist`` stages the few small collective payloads
through the host for that backend; under "nccl" = RCCL they stay on the
device).  Everything else is the production path: each rank solves its shard
with the HIP library, the all-gather returns the full solution, the backward
runs per shard, and -- in strict mode -- the library calls back after every
convergence check for the all-reduce of its counters."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import boxqp_oracle as O

pytestmark = pytest.mark.gpu
TOL = dict(eps_abs=1e-5, eps_rel=1e-5)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _data(n, B):
    Q, p, A, b, lb, ub = O.create_qp_data(n, B, seed=11)
    Q[B // 2:] *= 40.0                      # the two halves of the batch converge at different checks
    p[B // 2:] *= 3.0
    return Q, p, A, b, lb, ub


def _worker(rank, world, port, B, n, out_dir, strict):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import lqp_py_amd as L
    from lqp_py_amd.dist import ShardedBoxQP, shard_slice
    from lqp_py_amd.solve_box_qp_admm_torch import last_forward_status
    dev = torch.device("cuda:0")
    lo, hi = shard_slice(B, rank, world)
    Q, p, A, b, lb, ub = (t[lo:hi].to(dev) for t in _data(n, B))
    ctl = L.box_qp_control(**TOL)
    ctl["check_solved"] = 5
    if strict:
        ctl["dist_strict_stop"] = True
    Qg = Q.clone().requires_grad_(True)
    pg = p.clone().requires_grad_(True)
    x_local, x_all = ShardedBoxQP(ctl)(Qg, pg, A, b, lb, ub)
    st = last_forward_status(dev)
    cot = torch.ones_like(x_local)
    x_local.backward(cot)
    torch.cuda.synchronize()
    assert x_all.shape == (B, n, 1) and torch.equal(x_all[lo:hi], x_local.detach())
    assert "_check_hook" not in ctl and "_global_bounds" not in ctl
    torch.save({"x_all": x_all.cpu(), "iter": st["iters"], "mode": st["mode_used"], "dp": pg.grad.cpu(),
                "dQ_fro": torch.linalg.matrix_norm(Qg.grad).cpu()}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("strict", [False, True])
def test_two_ranks_drive_the_hip_layer(tmp_path, strict):
    assert torch.cuda.is_available()
    B, n, world = 7, 96, 2                   # unequal shards (4 + 3)
    mp.spawn(_worker, args=(world, _free_port(), B, n, str(tmp_path), strict), nprocs=world, join=True)
    outs = [torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(world)]
    assert torch.equal(outs[0]["x_all"], outs[1]["x_all"])
    Q, p, A, b, lb, ub = _data(n, B)
    ctl = O.make_control(**TOL)
    ctl["check_solved"] = 5
    ref = O.solve_box_qp(Q, p, A, b, lb, ub, ctl)
    err = float((outs[0]["x_all"] - ref["x"]).abs().max())
    if strict:
        # one all-reduce per check: the single-process stopping rule, hence its iteration count, on every rank
        assert [o["iter"] for o in outs] == [ref["iter"]] * world and all(o["mode"] == 1 for o in outs)
        assert err < 2e-5, err
    else:
        assert err < 1e-3 * max(1.0, float(ref["x"].abs().max())), err      # per-shard stop: within the stopping tolerance
    # gradients of each shard against the oracle's for the same problems
    g = O.solve_box_qp_grad(torch.ones(B, n, 1), ref["x"], ref["u"], ref["lams"], ref["nus"], Q, A, lb, ub, ref["rho"])
    from lqp_py_amd.dist import shard_slice
    for r, o in enumerate(outs):
        lo, hi = shard_slice(B, r, world)
        scale = max(1.0, float(g[1][lo:hi].abs().max()))
        assert float((o["dp"] - g[1][lo:hi]).abs().max()) < (1e-3 if not strict else 2e-4) * scale


def _trigger_data(n=50, B=8):
    """first half: G6-like problems whose residual ratio asks for a new rho at iteration 100; second half: problems that
    are optimal after 20 iterations and never trigger anything"""
    Q, p, A, b, lb, ub = O.create_qp_data(n, B, seed=3)
    Q = Q * 50
    Q[B // 2:] = 100 * torch.eye(n)
    return Q, p, A, b, lb, ub


def _trigger_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import lqp_py_amd as L
    from lqp_py_amd.dist import ShardedBoxQP, shard_slice
    from lqp_py_amd.solve_box_qp_admm_torch import last_forward_status
    dev = torch.device("cuda:0")
    Qa, pa, Aa, ba, lba, uba = _trigger_data()
    lo, hi = shard_slice(Qa.shape[0], rank, world)
    Q, p, A, b, lb, ub = (t[lo:hi].to(dev) for t in (Qa, pa, Aa, ba, lba, uba))
    ctl = L.box_qp_control(rho=100.0, scale=False, dist_strict_stop=True, **TOL)
    x_local, x_all = ShardedBoxQP(ctl)(Q, p, A, b, lb, ub)
    st = last_forward_status(dev)
    torch.cuda.synchronize()
    torch.save({"x_all": x_all.cpu(), "iter": st["iters"], "n_factor": st["n_factor"]}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_strict_stop_shares_the_ratio_trigger(tmp_path):
    """Only rank 0 holds problems whose residual ratio triggers the adaptive-rho step at iteration 100 (:244-246); the
    decision is global in the reference, so BOTH ranks refactorise (all four counter words are all-reduced) and both
    report the single-process iteration count."""
    world = 2
    mp.spawn(_trigger_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(world)]
    Q, p, A, b, lb, ub = _trigger_data()
    tr = {}
    ref = O.solve_box_qp(Q, p, A, b, lb, ub, O.make_control(rho=100.0, scale=False, **TOL), trace=tr)
    assert tr["n_factor"] == 2 and ref["iter"] == 100
    assert [o["iter"] for o in outs] == [ref["iter"]] * world, [o["iter"] for o in outs]
    assert [o["n_factor"] for o in outs] == [2] * world, [o["n_factor"] for o in outs]
    assert torch.equal(outs[0]["x_all"], outs[1]["x_all"])
    e = float((outs[0]["x_all"] - ref["x"]).abs().max())
    assert e < 5e-5 * max(1.0, float(ref["x"].abs().max())), e


def _nccl_one_rank_worker(port, out_path):
    """child process: a process group of ONE rank over "nccl" (= RCCL) on the only GPU of the box; the sharded layer is
    told to issue its collectives all the same (lqp_py_amd.dist._ALWAYS), so the RCCL initialisation, the device-tensor
    all-reduce of the bound flags, all_gather_into_tensor and the pipelined (sync=False) path run for real"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    import lqp_py_amd as L
    import lqp_py_amd.dist as D
    D._ALWAYS[0] = True
    n, B = 96, 6
    Q, p, A, b, lb, ub = (t.to(dev) for t in O.create_qp_data(n, B, seed=13))
    out = {}
    for sync in (True, False):
        ctl = L.box_qp_control(sync=sync, **TOL)
        layer = D.ShardedBoxQP(ctl, shard_sizes=[B])
        Qg = Q.clone().requires_grad_(True)
        pg = p.clone().requires_grad_(True)
        for _ in range(3):                                   # steady state of a pipelined loop: nothing waits for the host
            x_local, x_all = layer(Qg, pg, A, b, lb.clone(), ub.clone())
        x_local.backward(torch.ones_like(x_local))
        L.synchronize()
        torch.cuda.synchronize()
        assert torch.equal(x_all, x_local.detach())
        out[f"x_{int(sync)}"] = x_all.cpu()
        out[f"dp_{int(sync)}"] = pg.grad.cpu()
    # unequal-shard code path (sizes exchanged on the device) and the strict stop's all-reduce hook
    ctl = L.box_qp_control(dist_strict_stop=True, **TOL)
    x_local, x_all = D.ShardedBoxQP(ctl)(Q, p, A, b, lb, ub)
    torch.cuda.synchronize()
    out["x_strict"] = x_all.cpu()
    out["backend"] = dist.get_backend()
    torch.save(out, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_one_rank_nccl_group_drives_the_sharded_layer(tmp_path):
    """The production backend on the hardware at hand: RCCL with a world of one rank (a one-GPU box cannot form a larger
    one).  Every collective of lqp_py_amd.dist is issued on device tensors; results equal the CPU oracle's."""
    out_path = os.path.join(tmp_path, "nccl1.pt")
    ctx = mp.get_context("spawn")
    pr = ctx.Process(target=_nccl_one_rank_worker, args=(_free_port(), out_path))
    pr.start()
    pr.join(300)
    assert pr.exitcode == 0, f"child exit code {pr.exitcode}"
    out = torch.load(out_path)
    assert out["backend"] == "nccl"
    n, B = 96, 6
    Q, p, A, b, lb, ub = O.create_qp_data(n, B, seed=13)
    ref = O.solve_box_qp(Q, p, A, b, lb, ub, O.make_control(**TOL))
    g = O.solve_box_qp_grad(torch.ones(B, n, 1), ref["x"], ref["u"], ref["lams"], ref["nus"], Q, A, lb, ub, ref["rho"])
    for key in ("x_1", "x_0", "x_strict"):
        assert float((out[key] - ref["x"]).abs().max()) < 2e-5, key
    assert torch.equal(out["x_1"], out["x_0"])
    for key in ("dp_1", "dp_0"):
        assert float((out[key] - g[1]).abs().max()) < 2e-4 * max(1.0, float(g[1].abs().max())), key


def _unbounded_shard_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import lqp_py_amd as L
    from lqp_py_amd.dist import ShardedBoxQP, shard_slice
    from lqp_py_amd.solve_box_qp_admm_torch import last_forward_status
    dev = torch.device("cuda:0")
    n, B = 64, 6
    Q, p, A, b, lb, ub = O.create_qp_data(n, B, seed=21)
    lb[B // 2:] = -float("inf")              # rank 1's shard holds no finite bound at all; the batch does
    ub[B // 2:] = float("inf")
    lo, hi = shard_slice(B, rank, world)
    Q, p, A, b, lb, ub = (t[lo:hi].to(dev) for t in (Q, p, A, b, lb, ub))
    ctl = L.box_qp_control(sync=False, **TOL)
    keys = set(ctl)
    layer = ShardedBoxQP(ctl, shard_sizes=[3, 3])
    x_local, x_all = layer(Q, p, A, b, lb, ub)          # the FIRST un-synchronised call of this layer
    L.synchronize()                                      # (a schedule chosen from the local bounds would raise here)
    st = last_forward_status(dev)
    x_local2, x_all2 = layer(Q, p, A, b, lb, ub)
    L.synchronize()
    torch.cuda.synchronize()
    assert set(ctl) == keys, set(ctl) ^ keys             # nothing private is left in the caller's dict
    torch.save({"x_all": x_all.cpu(), "x_all2": x_all2.cpu(), "iter": st["iters"], "rho": ctl.get("rho")},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_first_pipelined_call_of_a_shard_without_bounds(tmp_path):
    """the first `sync=False` call of a layer picked rho = 0 vs ADMM from THIS rank's bounds; a shard without
    any finite bound inside a batch that has some must enqueue the ADMM schedule its peers enqueue (the flags of the
    whole batch decide, :33-38 / :129-131) -- no late 'bound flags differ' error on one rank, no hang on the others."""
    world = 2
    mp.spawn(_unbounded_shard_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(world)]
    assert torch.equal(outs[0]["x_all"], outs[1]["x_all"]) and torch.equal(outs[0]["x_all"], outs[0]["x_all2"])
    n, B = 64, 6
    Q, p, A, b, lb, ub = O.create_qp_data(n, B, seed=21)
    lb[B // 2:] = -float("inf")
    ub[B // 2:] = float("inf")
    ref = O.solve_box_qp(Q, p, A, b, lb, ub, O.make_control(**TOL))
    assert all(o["rho"] is None for o in outs)           # the rho = 0 side effect did not fire: the batch has bounds
    e = float((outs[0]["x_all"] - ref["x"]).abs().max())
    assert e < 1e-3 * max(1.0, float(ref["x"].abs().max())), e      # (per-shard stop: within the stopping tolerance)
