"""Batch-sharded box-QP solve across the GPUs of one node (one process per GPU).

Every QP of a batch is independent, so each rank solves its own contiguous
slice with no data-path communication; the only exchanges are
  * one 2-flag MAX all-reduce before the solve, because the reference's
    ``any_lb``/``any_ub`` (and hence the rho=0 shortcut) are global over the
    whole batch (lqp_py/solve_box_qp_admm_torch.py:33-38, 129-131), and
  * ONE all-gather of the solutions ``x`` (B_local, n, 1) at the end of the
    forward (RCCL over xGMI when the backend is "nccl").
The global stopping rule (:312) is evaluated per shard: a shard stops when all
of ITS problems are optimal, so iteration counts may differ between shards
(results agree with the single-process solve within the tolerances).
``dl_dQ`` is never gathered.
"""
import torch
import torch.distributed as dist

from .solve_box_qp_admm_torch import SolveBoxQPLayer

_INF = float("inf")


_flag_cache = []     # [(weakref(lb), version, weakref(ub), version, group, result)]


def global_bound_flags(lb, ub, group=None):
    """(any_lb, any_ub) over the batches of ALL ranks.  Two reductions, one tiny all-reduce and a host
    sync -- remembered for the same live tensor objects at the same in-place version.  (Every rank
    runs the same program on its own shard, so all ranks hit or miss together and the collective
    stays matched; set LQP_DIST_NO_CACHE=1 if your ranks do not pass bounds in lockstep.)"""
    import os
    import weakref
    use_cache = not os.environ.get("LQP_DIST_NO_CACHE")
    if use_cache:
        for rl, vl, ru, vu, g, res in _flag_cache:
            if rl() is lb and ru() is ub and vl == lb._version and vu == ub._version and g is group:
                return res
    flags = torch.stack(((torch.max(lb) > -_INF), (torch.min(ub) < _INF))).to(torch.int32)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=group)
    f = flags.tolist()
    res = (bool(f[0]), bool(f[1]))
    if use_cache:
        _flag_cache[:] = [e for e in _flag_cache if e[0]() is not None and e[2]() is not None][-7:]
        _flag_cache.append((weakref.ref(lb), lb._version, weakref.ref(ub), ub._version, group, res))
    return res


def _local_flags(lb, ub):
    if not lb.is_cuda:                      # CPU stand-in used by the gloo tests
        return bool(torch.max(lb) > -_INF), bool(torch.min(ub) < _INF)
    from .solve_box_qp_admm_torch import _finite_bounds
    return _finite_bounds(lb, ub)


def all_gather_solutions(x_local, group=None):
    """(B_local, n, 1) on every rank -> (world * B_local, n, 1) on every rank, one collective."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return x_local
    world = dist.get_world_size(group)
    out = torch.empty((world * x_local.shape[0],) + tuple(x_local.shape[1:]), dtype=x_local.dtype,
                      device=x_local.device)
    dist.all_gather_into_tensor(out, x_local.contiguous(), group=group)
    return out


def shard_slice(n_total, rank, world):
    """contiguous slice [lo, hi) of rank `rank`; remainders go to the low ranks"""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class ShardedBoxQP(torch.nn.Module):
    """Same call signature as ``SolveBoxQP`` but the tensors passed in are this rank's
    slice of the batch.  ``forward`` returns (x_local, x_all): x_local carries the autograd
    graph (fixed-point backward on this rank's problems), x_all is the gathered solution."""

    def __init__(self, control, group=None, layer_apply=None):
        super().__init__()
        self.control = control
        self.group = group
        self._apply = layer_apply or SolveBoxQPLayer.apply      # tests inject a CPU stand-in here

    def forward(self, Q, p, A, b, lb, ub):
        has_lb, has_ub = global_bound_flags(lb, ub, self.group)
        ctl = self.control
        if not (has_lb or has_ub):
            ctl['rho'] = 0
        elif _local_flags(lb, ub) == (False, False):
            # this shard alone has no finite bound but another rank does: keep the ADMM path
            # (a private copy of the dict protects the caller's rho from the layer's side effect)
            ctl = dict(ctl)
        x_local = self._apply(Q, p, A, b, lb, ub, ctl)
        with torch.no_grad():
            x_all = all_gather_solutions(x_local.detach(), self.group)
        return x_local, x_all
