"""Batch-sharded box-QP solve across the GPUs of one node (one process per GPU).

Every QP of a batch is independent, so each rank solves its own contiguous
slice with no data-path communication; the only exchanges are
  * one 2-flag MAX all-reduce before the solve, because the reference's
    ``any_lb``/``any_ub`` (and hence the rho=0 shortcut) are global over the
    whole batch (lqp_py/solve_box_qp_admm_torch.py:33-38, 129-131); the flags
    stay on the device (the layer's setup kernel consumes them), nobody waits;
  * ONE all-gather of the solutions ``x`` (B_local, n, 1) at the end of the
    forward (RCCL over xGMI when the backend is "nccl"); shards may be of
    different sizes (B not a multiple of the world size);
  * only with ``control['dist_strict_stop'] = True``: one 4-word SUM all-reduce
    per convergence check, which makes the stop test (:312) and the adaptive-rho
    decision (:244-246) global like in the single-process reference -- every
    rank then reports the single-process iteration count.
Default: the stopping rule is evaluated per shard -- a shard stops when all of
ITS problems are optimal, so iteration counts may differ between shards
(results agree with the single-process solve within the tolerances).
``dl_dQ`` is never gathered.
"""
import torch
import torch.distributed as dist

from .solve_box_qp_admm_torch import SolveBoxQPLayer

_INF = float("inf")


_ALWAYS = [False]       # tests: issue the collectives even in a world of one rank (one-GPU boxes: the RCCL code paths)


def _active(group=None):
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or _ALWAYS[0])


def _needs_host_staging(t, group=None):
    """gloo moves host memory: device tensors are staged through the CPU (tests on a one-GPU box run two ranks
    over gloo; production uses "nccl" = RCCL, which takes device tensors as they are)."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def all_reduce_(t, op, group=None):
    """in-place all-reduce that also works for device tensors under the gloo backend"""
    if not _active(group):
        return t
    if _needs_host_staging(t, group):
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)
    return t


def global_bound_flags(lb, ub, group=None):
    """(any_lb, any_ub) over the batches of ALL ranks as HOST booleans: the local reductions and one tiny MAX
    all-reduce, read back (a host round trip).  Used for CPU tensors (the gloo tests' stand-in solver); device tensors go
    through device_bound_flags, which nobody waits for."""
    local = (bool(torch.max(lb) > -_INF), bool(torch.min(ub) < _INF))
    if not _active(group):
        return local
    flags = torch.tensor([int(local[0]), int(local[1])], dtype=torch.int32, device=lb.device)
    all_reduce_(flags, dist.ReduceOp.MAX, group)
    f = flags.tolist()
    return bool(f[0]), bool(f[1])


def device_bound_flags(lb, ub, group=None):
    """int32 [any_lb, any_ub] over the batches of ALL ranks, ON THE DEVICE: two reductions and one 8-byte MAX all-reduce
    (RCCL), all asynchronous.  The layer's setup kernel ORs them into what it reports back (lqp_boxqp_ctrl.bound_flags_in),
    so every rank verifies its schedule against the flags of the whole batch, as the reference would see them
    (:33-38, :129-131).  The collective is issued on EVERY call of every rank."""
    flags = torch.stack((torch.max(lb) > -_INF, torch.min(ub) < _INF)).to(torch.int32)
    return all_reduce_(flags, dist.ReduceOp.MAX, group)


def all_gather_solutions(x_local, group=None, sizes=None):
    """(B_r, n, 1) on rank r -> (sum_r B_r, n, 1) on every rank, rank order.  Equal shards: one
    all_gather_into_tensor.  Unequal shards: the sizes travel first (one small all-gather), the payloads are padded
    to the largest shard for the single payload collective and trimmed afterwards.  `sizes`: the shard sizes of all
    ranks when the caller knows them (it cut the batch): no size exchange, hence no host round trip per call."""
    if not _active(group):
        return x_local
    world = dist.get_world_size(group)
    x_local = x_local.contiguous()
    stage = _needs_host_staging(x_local, group)
    src = x_local.cpu() if stage else x_local
    if sizes is not None:
        sizes = [int(v) for v in sizes]
        if len(sizes) != world or sizes[dist.get_rank(group)] != src.shape[0]:
            raise ValueError(f"shard_sizes {sizes} do not describe this rank's shard of {src.shape[0]} problems")
    else:
        szt = torch.zeros(world, dtype=torch.int64, device=src.device)
        mine = torch.tensor([src.shape[0]], dtype=torch.int64, device=src.device)
        dist.all_gather_into_tensor(szt, mine, group=group)
        sizes = szt.tolist()
    bmax = max(sizes)
    tail = tuple(src.shape[1:])
    if all(s == bmax for s in sizes):
        out = torch.empty((world * bmax,) + tail, dtype=src.dtype, device=src.device)
        dist.all_gather_into_tensor(out, src, group=group)
    else:
        padded = src
        if src.shape[0] < bmax:
            padded = torch.zeros((bmax,) + tail, dtype=src.dtype, device=src.device)
            padded[:src.shape[0]] = src
        buf = torch.empty((world * bmax,) + tail, dtype=src.dtype, device=src.device)
        dist.all_gather_into_tensor(buf, padded, group=group)
        out = torch.cat([buf[r * bmax:r * bmax + sizes[r]] for r in range(world)], dim=0)
    return out.to(x_local.device) if stage else out


def shard_slice(n_total, rank, world):
    """contiguous slice [lo, hi) of rank `rank`; remainders go to the low ranks"""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class ShardedBoxQP(torch.nn.Module):
    """Same call signature as ``SolveBoxQP`` but the tensors passed in are this rank's
    slice of the batch.  ``forward`` returns (x_local, x_all): x_local carries the autograd
    graph (fixed-point backward on this rank's problems), x_all is the gathered solution.

    ``control['dist_strict_stop'] = True`` reproduces the single-process stopping rule and adaptive-rho decision
    exactly (one small all-reduce per convergence check); every rank must then pass the same control.
    ``shard_sizes`` (the number of problems on every rank, rank order): spares the size exchange of every forward; with it
    and finite bounds on this rank a pipelined forward (``control['sync'] = False``) has no host round trip."""

    def __init__(self, control, group=None, layer_apply=None, shard_sizes=None):
        super().__init__()
        self.control = control
        self.group = group
        self.shard_sizes = shard_sizes      # sizes of all ranks' shards, if the caller knows them (see all_gather_solutions)
        self._apply = layer_apply or SolveBoxQPLayer.apply      # tests inject a CPU stand-in here

    def _check_hook(self, counters, check_index):
        """counters: 4 int32 {not optimal, arrivals, wants rho, ratio trigger} of one check, in place"""
        all_reduce_(counters, dist.ReduceOp.SUM, self.group)

    def forward(self, Q, p, A, b, lb, ub):
        ctl = self.control
        if _active(self.group):
            # private keys for the layer: the flags of the WHOLE batch (a shard without any finite bound must still run
            # the ADMM path the whole batch runs) and, in strict mode, the per-check all-reduce
            ctl = dict(ctl)
            ctl['_owner'] = self.control        # (the dict side effect control['rho'] = 0 belongs to the caller's dict)
            ctl['_holder'] = self               # (what the layer remembers between calls, it remembers per module)
            if lb.is_cuda:
                ctl['_bound_flags_dev'] = device_bound_flags(lb, ub, self.group)
            else:                               # CPU stand-in solver of the gloo tests: host flags
                ctl['_global_bounds'] = global_bound_flags(lb, ub, self.group)
            if ctl.get('dist_strict_stop', False) and any(ctl.get('_global_bounds', (True,))):
                ctl['_check_hook'] = self._check_hook
        x_local = self._apply(Q, p, A, b, lb, ub, ctl)
        if ctl is not self.control and ctl.get('rho', None) is not self.control.get('rho', None):
            self.control['rho'] = ctl['rho']        # the layer's dict side effect (:37-38) belongs to the caller's dict
        with torch.no_grad():
            x_all = all_gather_solutions(x_local.detach(), self.group, self.shard_sizes)
        return x_local, x_all
