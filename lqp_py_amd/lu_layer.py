"""Cached batched LU solve as a differentiable op on MI355X.

Drop-in for ``lqp_py/lu_layer.py`` (``TorchLU`` :5-16, ``TorchLULayer`` :19-58).
``LU``/``P`` follow ``torch.linalg.lu_factor`` conventions (packed LU, int32
1-based pivots), so factors made by torch and by ``lu_factor`` below are
interchangeable.  The factor is re-laid out once into solve-ordered panels
(``lqp_lu_pack``) and cached on the module, then every solve streams it.
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib


def lu_factor(A):
    """Batched LU with partial pivoting on the GPU -> (LU, pivots); raises on an exactly zero pivot
    like torch.linalg.lu_factor does."""
    _lib.require_gpu(A)
    lib = _lib.load()
    B, N = A.shape[0], A.shape[1]
    dt = _lib.dtype_code(A)
    LU = A.detach().clone().contiguous()
    piv = torch.empty((B, N), dtype=torch.int32, device=A.device)
    info = torch.empty((B,), dtype=torch.int32, device=A.device)
    ws = _lib.workspace(A.device, lib.lqp_lu_factor_workspace_bytes(dt, B, N), "lu")
    with torch.cuda.device(A.device):
        st = lib.lqp_lu_factor_batched(_lib.stream_ptr(A.device), dt, B, N, _lib.ptr(LU), _lib.ptr(piv),
                                       _lib.ptr(info), _lib.ptr(ws), ws.numel())
    _lib.check(st, "lu_factor")
    bad = torch.nonzero(info)
    if bad.numel():
        i = int(bad[0])
        raise RuntimeError(f"lqp_py_amd.lu_factor: (Batch element {i}): U[{int(info[i])},{int(info[i])}] is zero "
                           "and using it on lu_solve would result in a division by zero.")
    return LU, piv


class _PackedFactor:
    """Solve-ordered copy of (LU, P), built lazily, reused by every solve."""

    def __init__(self, LU, P):
        self.LU, self.P = LU, P
        self._buf = None

    def buffer(self):
        if self._buf is None:
            lib = _lib.load()
            LU, P = _lib.c(self.LU), _lib.c(self.P).to(torch.int32)
            B, N = LU.shape[0], LU.shape[1]
            dt = _lib.dtype_code(LU)
            buf = torch.empty(lib.lqp_lu_packed_bytes(dt, B, N), dtype=torch.uint8, device=LU.device)
            with torch.cuda.device(LU.device):
                _lib.check(lib.lqp_lu_pack(_lib.stream_ptr(LU.device), dt, B, N, _lib.ptr(LU), _lib.ptr(P),
                                           _lib.ptr(buf)), "lu_pack")
            self._buf = buf
        return self._buf

    def solve(self, rhs):
        lib = _lib.load()
        LU = self.LU
        B, N = LU.shape[0], LU.shape[1]
        out = rhs.detach().to(LU.dtype).contiguous().clone()
        squeeze = out.dim() == 2
        k = 1 if squeeze else out.shape[2]
        with torch.cuda.device(LU.device):
            _lib.check(lib.lqp_lu_solve_packed(_lib.stream_ptr(LU.device), _lib.dtype_code(LU), B, N, k,
                                               _lib.ptr(self.buffer()), _lib.ptr(out)), "lu_solve")
        return out


def lu_solve(LU, P, rhs):
    """One-off solve with a torch-layout factor (packs, then solves)."""
    _lib.require_gpu(LU, P, rhs)
    return _PackedFactor(LU, P).solve(rhs)


class TorchLU(nn.Module):
    def __init__(self, A=None, LU=None, P=None):
        super().__init__()
        if LU is None or P is None:
            with torch.no_grad():
                LU, P = lu_factor(A)
        self.LU = LU
        self.P = P
        self._packed = _PackedFactor(LU, P)

    def forward(self, A, b):
        return TorchLULayer.apply(A, b, self.LU, self.P, self._packed)


class TorchLULayer(torch.autograd.Function):
    """x = A^-1 b through the cached factor; analytic backward valid for symmetric A
    (reference :24-58): dA = dx x^T, db = -dx with dx = lu_solve(LU, P, -dl_dx)."""

    @staticmethod
    def forward(ctx, A, b, LU=None, P=None, packed=None):
        if LU is None or P is None:
            with torch.no_grad():
                LU, P = lu_factor(A)
            packed = None
        if packed is None:
            packed = _PackedFactor(LU, P)
        x = packed.solve(b)
        ctx.packed = packed
        ctx.save_for_backward(x)
        return x

    @staticmethod
    def backward(ctx, dl_dx):
        (x,) = ctx.saved_tensors
        dx = ctx.packed.solve(-dl_dx)
        xt = x.unsqueeze(1) if x.dim() < 3 else torch.transpose(x, 1, 2)
        dl_dA = torch.matmul(dx if dx.dim() == 3 else dx.unsqueeze(2), xt)
        return dl_dA, -dx, None, None, None
