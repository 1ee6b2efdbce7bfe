"""ctypes binding of the C ABI declared in include/lqp_amd.h.

The shared library is built in-tree (``lqp_py_amd/csrc/liblqp_amd.so``) by
``build_library()`` / ``__graft_entry__.build()``.  Loading fails loudly: there
is no fallback path.
"""
import ctypes
import os
import subprocess
import threading

import numpy

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.environ.get("LQP_LIB", os.path.join(CSRC, "liblqp_amd.so"))   # LQP_LIB: A/B builds
SOURCES = ["lqp_amd.hip", "lqp_unroll.hpp", "lqp_boxqp.hpp", "lqp_lu.hpp", "lqp_lu_big.hpp", "lqp_lu2.hpp", "lqp_lu_wide.hpp", "lqp_dense.hpp", "lqp_trsv.hpp", "lqp_spd.hpp", "lqp_common.hpp"]

LQP_F32, LQP_F64 = 0, 1
ABI_VERSION = 13
STATUS = {0: "ok", 1: "invalid argument", 2: "workspace too small", 3: "singular", 4: "HIP error",
          5: "grid barrier timeout", 6: "unsupported size (n + m <= 4096 in float32, 2048 in float64)", 7: "matrix outside the symmetric x-update"}

c_void_p, c_int, c_size_t, c_double = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_double


# int check_hook(void* user, void* stream, void* counters_dev, int check_index)   (include/lqp_amd.h)
CHECK_HOOK = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int)


class BoxQPCtrl(ctypes.Structure):
    _fields_ = [(k, ctypes.c_int32) for k in (
        "max_iters", "check_solved", "adaptive_rho", "adaptive_rho_iter", "adaptive_rho_max_iter", "scale",
        "any_lb", "any_ub", "rho_mode", "beta_mode", "launch_mode", "reserved", "linsolve", "reserved2")] + [
        (k, ctypes.c_double) for k in (
            "eps_abs", "eps_rel", "rho_value", "rho_min", "rho_max", "adaptive_rho_tol",
            "adaptive_rho_threshold", "beta_value")] + [
        ("beta_in", ctypes.c_void_p), ("check_hook", CHECK_HOOK), ("check_hook_user", ctypes.c_void_p),
        ("bound_flags_in", ctypes.c_void_p), ("host_report", ctypes.c_void_p)]


class BoxQPStats(ctypes.Structure):
    _fields_ = [(k, ctypes.c_int32) for k in (
        "iters", "n_factor", "n_solve", "n_check", "rho_updated", "fail_index", "n_launch", "mode_used", "linsolve_used",
        "factor_launches", "loop_workgroups", "any_lb", "any_ub")]


# every symbol include/lqp_amd.h declares: name -> (restype, argtypes)
_P = c_void_p
SYMBOLS = {
    "lqp_abi_version": (c_int, []),
    "lqp_status_string": (ctypes.c_char_p, [c_int]),
    "lqp_profile_enable": (None, [c_int]),
    "lqp_profile_reset": (None, []),
    "lqp_profile_classes": (c_int, []),
    "lqp_profile_class_name": (ctypes.c_char_p, [c_int]),
    "lqp_profile_get": (c_int, [ctypes.POINTER(c_double), ctypes.POINTER(ctypes.c_longlong), c_int]),
    "lqp_debug_set_lu_counters": (None, [_P]),
    "lqp_debug_spin": (c_int, [_P, c_int, c_int, c_int]),
    "lqp_debug_xcd": (c_int, [_P, c_int, _P]),
    "lqp_debug_lu_inverse": (c_int, [_P, c_int, c_int, c_int, _P, _P]),
    "lqp_boxqp_forward_workspace_bytes": (c_size_t, [c_int] * 4),
    "lqp_boxqp_forward": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P,
                                  ctypes.POINTER(BoxQPCtrl), _P, _P, _P, _P, _P, _P, _P,
                                  ctypes.POINTER(BoxQPStats), _P, c_size_t]),
    "lqp_boxqp_forward_finish": (c_int, [_P, c_int, c_int, c_int, _P, ctypes.POINTER(BoxQPStats)]),
    "lqp_boxqp_forward_layout": (c_int, [c_int] * 4 + [ctypes.POINTER(c_size_t)] * 4),
    "lqp_boxqp_unroll_backward_workspace_bytes": (c_size_t, [c_int] * 4),
    "lqp_boxqp_unroll_backward": (c_int, [_P, c_int, c_int, c_int, _P, c_size_t, c_int] + [_P] * 10 + [c_size_t]),
    "lqp_boxqp_unroll_backward_lu_workspace_bytes": (c_size_t, [c_int] * 5),
    "lqp_boxqp_unroll_backward_lu": (c_int, [_P, c_int, c_int, c_int, c_int, _P, c_size_t, c_int] + [_P] * 10 + [c_size_t]),
    "lqp_boxqp_unroll_tape_workspace_bytes": (c_size_t, [c_int] * 5),
    "lqp_boxqp_unroll_tape_segment": (c_int, [_P, c_int, c_int, c_int, c_int, _P, c_size_t, c_int, c_int, c_int, c_int, _P, _P, _P, c_int]
                                      + [_P] * 8 + [c_size_t] + [_P] * 3),
    "lqp_boxqp_unroll_tape_finish": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_size_t]),
    "lqp_unroll_scale_colmax": (c_int, [_P, c_int, c_int, _P, _P, _P, _P]),
    "lqp_unroll_scale_grad_slabs": (c_int, [c_int, c_int]),
    "lqp_unroll_scale_grad": (c_int, [_P, c_int, c_int, _P, _P, _P, _P, _P, c_int]),
    "lqp_unroll_scale_vectors": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, c_double] + [_P] * 13 + [c_int] + [_P] * 7),
    "lqp_unroll_scale_scatter": (c_int, [_P, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    "lqp_boxqp_last_residuals": (c_int, [_P, c_int, c_int, c_int, c_int, _P, c_size_t, _P, _P]),
    "lqp_boxqp_check_trace": (c_int, [_P, c_int, c_int, c_int, c_int, _P, c_size_t, c_int, _P]),
    "lqp_boxqp_backward_fp_workspace_bytes": (c_size_t, [c_int] * 4),
    "lqp_boxqp_backward_fp_prefactor": (c_int, [_P, c_int, c_int, c_int, c_int] + [_P] * 6 + [_P, c_size_t, c_int, _P]),
    "lqp_boxqp_backward_fp": (c_int, [_P, c_int, c_int, c_int, c_int] + [_P] * 9 + [c_int, c_double, _P] + [_P] * 6 +
                              [ctypes.POINTER(ctypes.c_int32), _P, c_size_t, c_int, _P]),
    "lqp_boxqp_backward_kkt": (c_int, [_P, c_int, c_int, c_int, c_int] + [_P] * 8 + [_P] * 6 +
                               [ctypes.POINTER(ctypes.c_int32), _P, c_size_t, c_int, _P]),
    "lqp_lu_factor_workspace_bytes": (c_size_t, [c_int] * 3),
    "lqp_lu_factor_batched": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P, _P, c_size_t]),
    "lqp_lu_solve_workspace_bytes": (c_size_t, [c_int] * 3),
    "lqp_lu_solve_batched": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_size_t]),
    "lqp_lu_packed_bytes": (c_size_t, [c_int] * 3),
    "lqp_lu_pack": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P]),
    "lqp_lu_solve_packed": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P]),
    "lqp_spd_inverse_workspace_bytes": (c_size_t, [c_int] * 3),
    "lqp_spd_inverse_batched": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P, _P, c_size_t]),
    "lqp_kkt_solve_workspace_bytes": (c_size_t, [c_int] * 4),
    "lqp_kkt_solve": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P,
                              ctypes.POINTER(ctypes.c_int32), _P, c_size_t]),
    "lqp_qp_outer_grads": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P]),
}

_lock = threading.Lock()
_lib = None


HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
               "-mllvm", "-pragma-unroll-threshold=200000",      # the 32-column panel loops must fully unroll
               "-fno-slp-vectorize"]    # SLP packing of f32 ops (v_pk_*) blows up register pressure in the LU panel


def build_library(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> csrc/liblqp_amd.so (cross-compiles without a GPU).

    Default: the split build -- the host translation unit (csrc/lqp_amd.hip with `extern template` declarations of every
    kernel instance, csrc/split/lqp_extern.inc) and one translation unit per group of kernel instances
    (csrc/split/lqp_tu_*.hip), compiled in parallel and linked; same headers and flags, hence the same code per kernel
    as the single-source build (LQP_UNITY_BUILD=1: one hipcc command, ~3.5 minutes of serial code generation).  The
    lists are generated from a built library's kernel names by tools/gen_split_build.py."""
    split_dir = os.path.join(CSRC, "split")
    tus = sorted(f for f in os.listdir(split_dir) if f.endswith(".hip")) if os.path.isdir(split_dir) else []
    unity = bool(os.environ.get("LQP_UNITY_BUILD")) or not tus
    srcs = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(os.path.dirname(HERE), "include", "lqp_amd.h")]
    if not unity:
        srcs += [os.path.join(split_dir, f) for f in tus] + [os.path.join(split_dir, "lqp_extern.inc")]
    newest = max(os.path.getmtime(s) for s in srcs if os.path.exists(s))
    if not force and os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= newest:
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("LQP_EXTRA_FLAGS", "").split()       # (A/B builds: LQP_LIB=<other .so> LQP_EXTRA_FLAGS="-D...")
    if unity:
        cmd = [hipcc] + HIPCC_FLAGS + extra + ["-shared", "-o", LIB_PATH, os.path.join(CSRC, "lqp_amd.hip")]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=CSRC)
        return LIB_PATH
    from concurrent.futures import ThreadPoolExecutor
    objdir = os.path.join(CSRC, "build" if LIB_PATH.endswith("liblqp_amd.so") else "build_" + os.path.basename(LIB_PATH)[:-3])
    os.makedirs(objdir, exist_ok=True)
    jobs = [(os.path.join(CSRC, "lqp_amd.hip"), os.path.join(objdir, "lqp_amd.o"), ["-DLQP_SPLIT_BUILD"])]
    jobs += [(os.path.join(split_dir, f), os.path.join(objdir, f[:-4] + ".o"), []) for f in tus]

    def compile_one(job):
        src, obj, extra_tu = job
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= newest:
            return obj
        cmd = [hipcc] + HIPCC_FLAGS + extra + extra_tu + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=CSRC)
        return obj

    workers = int(os.environ.get("LQP_BUILD_JOBS", str(min(len(jobs), os.cpu_count() or 1))))
    with ThreadPoolExecutor(max_workers=max(1, workers)) as pool:
        objs = list(pool.map(compile_one, jobs))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    return LIB_PATH


def load():
    """Load the library (once).  Raises RuntimeError if it is missing -- no fallback."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"lqp_py_amd: HIP library not built ({LIB_PATH}); run `python -c 'import __graft_entry__ as g; "
                "g.build()'` (or lqp_py_amd._lib.build_library()). There is no CPU fallback.")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)          # AttributeError if the symbol is missing
            fn.restype, fn.argtypes = res, args
        if lib.lqp_abi_version() != ABI_VERSION:
            raise RuntimeError("lqp_py_amd: ABI version mismatch")
        _lib = lib
        return lib


def profile(enable=None, reset=False):
    """Per-kernel-class device times (ms) and launch counts since the last reset."""
    lib = load()
    if reset:
        lib.lqp_profile_reset()
    if enable is not None:
        lib.lqp_profile_enable(1 if enable else 0)
        return None
    n = lib.lqp_profile_classes()
    ms = (c_double * n)()
    cnt = (ctypes.c_longlong * n)()
    check(lib.lqp_profile_get(ms, cnt, n), "profile_get")
    return {lib.lqp_profile_class_name(i).decode(): (ms[i], cnt[i]) for i in range(n)}


def check(status, what, extra=""):
    if status == 0:
        return
    msg = STATUS.get(status, f"status {status}")
    raise RuntimeError(f"lqp_py_amd.{what}: {msg}{extra}")


def dtype_code(t):
    if t.dtype == torch.float32:
        return LQP_F32
    if t.dtype == torch.float64:
        return LQP_F64
    raise TypeError(f"lqp_py_amd supports float32/float64 tensors, got {t.dtype}")


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("lqp_py_amd: inputs must live on the GPU (HIP device); there is no CPU path")


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def current_stream_handle(device):
    """The raw handle of torch's current stream on `device` (an int).  torch.cuda.current_stream builds a Stream object around
    it (~2.5 us, in front of the first launch of every call); the C hook underneath returns the number itself."""
    if _raw_stream is not None and device.index is not None:
        return _raw_stream(device.index)
    return torch.cuda.current_stream(device).cuda_stream


def stream_ptr(device):
    return ctypes.c_void_p(current_stream_handle(device))


# ---- deferred error reporting for calls that did not synchronise with the host -----------------
class _Pending:
    """status / LU-info words of an un-synchronised call, copied to pinned memory on the call's stream"""
    __slots__ = ("what", "event", "status", "info", "bounds_check", "keep", "flags")

    def __init__(self, what, event, status, info, bounds_check=None):
        self.what, self.event, self.status, self.info, self.bounds_check = what, event, status, info, bounds_check


_pending = []
_pending_lock = threading.Lock()


_pinned_free = {}        # number of int32 words -> pinned host buffers waiting for re-use (cudaHostAlloc costs ~50 us)
_pinned_quarantine = []  # report buffers a kernel that is still queued may write (a prefactored backward that was dropped)


_pinned_lock = threading.Lock()      # (the forward's thread and the autograd thread both allocate and release report buffers)


def pinned_release(report):
    """A report buffer goes back to the pool -- at once when nothing can write it any more (every word has arrived: a kernel
    stores each word exactly once, the library set them to -1 before), otherwise when a later allocation finds it complete."""
    with _pinned_lock:
        if bool((report != -1).all()):
            _pinned_free.setdefault(report.numel(), []).append(report)
        else:
            _pinned_quarantine.append(report)



def _pinned(words):
    """One pinned int32 buffer of `words` words.  An empty pool is refilled SIXTEEN buffers at a time from one pinned
    allocation (a pipelined loop keeps a dozen reports in flight before the first one comes back: sixteen host allocations of
    ~30-100 us each used to sit in the first steps on fresh tensors)."""
    with _pinned_lock:
        if _pinned_quarantine:
            done = [r for r in _pinned_quarantine if bool((r != -1).all())]
            _pinned_quarantine[:] = [r for r in _pinned_quarantine if not any(r is d for d in done)]
            for rep in done:
                _pinned_free.setdefault(rep.numel(), []).append(rep)
        pool = _pinned_free.setdefault(words, [])
        if not pool:
            stride = (words + 15) // 16 * 16                      # (64-byte aligned slices)
            slab = torch.empty(16 * stride, dtype=torch.int32, pin_memory=True)
            pool.extend(slab[i * stride:i * stride + words] for i in range(16))
        return pool.pop()


ST_WORDS = 16                # status block, include/lqp_amd.h (lqp_boxqp_ctrl.host_report)
RP_LB, RP_UB, RP_TIMEOUT, RP_NOTSPD = 1, 2, 4, 8


def host_report(words):
    """Pinned int32 buffer the kernels of an un-synchronised call report into (no device-to-host copy: the forward's / the
    backward's last kernel stores the words straight into host memory).  Word 0 starts as -1 = "nothing arrived"."""
    return _pinned(words)        # (the library sets every word to -1 before its first launch)


def defer_check(what, device, report, B, forward, bounds_check=None):
    """Queue the report of an un-synchronised call: `report` is the pinned buffer given to the library as host_report
    (forward: 16 status words | B info words | B flag words; backward: B info words).
    bounds_check = (assumed, control, mutate, remember): the call was enqueued ASSUMING that the batch holds some /
    no finite bound; status words 12 / 13 hold what the setup kernel found."""
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(device))
    with _pending_lock:
        p = _Pending(what, ev, report[:ST_WORDS] if forward else None,
                     report[ST_WORDS:ST_WORDS + B] if forward else report[:B], bounds_check)
        p.keep = (report,)
        p.flags = report[ST_WORDS + B:ST_WORDS + 2 * B] if forward else None
        _pending.append(p)
        backlog = len(_pending)
    if backlog > 64:
        poll_errors(block=True)


def poll_errors(block=False):
    """Raise the first error of an earlier un-synchronised call whose results have arrived.
    block=True waits for all of them (``lqp_py_amd.synchronize()``)."""
    while True:
        with _pending_lock:
            if not _pending:
                return
            p = _pending[0]
            if not block and not p.event.query():
                return
            _pending.pop(0)
        p.event.synchronize()
        rep = p.keep[0].numpy()
        if int(rep[0]) == -1:               # (-1: the library's "nothing arrived"; every stored word is >= 0 or -7)
            raise RuntimeError(f"lqp_py_amd.{p.what}: the call's report never arrived in host memory")
        info = p.info.numpy()
        if (info == -7).any():              # a workgroup of a shared LU (two per matrix / wide) waited for its partner in vain
            for t in getattr(p, "keep", ()):
                _pinned_free.setdefault(t.numel(), []).append(t)
            raise RuntimeError(f"lqp_py_amd.{p.what} (reported late: the call did not synchronise): a factorisation shared between "
                               f"workgroups timed out waiting for its partner (batch index {int((info == -7).nonzero()[0][0])}; the "
                               "CUs were held by somebody else); the outputs are not valid.  Repeat with control['sync']=True, which "
                               "falls back to one workgroup per matrix by itself, or pass reserved2 bit 1 (lqp_boxqp_ctrl) for that schedule")
        bad = info.nonzero()[0]
        flags = int(numpy.bitwise_or.reduce(p.flags.numpy())) if p.flags is not None else 0
        status7 = (int(p.status[7]) or (flags & RP_NOTSPD)) if p.status is not None else 0
        status5 = (int(p.status[5]) or (flags & RP_TIMEOUT)) if p.status is not None else 0
        seen_words = (int(p.status[12]), int(p.status[13])) if p.status is not None else (1, 1)
        for t in getattr(p, "keep", ()):            # (values are read: the pinned buffers can serve the next call)
            _pinned_free.setdefault(t.numel(), []).append(t)
        if p.bounds_check is not None and p.status is not None:
            assumed, control, mutate, remember = p.bounds_check
            seen = bool(seen_words[0] or seen_words[1])
            remember(control, seen)
            if seen != assumed:
                if mutate and not seen:
                    control['rho'] = 0      # the reference layer's dict side effect (:37-38), applied late
                raise RuntimeError(
                    f"lqp_py_amd.{p.what} (reported late: the call did not synchronise): the batch held "
                    f"{'a' if seen else 'NO'} finite bound while the previous solve with this control "
                    f"{'had none' if seen else 'had some'}; the reference switches between its ADMM loop and the "
                    "rho = 0 one-shot solve on that (:157-158), and this call was enqueued for the other one -- its outputs "
                    "are not the reference's.  Repeat the call (the layer now assumes what it saw), or pass "
                    "control['sync']=True, which repeats by itself")
        if status7:
            raise RuntimeError(f"lqp_py_amd.{p.what} (reported late: the call did not synchronise): Q + rho I is not "
                               f"positive definite in float32 (batch index {int(bad[0]) if bad.size else -1}); the "
                               f"symmetric-inverse x-update does not apply: pass control['linsolve']='lu' (or "
                               f"control['sync']=True, which falls back by itself)")
        if bad.size:
            raise RuntimeError(f"lqp_py_amd.{p.what} (reported late: the call did not synchronise): LU hit an exactly "
                               f"zero pivot for batch index {int(bad[0])}; the matrix is singular")
        if status5:
            raise RuntimeError(f"lqp_py_amd.{p.what}: in-kernel grid barrier timed out")


_ws_cache = {}
_ws_uses = {}
_ws_lock = threading.Lock()


class _NoSwitch:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_no_switch = _NoSwitch()


def on_device(device):
    """`with torch.cuda.device(device)` -- skipped (it costs ~4 us around every call) when `device` is already current"""
    return _no_switch if torch.cuda.current_device() == device.index else torch.cuda.device(device)


def workspace(device, nbytes, tag, stream=None):
    """Reusable device scratch buffer per (device, STREAM, tag); grows monotonically.  Kernels of one stream are
    ordered, so re-using the buffer call after call is safe; two streams (pipelined layers, threads) never share
    one -- a persistent loop of one stream would otherwise read factors another stream is overwriting."""
    key = (device.index, current_stream_handle(device) if stream is None else stream, tag)
    with _ws_lock:
        buf = _ws_cache.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
            _ws_cache[key] = buf
        _ws_uses[key] = _ws_uses.get(key, 0) + 1
        return buf


def workspace_uses(device, tag, stream):
    """How often `workspace` has handed out the buffer of (device, stream, tag): somebody who left state in it (the
    backward's factorisation made ahead of the cotangent) compares the count to know whether it is still his."""
    with _ws_lock:
        return _ws_uses.get((device.index, stream, tag), 0)


def workspace_touch(device, tag, stream):
    """Count a WRITE to the buffer of (device, stream, tag) that did not go through `workspace`: whoever left state there
    before (see workspace_uses) must find the count changed."""
    with _ws_lock:
        key = (device.index, stream, tag)
        _ws_uses[key] = _ws_uses.get(key, 0) + 1


def release_workspaces(device=None):
    """Drop the cached scratch buffers (all devices, or one).  A long-lived process that used many streams holds one
    forward and one backward workspace per (device, stream); nothing else ever frees them.  Only call this when no
    solve is in flight on the streams concerned (``lqp_py_amd.synchronize()`` first)."""
    with _ws_lock:
        for key in [k for k in _ws_cache if device is None or k[0] == torch.device(device).index]:
            del _ws_cache[key]


def norm(t, dtype):
    """contiguous, right dtype -- the tensor itself when it already is (only its address is used)"""
    if t is None or (t.dtype == dtype and t.is_contiguous()):
        return t
    t = t.detach()
    if t.dtype != dtype:
        t = t.to(dtype)
    return t if t.is_contiguous() else t.contiguous()


def c(t):
    """contiguous, detached view of a tensor (None passes through)"""
    return None if t is None else t.detach().contiguous()
