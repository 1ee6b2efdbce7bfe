#!/usr/bin/env python3
"""Headline benchmark: QPs/sec forward+backward of the box-QP ADMM layer.

Workload (BASELINE.json configs[2], the configuration the metric is quoted
on): batch=128 per GPU, dz=500, one equality constraint (A = ones), random
box bounds, fp32, eps_abs = eps_rel = 1e-5, default box_qp_control
(scale=True, adaptive_rho=True, rho=None); synthetic inputs drawn exactly like
experiments/utils.py:41-61 of the reference, resident in HBM before timing.
One step = SolveBoxQP forward + x.backward(ones) (experiments/experiment_1.py:
69-78) on one batch, plus -- for N > 1 -- the single all-gather of x.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

B_PER_GPU, N_X, N_EQ = 128, 500, 1
TOL = 1e-5
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 matrix peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=B_PER_GPU, help="QPs per GPU")
    ap.add_argument("--n", type=int, default=N_X)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--linsolve", choices=["auto", "lu", "spd"], default="auto",
                    help="x-update of the forward solve (control['linsolve']); lu = the reference's cached pivoted LU")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the forward-only extras (configs 2 and 4)")
    ap.add_argument("--cpu-reps", type=int, default=5)
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--sync", action="store_true", help="layer calls wait for the GPU (reference-style error timing)")
    return ap.parse_args()


def algorithmic_bytes(es, n, m, iters, n_refactor, linsolve, scale=True):
    """Per-QP algorithmic bytes of forward / backward / the loop kernel alone.

    linsolve 1 (pivoted LU, the reference's algorithm): SURVEY.md 8(d) -- every x-update streams the N x N
    factor; the convergence check no longer reads Q (KKT identity), so its C n^2 term is gone.
    linsolve 2 (symmetric inverse): every x-update is a symmetric product with H = (Qs + rho I)^-1 (corrected for
    the equality rows), whose lower triangle n(n+1)/2 is all that has to move; the factorisation reads the lower
    triangle of Qs and writes that of H."""
    N = n + m
    I = iters + 1
    S = 1 if scale else 0
    if linsolve == 2:
        tri = n * (n + 1) // 2
        loop = es * I * tri
        fwd = es * (2 * S * n * n + (1 + n_refactor) * 2 * tri) + loop
    else:
        loop = es * I * N * N
        fwd = es * (2 * S * n * n + n * n + N * N + 2 * N * N + n_refactor * (n * n + 3 * N * N)) + loop
    bwd = es * (3 * n * n + 4 * N * N)
    return fwd, bwd, loop, es * I * N * N


LOOP_KERNEL = {1: "lqp::k_admm_loop<float, true, false, 1024, false>",
               2: "lqp::k_admm_loop<float, true, false, 1024, true>"}
TRAFFIC_FILE = "profiles/r01_i_traffic.json"


def measured_traffic(kernel, mode, B, n):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (FETCH_SIZE / WRITE_SIZE collected in separate runs, FETCH_SIZE doubled per the gfx950 note in
    MI355X_MICROARCH.md).  Only valid for the configuration it was measured on; otherwise null."""
    try:
        d = json.load(open(os.path.join(REPO, TRAFFIC_FILE)))
        if d.get("launch_mode") == mode and B == B_PER_GPU and n == N_X:
            return d["kernels"][kernel]["hbm_bytes_per_launch_corrected"], TRAFFIC_FILE + " (rocprofv3 --pmc)"
    except Exception:
        pass
    return None, None


def cpu_baseline_worker(args):
    """(child process, OMP/MKL threads pinned by the environment before torch was imported)"""
    from oracle import boxqp_oracle as O
    B, n = args.batch, args.n
    Q, p, A, b, lb, ub = O.create_qp_data(n, B, seed=0)
    ctl = O.make_control(eps_abs=TOL, eps_rel=TOL)
    ones = torch.ones(B, n, 1)

    def once():
        sol = O.solve_box_qp(Q, p, A, b, lb, ub, dict(ctl))
        O.solve_box_qp_grad(ones, sol["x"], sol["u"], sol["lams"], sol["nus"], Q, A, lb, ub, sol["rho"])

    once()
    t0 = time.perf_counter()
    for _ in range(args.cpu_reps):
        once()
    dt = (time.perf_counter() - t0) / args.cpu_reps
    print(json.dumps({"value": B / dt, "unit": "QPs/sec", "cores": torch.get_num_threads(), "kind": "port",
                      "sample": f"{args.cpu_reps} x (forward+backward of one batch={B} dz={n} m=1 tol=1e-5), "
                                f"{dt:.2f} s each, torch {torch.__version__} CPU, os.cpu_count()={os.cpu_count()}"}))


def cpu_baseline(args):
    """The oracle (CPU restatement of the reference, torch CPU) timed on this host's cores, in a child
    process whose OMP/MKL thread count is the box's CPU share for one GPU (16).  (Changing the thread
    count of an already-imported torch breaks MKL's batched getrf on this image, and all 256 hardware
    threads make small batched LAPACK slower, not faster.)"""
    import subprocess
    threads = str(min(16, os.cpu_count() or 1))
    env = dict(os.environ, OMP_NUM_THREADS=threads, MKL_NUM_THREADS=threads, HIP_VISIBLE_DEVICES="")
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", "--batch", str(args.batch),
           "--n", str(args.n), "--cpu-reps", str(args.cpu_reps)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    for line in reversed(res.stdout.strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    return {"value": None, "error": (res.stderr or res.stdout)[-300:]}


def main():
    args = parse()
    if args.cpu_baseline_worker:
        return cpu_baseline_worker(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback of the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    import lqp_py_amd as L
    from lqp_py_amd import _lib
    from lqp_py_amd.dist import ShardedBoxQP
    from lqp_py_amd.synthetic import create_qp_data

    B, n, m = args.batch, args.n, N_EQ
    # a few distinct batches (different seeds, like the reference's per-simulation data), HBM resident
    n_sets = 3
    data = []
    for s in range(n_sets):
        Q, p, A, b, lb, ub = create_qp_data(n, B, seed=1000 * rank + s)
        data.append([t.to(dev) for t in (Q, p, A, b, lb, ub)])
    ones = torch.ones(B, n, 1, device=dev)
    control = L.box_qp_control(eps_rel=TOL, eps_abs=TOL, verbose=False, reduce='max')
    if args.linsolve != "auto":
        control['linsolve'] = args.linsolve
    control['sync'] = bool(args.sync)     # False: the pipelined training-loop mode (errors reported late, NaN on failure)
    layer = ShardedBoxQP(control) if world > 1 else None
    qp = L.SolveBoxQP(control=control)
    last = {}

    def step(i):
        Q, p, A, b, lb, ub = data[i % n_sets]
        Q = Q.detach().requires_grad_(True)          # experiment_1 differentiates w.r.t. Q and p
        p = p.detach().requires_grad_(True)
        if world > 1:
            x, x_all = layer(Q, p, A, b, lb, ub)
        else:
            x = qp(Q, p, A, b, lb, ub)
        x.backward(ones)
        last["x"] = x

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize(dev)

    for i in range(args.warmup):
        step(i)
    sync()
    # ---- timed region: exactly K steps, nothing else on the stream ----
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    sync()
    dt = time.perf_counter() - t0
    L.synchronize()                       # surface any deferred error of the un-synchronised layer calls
    # ---- the same K steps again with every library launch bracketed by HIP events on its stream:
    #      per-kernel device times for the roofline (kept out of the timed region: the event pairs
    #      cost a few microseconds per launch) ----
    _lib.profile(enable=True, reset=True)
    tp = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    sync()
    dt_prof = time.perf_counter() - tp
    prof = _lib.profile()
    _lib.profile(enable=False)
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t[0])
    ms_per_step = dt / args.steps * 1e3
    value = world * B / (dt / args.steps)

    # ---- roofline of the dominant kernel (the persistent / segmented ADMM loop) ----
    Q, p, A, b, lb, ub = data[0]
    sol = L.torch_solve_box_qp(Q, p, A, b, lb, ub, dict(control))
    st = sol["_stats"]
    es = 4
    ls = st["linsolve_used"]
    fwd_b, bwd_b, loop_b, loop_ref_b = algorithmic_bytes(es, n, m, st["iters"], st["n_factor"] - 1, ls)
    loop_ms, loop_launches = prof["admm_loop"]
    solves = args.steps
    loop_ms_per_solve = loop_ms / max(solves, 1)
    gbs = lambda nbytes: (nbytes * B) / (loop_ms_per_solve * 1e-3) / 1e9 if loop_ms_per_solve > 0 else 0.0
    achieved = gbs(loop_b)
    traffic, traffic_src = measured_traffic(LOOP_KERNEL[ls], 3 if not args.sync else st["mode_used"], B, n)
    roofline = {"bound": "hbm", "kernel": LOOP_KERNEL[ls], "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_source": traffic_src,
                "per": "the first (hot) loop launch of one forward solve of one batch",
                "algorithmic_bytes": loop_b * B, "ms": round(loop_ms_per_solve, 4),
                "launches_per_solve": loop_launches / max(solves, 1),
                "linsolve": {1: "lu", 2: "spd"}[ls],
                # the same launch priced with the REFERENCE algorithm's bytes (I * N^2 * es, the
                # cached-LU stream this kernel replaces): not a bandwidth, a like-for-like speed figure
                "reference_algorithm_bytes": loop_ref_b * B,
                "reference_algorithm_equiv_GBs": round(gbs(loop_ref_b), 1),
                "whole_step_frac_of_hbm_roofline": round(((fwd_b + bwd_b) * B / (ms_per_step * 1e-3) / 1e9) / HBM_PEAK_GBS, 4)}
    # second kernel of comparable weight on the symmetric path: the factorisation (MFMA block sweep).  Algorithmic
    # flops of inverting an SPD matrix: n^3 (Cholesky n^3/3 + inverse of the factor and product 2n^3/3).
    roofline_factor = None
    if ls == 2 and prof.get("spd_inverse", (0, 0))[1]:
        f_ms = prof["spd_inverse"][0] / max(solves, 1)
        tfl = B * float(n) ** 3 / (f_ms * 1e-3) / 1e12
        mode_t = 3 if not args.sync else st["mode_used"]
        if st.get("factor_launches", 1) > 1:
            # small batch: two workgroups per matrix, one launch per pivot step (begin | Ks steps | end)
            Ks = (n + 63) // 64
            parts = [measured_traffic(k, mode_t, B, n)[0] for k in ("lqp::k_spd_begin", "lqp::k_spd_step", "lqp::k_spd_end")]
            f_traffic = None if any(t is None for t in parts) else parts[0] + Ks * parts[1] + parts[2]
            f_kernel = f"lqp::k_spd_begin + {Ks} x lqp::k_spd_step + lqp::k_spd_end"
        else:
            f_traffic, f_kernel = measured_traffic("lqp::k_spd_inverse", mode_t, B, n)[0], "lqp::k_spd_inverse"
        roofline_factor = {"bound": "mfma", "kernel": f_kernel, "achieved": round(tfl, 2),
                           "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tfl / MFMA_F32_PEAK_TFLOPS, 4),
                           "traffic": f_traffic, "algorithmic_flops": B * float(n) ** 3, "ms": round(f_ms, 4),
                           "per": "one factorisation of the batch (all its launches)"}
    breakdown = {k: round(v[0] / max(solves, 1), 4) for k, v in prof.items() if v[1]}

    out = {"metric": "QPs/sec forward+backward, batch=128 dz=500 tol=1e-5", "value": round(value, 1),
           "unit": "QPs/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"BASELINE configs[2]: batch={B}/GPU dz={n} m={m} box+equality QP, "
                                  "ADMM forward + fixed-point backward, eps 1e-5, scale+adaptive_rho defaults",
                      "global_batch": world * B, "iters": st["iters"], "checks": st["n_check"],
                      "launch_mode": st["mode_used"], "parallelism": f"batch-sharded x{world}"},
           "roofline": roofline, "roofline_factorisation": roofline_factor, "kernel_ms_per_step": breakdown,
           "profiled_pass_ms_per_step": round(dt_prof / args.steps * 1e3, 4)}
    # "also report forward-only for configs 2 and 4" (B=128, n=100 box-only / n=1000 with the
    # equality row) -- extras next to the headline number, single GPU only, inputs drawn on the device
    if world == 1 and not args.no_other_configs and B == B_PER_GPU and n == N_X:
        extras = {}
        for name, (nn, with_eq) in {"config2_fwd_n100_box": (100, False), "config4_fwd_n1000_eq": (1000, True)}.items():
            gen = torch.Generator(device=dev).manual_seed(4242 + nn)
            Lm = torch.randn(B, 2 * nn, nn, device=dev, generator=gen)
            Qx = torch.matmul(Lm.transpose(1, 2), Lm) / (2 * nn)
            del Lm
            px = torch.randn(B, nn, 1, device=dev, generator=gen)
            Ax = torch.ones(B, 1, nn, device=dev) if with_eq else None
            bx = torch.ones(B, 1, 1, device=dev) if with_eq else None
            lbx = -(torch.rand(B, nn, 1, device=dev, generator=gen) + 1)
            ubx = torch.rand(B, nn, 1, device=dev, generator=gen) + 1
            fwd = lambda: qp(Qx, px, Ax, bx, lbx, ubx)
            for _ in range(2):
                fwd()
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                fwd()
            torch.cuda.synchronize(dev)
            dtx = (time.perf_counter() - t1) / reps
            extras[name] = {"QPs_per_sec": round(B / dtx, 1), "ms": round(dtx * 1e3, 4), "batch": B}
            del Qx
        L.synchronize()
        out["other_configs_forward_only"] = extras
    if rank == 0:
        # no number is published for a GPU; the reference's own chart for this config (6-core i7
        # CPU, images_paper/dz_500.pdf) reads 112.6 QPs/s -- quoted for orientation, vs_baseline stays null
        out["reference_published_other_hardware"] = {"value": 112.6, "unit": "QPs/sec", "hardware": "6-core i7 2.6 GHz CPU",
                                                     "source": "BASELINE.md §1 (bar chart of the reference)"}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args)
        elif not args.no_cpu_baseline:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
