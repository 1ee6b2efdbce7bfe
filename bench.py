#!/usr/bin/env python3
"""Headline benchmark: QPs/sec forward+backward of the box-QP ADMM layer.

Workload (BASELINE.json configs[2], the configuration the metric is quoted
on): batch=128 per GPU, dz=500, one equality constraint (A = ones), random
box bounds, fp32, eps_abs = eps_rel = 1e-5, default box_qp_control
(scale=True, adaptive_rho=True, rho=None); synthetic inputs drawn exactly like
experiments/utils.py:41-61 of the reference with seeds 0..9 (one batch per
seed, experiments/experiment_1.py:53-58), resident in HBM before timing.
One step = SolveBoxQP forward + x.backward(ones) (experiment_1.py:69-78) on
one batch, plus -- for N > 1 -- the single all-gather of x.

    python bench.py [--gpus N] [--steps K] [--warmup W]

With N > 1 and no torchrun environment this process touches no GPU: it starts
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`
as a child (one rank per GPU over RCCL) and exits with its code.  Rank 0
prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

B_PER_GPU, N_X, N_EQ = 128, 500, 1
TOL = 1e-5
N_SEEDS = 10                  # experiment_1.py: n_sims = 10, seed = simulation index
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 matrix peak
MFMA_F16_PEAK_TFLOPS = 2516.6 # MI355X_MICROARCH.md: dense f16 / bf16 matrix peak (v_mfma_f32_32x32x16_f16: 32 k flop in 32 cycles per SIMD)
MFMA_F64_PEAK_TFLOPS = 78.6   # MI355X: fp64 matrix = fp64 vector peak (v_mfma_f64_16x16x4_f64, 64 cycles per SIMD)
INFINITY_CACHE_BYTES = 256 * 2 ** 20
TRAFFIC_FILE = "profiles/r06_b_headline_traffic.json"
PROFILE_TAG = "r06_b"          # profiles/<tag>_<workload>_{run.json,kernel_stats.csv,traffic.json,pmc_sq_raw.json}: tools/profile_workload.sh


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=B_PER_GPU, help="QPs per GPU")
    ap.add_argument("--n", type=int, default=N_X)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--linsolve", choices=["auto", "lu", "spd"], default="auto",
                    help="x-update of the forward solve (control['linsolve']); lu = the reference's cached pivoted LU")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the extras (configs 2 and 4, LU step, sync step)")
    ap.add_argument("--cpu-reps", type=int, default=5)
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling as the metric string states it: ONE batch of --batch QPs (default 128) split "
                         "over the N GPUs (128/N per GPU: 64 / 32 / 16 at N = 2 / 4 / 8); default: weak scaling, --batch per GPU")
    ap.add_argument("--config5", action="store_true",
                    help="BASELINE configs[4]: batch=8192 dz=500 sharded over 8 GPUs = 1024 QPs per GPU (weak: 1024 per "
                         "GPU at any N)")
    ap.add_argument("--sync", action="store_true",
                    help="time the layer with its default control (calls wait for the GPU and raise at the call, like the "
                         "reference); default here: control['sync']=False, the pipelined training-loop mode")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------
# N > 1 without a torchrun environment: spawn the ranks; this parent makes no GPU call
# ---------------------------------------------------------------------------------------------------
def spawn_ranks(args):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def algorithmic_bytes(es, n, m, iters, n_refactor, linsolve, scale=True, bwd_chol=False, n_free=None):
    """Per-QP algorithmic bytes of forward / backward / the loop kernel alone.

    linsolve 1 (pivoted LU, the reference's algorithm): SURVEY.md 8(d) -- every x-update streams the N x N factor;
    the convergence check no longer reads Q (KKT identity), so its C n^2 term is gone.
    linsolve 2 (symmetric inverse): every x-update is a symmetric product with H = (Qs + rho I)^-1 (corrected for the
    equality rows), whose lower triangle n(n+1)/2 is all that has to move; the factorisation reads the lower triangle
    of Qs and writes that of H.  Backward: the LU form (3 n^2 + 4 N^2) or, when the Cholesky form ran,
    n^2 (gather of Q_FF, at most) + f(f+1) (its factor, written and read) + n^2 (Q dv) + n^2 (dQ), f = free set."""
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
    if bwd_chol:
        f = n if n_free is None else n_free
        bwd = es * (3 * n * n + f * (f + 1))
    else:
        bwd = es * (3 * n * n + 4 * N * N)
    return fwd, bwd, loop, es * I * N * N


def measured_traffic(kernel, B, n):
    """HBM-side bytes per launch of a kernel from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE collected
    in separate runs, FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md).  Only valid for the
    configuration it was measured on; otherwise null.  Kernel names are matched by prefix (template arguments vary)."""
    try:
        d = json.load(open(os.path.join(REPO, TRAFFIC_FILE)))
        if B == B_PER_GPU and n == N_X:
            for name, rec in d["kernels"].items():
                if name.startswith(kernel):
                    return rec["hbm_bytes_per_launch_corrected"], TRAFFIC_FILE + " (rocprofv3 --pmc)"
    except Exception:
        pass
    return None, None


def profile_twin(workload):
    """What the committed rocprofv3 set of a workload (tools/profile_workload.{py,sh}: the same inputs and control as the row
    of this benchmark it stands beside) says about it: the dominant kernel with its average duration, the counter traffic of
    the whole step (sum over kernels of bytes per launch x launches per step; FETCH_SIZE doubled per the gfx950 note), and --
    where matrix instructions ran -- their rate against the dense peak.  None when the set is not in the tree."""
    import csv
    base = os.path.join(REPO, "profiles", f"{PROFILE_TAG}_{workload}_")
    try:
        run = json.load(open(base + "run.json"))
        tr = json.load(open(base + "traffic.json"))["kernels"]
        sq = json.load(open(base + "pmc_sq_raw.json"))
        rows = [r for r in csv.DictReader(open(base + "kernel_stats.csv")) if r["Name"].replace("void ", "").startswith("lqp::")]
    except Exception:
        return None
    passes = run["steps"] + run["warmup"]
    step_bytes = sum(rec["hbm_bytes_per_launch_corrected"] * rec["launches"] for rec in tr.values()) / passes
    step_ns = sum(float(r["TotalDurationNs"]) for r in rows) / passes
    dom = max(rows, key=lambda r: float(r["TotalDurationNs"]))
    name = dom["Name"].replace("void ", "").split("(")[0]
    out = {"source": f"profiles/{PROFILE_TAG}_{workload}_* (rocprofv3 --kernel-trace --stats; --pmc FETCH_SIZE / WRITE_SIZE / SQ_* in separate passes)",
           "profiled_ms_per_step": run["ms_per_step"], "kernel_ms_per_step": round(step_ns / 1e6, 4),
           "traffic_bytes_per_step": int(step_bytes),
           "traffic_frac_of_hbm_peak": round(step_bytes / (step_ns * 1e-9) / 1e9 / HBM_PEAK_GBS, 4),
           "dominant_kernel": name, "dominant_kernel_avg_us": round(float(dom["AverageNs"]) / 1e3, 1),
           "dominant_kernel_share": round(float(dom["TotalDurationNs"]) / (step_ns * passes), 3),
           "dominant_kernel_traffic_bytes_per_launch": tr.get(name, {}).get("hbm_bytes_per_launch_corrected")}
    cnt = sq.get(name) or sq.get("void " + name) or {}
    g = lambda k: cnt.get(k, {}).get("mean")
    for key, peak, unit in (("SQ_INSTS_VALU_MFMA_MOPS_F32", MFMA_F32_PEAK_TFLOPS, "f32"), ("SQ_INSTS_VALU_MFMA_MOPS_F64", MFMA_F64_PEAK_TFLOPS, "f64"),
                            ("SQ_INSTS_VALU_MFMA_MOPS_F16", MFMA_F16_PEAK_TFLOPS, "f16")):
        if g(key):
            tfl = g(key) * 512 / (float(dom["AverageNs"]) * 1e-9) / 1e12          # one MOP = 512 flop (a 32x32x2 f32 instruction: 8 MOPs, 4096 flop)
            out[f"dominant_kernel_mfma_{unit}_TFLOPs_issued"] = round(tfl, 2)
            out[f"dominant_kernel_mfma_{unit}_frac_of_peak"] = round(tfl / peak, 4)
    if g("SQ_WAIT_ANY") and g("SQ_WAVE_CYCLES"):
        out["dominant_kernel_wave_cycles_parked"] = round(g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), 3)
    return out


def cpu_baseline_worker(args):
    """(child process, OMP/MKL threads pinned by the environment before torch was imported)"""
    import torch
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
                      "sample": f"{args.cpu_reps} x (forward+backward of one batch={B} dz={n} m=1 tol=1e-5, seed 0), "
                                f"{dt:.2f} s each, torch {torch.__version__} CPU, os.cpu_count()={os.cpu_count()}"}))


def cpu_baseline(args):
    """The oracle (CPU restatement of the reference, torch CPU) timed on this host's cores, in a child process whose
    OMP/MKL thread count is the box's CPU share for one GPU (16).  (Changing the thread count of an already-imported
    torch breaks MKL's batched getrf on this image, and all 256 hardware threads make small batched LAPACK slower.)"""
    threads = str(min(16, os.cpu_count() or 1))
    env = dict(os.environ, OMP_NUM_THREADS=threads, MKL_NUM_THREADS=threads, HIP_VISIBLE_DEVICES="")
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", "--batch", str(args.batch),
           "--n", str(args.n), "--cpu-reps", str(args.cpu_reps)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    for line in reversed(res.stdout.strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    return {"value": None, "error": (res.stderr or res.stdout)[-300:]}


def median(v):
    s = sorted(v)
    k = len(s)
    return s[k // 2] if k % 2 else 0.5 * (s[k // 2 - 1] + s[k // 2])


def main():
    args = parse()
    if args.cpu_baseline_worker:
        return cpu_baseline_worker(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))             # (no GPU call was made by this process)

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); reporting n_gpus={world}", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback of the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
        world = dist.get_world_size()           # the RCCL world actually formed
    import lqp_py_amd as L
    from lqp_py_amd import _lib
    from lqp_py_amd.dist import ShardedBoxQP
    from lqp_py_amd.solve_box_qp_admm_torch import last_forward_status
    from lqp_py_amd.synthetic import create_qp_data

    from lqp_py_amd.dist import shard_slice
    n, m = args.n, N_EQ
    if args.config5:
        args.batch = 1024
    B_total = args.batch if args.strong else args.batch * world
    if args.strong:
        # ONE batch of B_total problems, drawn identically on every rank (seeds 0..9 as in experiment_1.py), each rank
        # keeping its contiguous slice -- the batch the single-GPU line solves, cut across the GPUs
        lo, hi = shard_slice(B_total, rank, world)
        shard_sizes = [shard_slice(B_total, r, world)[1] - shard_slice(B_total, r, world)[0] for r in range(world)]
    else:
        lo, hi = 0, args.batch
        shard_sizes = [args.batch] * world
    B = hi - lo
    # one batch per simulation seed (weak scaling: rank r draws seeds 100 r + 0..9), HBM resident
    data = []
    n_seeds_here = 3 if args.config5 else N_SEEDS          # (1 GB of Q per batch at B = 1024)
    for s in range(n_seeds_here):
        Q, p, A, b, lb, ub = create_qp_data(n, B_total if args.strong else B, seed=s if args.strong else 100 * rank + s)
        data.append([t[lo:hi].contiguous().to(dev) for t in (Q, p, A, b, lb, ub)])
    ones = torch.ones(B, n, 1, device=dev)

    def make_layer(linsolve, sync):
        control = L.box_qp_control(eps_rel=TOL, eps_abs=TOL, verbose=False, reduce='max')
        if linsolve != "auto":
            control['linsolve'] = linsolve
        control['sync'] = bool(sync)     # False: the pipelined training-loop mode (errors reported late, NaN on failure)
        return control, (ShardedBoxQP(control, shard_sizes=shard_sizes) if world > 1 else L.SolveBoxQP(control=control))

    control, layer = make_layer(args.linsolve, args.sync)

    def forward(i, lay=None):
        Q, p, A, b, lb, ub = data[i % len(data)]
        Q = Q.detach().requires_grad_(True)          # experiment_1 differentiates w.r.t. Q and p
        p = p.detach().requires_grad_(True)
        out = (lay or layer)(Q, p, A, b, lb, ub)
        return out[0] if world > 1 else out

    def step(i, lay=None):
        forward(i, lay).backward(ones)

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize(dev)

    def timed(k_steps, lay=None, first=0):
        sync()
        t0 = time.perf_counter()
        for i in range(k_steps):
            step(first + i, lay)
        sync()
        return time.perf_counter() - t0

    # ---- the first pass over the ten synthetic batches, untimed: library / workspace start-up rides on the very first
    #      step; the other nine are what a step on never-seen tensors costs (config.first_use_ms_per_step).  Nothing about
    #      a tensor is remembered between calls: whether any bound is finite (reference :33-38, a host decision there) is
    #      found by the setup kernel from the data of every call. ----
    for _ in range(30):                   # (batch 0 alone: clocks, allocator and the pinned report pool of a fresh process --
        step(0)                           #  tools/gpu_first_use.py: with ONE such step the nine first-use steps below are 6-8 %
    sync()                                #  slower than the steady state, with thirty they equal it)
    # nine steps are 6.6 ms: one hiccup of the box is 5 % of that.  Three rounds, each on FRESHLY ALLOCATED copies of batches
    # 2..10 (never seen by the layer; the copies are made outside the clock), the median round is reported.
    rounds = []
    n_fresh = len(data) - 1
    for _r in range(1 if args.config5 else 3):
        fresh = [[t.clone() for t in data[i]] for i in range(1, len(data))]
        sync()
        t_first = time.perf_counter()
        for d_ in fresh:
            Q_, p_ = d_[0].requires_grad_(True), d_[1].requires_grad_(True)
            out_ = layer(Q_, p_, *d_[2:])
            (out_[0] if world > 1 else out_).backward(ones)
        sync()
        rounds.append((time.perf_counter() - t_first) / n_fresh * 1e3)
        del fresh, d_, Q_, p_, out_
    first_use_ms = sorted(rounds)[len(rounds) // 2]
    for i in range(args.warmup):
        step(i)
    # ---- timed region: exactly K steps, nothing else on the stream ----
    dt = timed(args.steps, first=args.warmup)
    L.synchronize()                       # surface any deferred error of the un-synchronised layer calls
    st_timed = last_forward_status(dev)   # iteration count / checks of the LAST timed forward, read from the device
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t[0])
    ms_per_step = dt / args.steps * 1e3
    value = B_total / (dt / args.steps)

    # ---- the reference's protocol (experiment_1.py:53-94): one simulation per seed 0..9, forward and
    #      backward timed separately (device events on the launch stream), medians over the simulations ----
    ev = lambda: torch.cuda.Event(enable_timing=True)
    t_fwd, t_bwd = [], []
    for s in range(len(data)):
        # fresh tensors for every simulation, as experiment_1.py:58 draws fresh data (copies made outside the timed phases)
        Qs_, ps_, As_, bs_, lbs_, ubs_ = (t.clone() for t in data[s])
        Qs_.requires_grad_(True)
        ps_.requires_grad_(True)
        e0, e1, e2 = ev(), ev(), ev()
        torch.cuda.synchronize(dev)
        e0.record()
        out_ = layer(Qs_, ps_, As_, bs_, lbs_, ubs_)
        x = out_[0] if world > 1 else out_
        e1.record()
        x.backward(ones)
        e2.record()
        torch.cuda.synchronize(dev)
        t_fwd.append(e0.elapsed_time(e1))
        t_bwd.append(e1.elapsed_time(e2))
        del Qs_, ps_, As_, bs_, lbs_, ubs_, x, out_
    L.synchronize()
    protocol = {"simulations": len(data), "seeds": f"{100 * rank}..{100 * rank + len(data) - 1}",
                "median_forward_ms": round(median(t_fwd), 4), "median_backward_ms": round(median(t_bwd), 4),
                "QPs_per_sec_median": round(B / ((median(t_fwd) + median(t_bwd)) * 1e-3), 1),
                "QPs_per_sec_mean": round(B / ((sum(t_fwd) + sum(t_bwd)) / len(data) * 1e-3), 1),
                "timing": "device events around each phase, one isolated simulation at a time (per GPU), every "
                          "simulation on freshly allocated copies of its inputs",
                "is": "SURVEY 8(d)'s metric: B / (median t_forward + median t_backward) over the 10 simulations"}

    # ---- the same K steps again with every library launch bracketed by HIP events on its stream: per-kernel
    #      device times for the roofline (kept out of the timed region: the event pairs cost microseconds) ----
    _lib.profile(enable=True, reset=True)
    dt_prof = timed(args.steps, first=args.warmup)
    prof = _lib.profile()
    _lib.profile(enable=False)
    solves = max(args.steps, 1)
    per_solve = lambda cls: prof.get(cls, (0.0, 0))[0] / solves
    breakdown = {k: round(v[0] / solves, 4) for k, v in prof.items() if v[1]}

    # ---- roofline objects ----
    es = 4
    ls = st_timed["linsolve_used"]
    iters = st_timed["iters"]
    fwd_b, bwd_b, loop_b, loop_ref_b = algorithmic_bytes(es, n, m, iters, st_timed["n_factor"] - 1, ls,
                                                         bwd_chol=(ls == 2 and prof.get("bwd_cholesky", (0, 0))[1] > 0),
                                                         n_free=int(0.63 * n))
    loop_ms = per_solve("admm_loop")
    gbs = lambda nbytes, ms: (nbytes * B) / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    split = st_timed["loop_workgroups_per_qp"] >= 2
    loop_kernel = ("lqp::k_admm_loop_split" if split else
                   "lqp::k_admm_loop<float, true, false, 1024, true>" if ls == 2 else "lqp::k_admm_loop<float, true, false, 1024, false>")
    traffic, traffic_src = measured_traffic(loop_kernel, B, n)
    working_set = B * (n * (n + 1) // 2 if ls == 2 else (n + m) ** 2) * es
    roof_loop = {"bound": "hbm", "kernel": loop_kernel, "achieved": round(gbs(loop_b, loop_ms), 1), "peak": HBM_PEAK_GBS,
                 "unit": "GB/s", "frac": round(gbs(loop_b, loop_ms) / HBM_PEAK_GBS, 4),
                 "traffic": traffic, "traffic_source": traffic_src,
                 "frac_of_peak_by_traffic": None if traffic is None or loop_ms <= 0 else round(traffic / (loop_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                 "per": "the first (hot) loop launch of one forward solve of one batch",
                 "algorithmic_bytes": loop_b * B, "ms": round(loop_ms, 4),
                 "launches_per_solve": prof.get("admm_loop", (0, 0))[1] / solves, "linsolve": {1: "lu", 2: "spd"}[ls],
                 "iterations": iters + 1,
                 "residency": ("two workgroups per QP, every block of H in registers: the algorithmic bytes never leave the "
                               "chip after the first read, so the algorithmic rate is not a memory bandwidth (frac may exceed 1)"
                               if split else
                               f"working set {working_set / 2**20:.0f} MiB vs {INFINITY_CACHE_BYTES // 2**20} MiB Infinity Cache: "
                               "a MALL-resident rate, not DRAM bandwidth" if working_set < INFINITY_CACHE_BYTES else "streamed from HBM"),
                 # the same launch priced with the REFERENCE algorithm's bytes (I * N^2 * es, the cached-LU
                 # stream this kernel replaces): not a bandwidth, a like-for-like speed figure
                 "reference_algorithm_bytes": loop_ref_b * B,
                 "reference_algorithm_equiv_GBs": round(gbs(loop_ref_b, loop_ms), 1)}
    # factorisation of the symmetric path (MFMA block sweep).  Algorithmic flops of inverting an SPD matrix: n^3
    # (Cholesky n^3/3 + inverse of the factor and product 2n^3/3).
    roof_factor = None
    if ls == 2 and prof.get("spd_inverse", (0, 0))[1]:
        f_ms = per_solve("spd_inverse")
        tfl = B * float(n) ** 3 / (f_ms * 1e-3) / 1e12
        Ks = (n + 63) // 64
        if st_timed["factor_launches"] == 3:
            # small batch: blocks built by k_spd_prep (the one pass over Q: + column maxima, symmetry verdict; k_spd_begin
            # when auto-scaling is off), every pivot step in ONE launch with the matrix in the registers of two workgroups
            # (k_spd_resident); the equality correction rides in the loop kernel
            # (round 4: the resident sweep makes the pass over Q itself -- no k_spd_prep launch; a traffic file taken on an
            #  older schedule still lists the kernel in front of it)
            names = [k for k in ("lqp::k_spd_prep", "lqp::k_spd_begin") if measured_traffic(k, B, n)[0] is not None][:1]
            names.append("lqp::k_spd_resident")
            parts = [measured_traffic(k, B, n)[0] for k in names]
            f_traffic = None if any(t is None for t in parts) else sum(parts)
            f_kernel = " + ".join(names)
        elif st_timed["factor_launches"] > 1:
            parts = [measured_traffic(k, B, n)[0] for k in ("lqp::k_spd_begin", "lqp::k_spd_step", "lqp::k_spd_end")]
            f_traffic = None if any(t is None for t in parts) else parts[0] + Ks * parts[1] + parts[2]
            f_kernel = f"lqp::k_spd_begin + {Ks} x lqp::k_spd_step + lqp::k_spd_end"
        else:
            f_traffic, f_kernel = measured_traffic("lqp::k_spd_inverse", B, n)[0], "lqp::k_spd_inverse"
        roof_factor = {"bound": "mfma", "kernel": f_kernel, "achieved": round(tfl, 2), "peak": MFMA_F32_PEAK_TFLOPS,
                       "unit": "TFLOP/s", "frac": round(tfl / MFMA_F32_PEAK_TFLOPS, 4), "traffic": f_traffic,
                       "algorithmic_flops": B * float(n) ** 3, "ms": round(f_ms, 4),
                       "minimum_traffic_bytes": 2 * B * (n * (n + 1) // 2) * es,
                       "per": "one factorisation of the batch (all its launches)",
                       "peak_is": "the float32 matrix peak (the arithmetic the path computes in: n^3 float32 flops per matrix)"}
        if st_timed["factor_launches"] == 3 and os.environ.get("LQP_SPD_F16", "1") != "0":
            # round 6: the sweep's panel products run on the float16 matrix pipe, every float32 operand as two halves
            # (csrc/lqp_f16x2.hpp): three matrix instructions per product term -- priced against THAT pipe's peak too
            roof_factor["matrix_pipe"] = ("f16, two-half float32 operands: three v_mfma_f32_32x32x16_f16 per 16-deep slice "
                                          "(issued flops = 3 x algorithmic for the panel products)")
            roof_factor["frac_of_f16_peak_issued"] = round(3 * tfl / MFMA_F16_PEAK_TFLOPS, 4)
    # `roofline` describes the kernel (group) that takes the largest share of the step
    dominant_is_factor = roof_factor is not None and roof_factor["ms"] > loop_ms
    roofline = dict(roof_factor if dominant_is_factor else roof_loop)
    step_bytes = (fwd_b + bwd_b) * B
    roofline["whole_step_frac_of_hbm_roofline"] = round((step_bytes / (ms_per_step * 1e-3) / 1e9) / HBM_PEAK_GBS, 4)
    roofline["whole_step_algorithmic_bytes"] = step_bytes
    roofline["whole_step_algorithmic_bytes_note"] = ("SURVEY 8(d)-style accounting of the path that RAN (forward: symmetric inverse; backward: "
                                                     + ("Cholesky form, free set taken as 0.63 n -- an estimate" if (ls == 2 and prof.get("bwd_cholesky", (0, 0))[1] > 0) else "LU form")
                                                     + "); most of these bytes stay in registers / LDS, so this is NOT a memory bandwidth -- "
                                                     "whole_step_traffic_bytes below is what the counters saw move")
    twin = profile_twin("headline") if (B == B_PER_GPU and n == N_X and not args.strong and not args.config5) else None
    roofline["whole_step_traffic_bytes"] = None if twin is None else twin["traffic_bytes_per_step"]
    roofline["whole_step_frac_of_hbm_peak_by_traffic"] = (None if twin is None else
                                                          round(twin["traffic_bytes_per_step"] / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
    roofline["whole_step_traffic_source"] = None if twin is None else twin["source"]

    workload = (f"BASELINE configs[4]: batch={B_total} dz={n} m={m} sharded over {world} GPU(s), {B}/GPU" if args.config5 else
                f"BASELINE configs[2], strong scaling: ONE batch={B_total} dz={n} m={m} split over {world} GPU(s), {B} on rank 0"
                if args.strong else f"BASELINE configs[2]: batch={B}/GPU dz={n} m={m}")
    out = {"metric": "QPs/sec forward+backward, batch=128 dz=500 tol=1e-5", "value": round(value, 1),
           "unit": "QPs/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong" if args.strong else "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": workload + " box+equality QP, ADMM forward + fixed-point backward, eps 1e-5, "
                                  "scale+adaptive_rho defaults",
                      "global_batch": B_total, "shard_sizes": shard_sizes,
                      "seeds": f"{len(data)} batches, seeds 0..{len(data) - 1}" + (" (one batch, cut)" if args.strong else " per rank") + ", cycled",
                      "first_use_ms_per_step": round(first_use_ms, 4),
                      "first_use": "step time over freshly allocated copies of batches 2..10 on their FIRST pass through the layer (median of three such rounds of nine steps, after thirty warm-up steps on batch 1 alone; nothing is cached per tensor: the bound flags of :33-38 are found on the device in every call)",
                      "iters": iters, "checks": st_timed["n_check"], "launch_mode": st_timed["mode_used"],
                      "stats_source": "device status block of the last timed forward",
                      "sync": bool(args.sync), "linsolve": {1: "lu", 2: "spd"}[ls],
                      "loop_workgroups_per_qp": st_timed["loop_workgroups_per_qp"],
                      "parallelism": f"batch-sharded x{world}", "rccl_world_size": world},
           "metric_keys": {"value": "B x K / wall time of the K timed steps (pipelined calls, control['sync']=False)",
                           "sec_8d_metric": "experiment_1_protocol.QPs_per_sec_median",
                           "unchanged_reference_harness": "step_sync_default (the layer's default control: every call waits "
                                                          "for its solve and raises at the call, as experiments/experiment_1.py "
                                                          "gets it without setting any extension key)"},
           "experiment_1_protocol": protocol,
           "roofline": roofline, "roofline_loop": roof_loop, "roofline_factorisation": roof_factor,
           "kernel_ms_per_step": breakdown, "profiled_pass_ms_per_step": round(dt_prof / args.steps * 1e3, 4)}

    if world == 1 and not args.no_other_configs and B == B_PER_GPU and n == N_X and not args.strong:
        # ---- the same step on the north-star-named algorithm (cached pivoted LU) and with the layer's default
        #      synchronous calls, driver-measured every round ----
        for key, (lsv, syncv) in {"step_linsolve_lu": ("lu", args.sync), "step_sync_default": (args.linsolve, True)}.items():
            if (lsv, syncv) == (args.linsolve, bool(args.sync)):
                continue
            _, lay = make_layer(lsv, syncv)
            for i in range(10):
                step(i, lay)
            # (the synchronous step is paced by the host between its calls: ten steps -- 8 ms -- moved by 5 % from run to run)
            k = 40 if key == "step_sync_default" else 10
            rounds_x = sorted(timed(k, lay) for _ in range(3 if key == "step_sync_default" else 1))
            dtx = rounds_x[len(rounds_x) // 2]
            L.synchronize()
            out[key] = {"value": round(B / (dtx / k), 1), "unit": "QPs/sec", "ms_per_step": round(dtx / k * 1e3, 4), "steps": k}
            if key == "step_sync_default":
                out["config"]["step_sync_default_QPs_per_sec"] = out[key]["value"]      # (the reference-compatible rate travels with the headline)
                out["config"]["step_sync_default_ms_per_step"] = out[key]["ms_per_step"]
                out[key]["timing"] = "median of three rounds of 40 steps after 10 warm-up steps"
                out[key]["rounds_ms_per_step"] = [round(t / k * 1e3, 4) for t in rounds_x]
            if key == "step_linsolve_lu":
                # the north-star-named algorithm: its kernel classes (HIP events on the launch stream) and the committed profile
                _lib.profile(enable=True, reset=True)
                timed(k, lay)
                pr = _lib.profile()
                _lib.profile(enable=False)
                out[key]["kernel_ms_per_step"] = {c: round(v[0] / k, 4) for c, v in pr.items() if v[1]}
                N_f, N_b = n + m, int(0.63 * n) + m
                lu_ms = pr.get("lu_factor", (0.0, 0))[0] / k
                out[key]["lu_factorisations"] = {
                    "kernel": "lqp::k_lu_factor2 (two workgroups per matrix, csrc/lqp_lu2.hpp)",
                    "algorithmic_flops_per_step": round(B * 2.0 / 3.0 * (N_f ** 3 + N_b ** 3)), "ms_per_step": round(lu_ms, 4),
                    "algorithmic_TFLOPs": round(B * 2.0 / 3.0 * (N_f ** 3 + N_b ** 3) / (lu_ms * 1e-3) / 1e12, 2) if lu_ms > 0 else None,
                    "algorithmic_flops_are": "an ESTIMATE: the backward's reduced system is taken as 0.63 n + m rows (the free set of the benchmark data), not read back from the run",
                    "note": "forward KKT matrix (n + m rows) + the backward's reduced system (~0.63 n + m rows); the panel chain of "
                            "partial pivoting is serial: the matrix-core share (trailing update, U12) is the profile's MFMA figure"}
                loop_lu_ms = pr.get("admm_loop", (0.0, 0))[0] / k
                lb = es * (iters + 1) * (n + m) ** 2 * B
                out[key]["loop"] = {"kernel": "lqp::k_admm_loop (cached triangular solves)", "ms_per_step": round(loop_lu_ms, 4),
                                    "algorithmic_bytes": lb, "achieved_GBs": round(lb / (loop_lu_ms * 1e-3) / 1e9, 1) if loop_lu_ms > 0 else None,
                                    "frac_of_hbm_peak": round(lb / (loop_lu_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if loop_lu_ms > 0 else None}
                out[key]["profile"] = profile_twin("lu")
        # ---- forward-only for configs 2 and 4 (B=128, n=100 box-only / n=1000 with the equality row),
        #      inputs drawn on the device, roofline terms from the run itself ----
        extras = {}
        qp_fwd = L.SolveBoxQP(control=dict(control, sync=False))
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
            fwd = lambda: qp_fwd(Qx, px, Ax, bx, lbx, ubx)
            for _ in range(2):
                fwd()
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                fwd()
            torch.cuda.synchronize(dev)
            dtx = (time.perf_counter() - t1) / reps
            stx = last_forward_status(dev)
            mm = 1 if with_eq else 0
            fb = algorithmic_bytes(es, nn, mm, stx["iters"], stx["n_factor"] - 1, stx["linsolve_used"])[0]
            fb_ref = algorithmic_bytes(es, nn, mm, stx["iters"], stx["n_factor"] - 1, 1)[0]
            extras[name] = {"QPs_per_sec": round(B / dtx, 1), "ms": round(dtx * 1e3, 4), "batch": B,
                            "iters": stx["iters"], "checks": stx["n_check"], "linsolve": {1: "lu", 2: "spd"}[stx["linsolve_used"]],
                            "forward_algorithmic_bytes": fb * B,
                            "forward_frac_of_hbm_roofline": round(fb * B / dtx / 1e9 / HBM_PEAK_GBS, 4),
                            "forward_frac_of_hbm_roofline_reference_algorithm_bytes": round(fb_ref * B / dtx / 1e9 / HBM_PEAK_GBS, 4),
                            "note": ("whole factor resident on chip (registers / LDS) for the loop: counter bytes << algorithmic"
                                     if nn <= 128 else ""),
                            "profile": profile_twin("config2" if nn == 100 else "config4")}
            del Qx
        L.synchronize()
        out["other_configs_forward_only"] = extras
        # ---- forward + backward at other batch sizes and in the layer's other modes (same distribution, inputs drawn
        #      on the device): B = 32 (the minibatch of experiments/experiment_2.py:12-20), B = 1024 (the per-GPU shard
        #      of BASELINE configs[4]), `unroll` and `backward='kkt'` (the published "ADMM Unroll" / "ADMM KKT" rows), and
        #      the hard distribution of experiments/experiment_1_hard.py in float64 ----
        def device_batch(Bx, nn, seed):
            gen = torch.Generator(device=dev).manual_seed(seed)
            Lm = torch.randn(Bx, 2 * nn, nn, device=dev, generator=gen)
            Qx = torch.matmul(Lm.transpose(1, 2), Lm) / (2 * nn)
            del Lm
            return (Qx, torch.randn(Bx, nn, 1, device=dev, generator=gen), torch.ones(Bx, 1, nn, device=dev),
                    torch.ones(Bx, 1, 1, device=dev), -(torch.rand(Bx, nn, 1, device=dev, generator=gen) + 1),
                    torch.rand(Bx, nn, 1, device=dev, generator=gen) + 1)

        def fwd_bwd_rate(lay, batch, reps, warm=2, grad_q=True):
            Qx, px = batch[0], batch[1]
            cot = torch.ones_like(px)

            def one():
                Qg = Qx.detach().requires_grad_(grad_q)
                pg = px.detach().requires_grad_(True)
                lay(Qg, pg, *batch[2:]).backward(cot)
            for _ in range(warm):
                one()
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(reps):
                one()
            torch.cuda.synchronize(dev)
            L.synchronize()
            dtx = (time.perf_counter() - t1) / reps
            return {"QPs_per_sec": round(batch[0].shape[0] / dtx, 1), "ms_per_step": round(dtx * 1e3, 4), "batch": batch[0].shape[0],
                    "n": batch[0].shape[1], "steps": reps}

        more = {}
        piped = lambda **kw: L.SolveBoxQP(control=dict(L.box_qp_control(eps_rel=TOL, eps_abs=TOL, verbose=False, **kw), sync=False))
        # the per-GPU shards of the metric string's own split ("batch=128 ... at 1/2/4/8 GPUs": 64 / 32 / 16 QPs per GPU)
        more["b64_n500_fwd_bwd"] = dict(fwd_bwd_rate(piped(), device_batch(64, n, 75), 20, warm=5), is_shard_of="batch=128 at 2 GPUs",
                                        profile=profile_twin("b64"))
        more["b32_n500_fwd_bwd"] = dict(fwd_bwd_rate(piped(), device_batch(32, n, 77), 20, warm=5), is_shard_of="batch=128 at 4 GPUs",
                                        profile=profile_twin("b32"))
        more["b16_n500_fwd_bwd"] = dict(fwd_bwd_rate(piped(), device_batch(16, n, 76), 20, warm=5), is_shard_of="batch=128 at 8 GPUs",
                                        profile=profile_twin("b16"))
        more["b1024_n500_fwd_bwd_config5_shard"] = dict(fwd_bwd_rate(piped(), device_batch(1024, n, 78), 5, warm=2),
                                                        profile=profile_twin("config5shard"))
        b128 = device_batch(B, n, 79)
        more["b128_n500_backward_kkt"] = fwd_bwd_rate(piped(backward='kkt'), b128, 10, warm=3)
        more["b128_n500_unroll"] = fwd_bwd_rate(L.SolveBoxQP(control=L.box_qp_control(eps_rel=TOL, eps_abs=TOL, verbose=False, unroll=True)),
                                                b128, 10, warm=3)
        more["b128_n500_unroll"]["profile"] = profile_twin("unroll")
        del b128
        # above the on-chip tiers (n + m > 1024: pivoted LU with two panel rows per thread, factor streamed from HBM / L2)
        more["b8_n1500_fwd_bwd_lu_tier"] = dict(fwd_bwd_rate(piped(), device_batch(8, 1500, 80), 3, warm=1), profile=profile_twin("n1500"))
        from lqp_py_amd.synthetic import create_hard_qp_data
        hard = create_hard_qp_data(250, 0.85, range(B), dtype=torch.float64, device=dev)      # prob 0.85 (experiment_1_hard.py:15), m = round(sqrt(250)) = 16
        more["b128_n250_m16_hard_fp64"] = dict(fwd_bwd_rate(piped(), hard, 20, warm=5), dtype="f64", linsolve="lu (pivoted LU: f64)",
                                               profile=profile_twin("hard64"))
        # ... and with the layer's default synchronous calls: what an unchanged experiments/experiment_1_hard.py gets (the LU form of
        # the backward factorised ahead of the cotangent and reporting behind its LU, ABI 11)
        more["b128_n250_m16_hard_fp64_sync_default"] = dict(
            fwd_bwd_rate(L.SolveBoxQP(control=L.box_qp_control(eps_rel=TOL, eps_abs=TOL, verbose=False)), hard, 40, warm=10), dtype="f64")
        del hard
        # the same distribution in float32: the symmetric tier with sixteen equality rows (their m x m systems -- the forward's
        # correction, the backward's Schur complement -- on one thread per entry since round 5: 0.96 -> 0.42 ms per step)
        hard32 = create_hard_qp_data(250, 0.85, range(B), dtype=torch.float32, device=dev)
        more["b128_n250_m16_hard_fp32"] = dict(fwd_bwd_rate(piped(), hard32, 20, warm=5), dtype="f32", linsolve="spd (m = 16 equality rows)")
        del hard32
        out["other_workloads_fwd_bwd"] = more
        # ---- the reference's training experiment (experiments/experiment_2.py:12-20,57-99): Linear(5 -> 500) -> layer -> QP
        #      loss -> SGD, minibatch 32, 100 epochs, tol 1e-5; published (tol 1e-3 variant): 25.3 s ----
        try:
            sys.path.insert(0, os.path.join(REPO, "examples"))
            import experiment_2 as E2
            E2.train(n_x=n, n_epochs=5, dev=dev, verbose=False)              # (start-up)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            losses, _, _ = E2.train(n_x=n, n_epochs=100, dev=dev, verbose=False)
            torch.cuda.synchronize(dev)
            out["experiment_2_n500_100_epochs"] = {"seconds": round(time.perf_counter() - t1, 4), "minibatch": 32, "epochs": 100,
                                                   "loss_first": round(losses[0], 4), "loss_last": round(losses[-1], 4),
                                                   "published_other_hardware_s": 25.3}
        except Exception as exc:                                              # (an extra: never takes the headline down)
            out["experiment_2_n500_100_epochs"] = {"error": repr(exc)[:200]}
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
