#!/usr/bin/env python
"""Benchmark: ELBO+entropy evaluations per second at BASELINE.json's headline shape.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python bench.py --gpus N --steps K --warmup W          # N > 1: starts its own N ranks (one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one `_neg_elcbo(theta, gp, vp, beta=0, Ns=NsK, compute_grad=True,
compute_var=False, theta_bnd)` -- the call Adam makes once per iteration
(reference vbmc/variational_optimization.py:238-249) -- on synthetic inputs of
BASELINE config 3 (D=10, K=50, N=400, Ns=1e6 Monte-Carlo samples, S=1 GP
hyper-sample), everything already resident in HBM.  With N GPUs the job is
BASELINE config 4's shape: Ns = N x 1e6 samples sharded over the ranks (weak
scaling), ONE RCCL all-reduce per evaluation.  `value` counts evaluations in
units of 1e6 samples, i.e. value = evals/s x (Ns_job / 1e6): at N=1 it is exactly
evals/s at Ns=1e6.  With N > 1 the line also carries `strong_scaling`: the SAME Ns=1e6 job split
N ways (what `metric` literally says; `--scaling strong` makes that one `value`).
`--config 4` runs BASELINE config 4's JOB (Ns=8e6) split over the N ranks -- on one GPU the whole
job, several grid rounds of the entropy kernel; `--config 5 --job` likewise (Ns=4e6).

Prints ONE JSON line on rank 0.  No PyTorch anywhere in the measured path.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

FP64_PEAK_TFLOPS = 78.6  # MI355X datasheet: FP64 vector == FP64 matrix (MFMA) peak
HBM_PEAK_GBS = 8000.0    # /opt/skills/guides/MI355X_MICROARCH.md


# per-GPU Monte-Carlo sample count of each BASELINE config that has a bench line (config 4 is
# config 3's shape on 8 GPUs, config 5 is quoted on 8 GPUs: its per-GPU share is Ns/8)
PER_GPU_NS = {2: 100_000, 3: 1_000_000, 4: 1_000_000, 5: 4_000_000 // 8}
JOB_NS = {2: 100_000, 3: 1_000_000, 4: 8_000_000, 5: 4_000_000}  # BASELINE.json `configs`
# The timed region repeats its K steps until it is at least this long.  The headline line's is
# long enough for the driver's 5-second GPU-activity sampler to land inside it.
MIN_TIMED_S = 6.0
MIN_TIMED_SECONDARY_S = 0.5


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--config", type=int, default=3, choices=sorted(PER_GPU_NS),
                   help="BASELINE config whose per-GPU shape is run (3 = the headline metric's; 4 = config 4's "
                        "job, Ns = 8e6, split over the ranks)")
    p.add_argument("--S", type=int, default=1,
                   help="GP hyper-parameter samples of the measured workload (1 = the headline; the reference's default is "
                        "about 80 / sqrt(N), gaussian_process_train.py:455-472: 4 at N = 400; S = 4 and 8 are also "
                        "reported as secondary figures of the default run)")
    p.add_argument("--rng", choices=["philox", "resident"], default="philox",
                   help="philox: fresh in-kernel draws every eval; resident: HBM-resident eps reused")
    p.add_argument("--job", action="store_true",
                   help="run the config's JOB size split over the ranks (config 4 always does): Ns = 8e6 (config 4) / "
                        "4e6 (config 5) over N GPUs, the whole job on one GPU at N = 1")
    p.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                   help="N > 1, config 3: what `value` is -- weak: Ns = N x 1e6 (config 4's reading; the default), "
                        "strong: Ns = 1e6 split N ways.  The other one is reported beside it.")
    p.add_argument("--min-timed-s", type=float, default=None,
                   help=f"minimum length of the timed region (default {MIN_TIMED_S} s for the headline line, "
                        f"{MIN_TIMED_SECONDARY_S} s for secondary lines)")
    p.add_argument("--spawn", action="store_true",
                   help="start the ranks from this process even for --gpus 1 (what --gpus N > 1 does when it is not "
                        "already running under a launcher)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-secondary", action="store_true",
                   help="skip the secondary figures (predict roofline, device-resident loop, reference stream)")
    p.add_argument("--cpu-sample-nsk", type=int, default=0,
                   help="per-component samples of the CPU-baseline run (0 = a sample sized for ~10-30 s of "
                        "host work: the whole workload at configs 2 and 3, a fifth of it at config 5)")
    p.add_argument("--cpu-reps", type=int, default=3, help="CPU-baseline runs; the MEDIAN is reported (SURVEY 8d: median of >= 3)")
    return p.parse_args()


def algorithmic_flops(D, K, ns_rows_total, grad=True):
    """entropy value Ns*K*(3D+4) flops (+ grads Ns*K*(3D+3));
    exp/log are NOT counted as flops.  ns_rows_total = samples this launch covers."""
    f = ns_rows_total * K * (3 * D + 4)
    if grad:
        f += ns_rows_total * K * (3 * D + 3)
    return float(f)


def cpu_baseline(wl, sample_nsk, reps=3):
    """The oracle (a NumPy port structurally identical to the reference's loops)
    timed on a bounded sample of the same workload on this box's host cores."""
    from oracle import elbo_ref, gp_ref, mixture_ref
    from pyvbmc_amd import synthetic

    mix = mixture_ref.Mixture.make(wl.mu, wl.sigma, wl.lambd, wl.w, wl.eta)
    ogp = gp_ref.make_gp(wl.X, wl.y, wl.hyp)
    bnd = synthetic.default_theta_bnd(wl)
    sample_nsk = sample_nsk or wl.NsK
    eps = synthetic.draw_eps_half(wl.K, wl.D, sample_nsk, seed=99)
    dts = []
    cpu0, wall0 = time.process_time(), time.perf_counter()
    for _ in range(max(1, reps)):
        t0 = time.perf_counter()
        elbo_ref.neg_elcbo(wl.theta.copy(), ogp, mix, 0.0, sample_nsk, True, False, bnd, eps_half=eps)
        dts.append(time.perf_counter() - t0)
    dt = float(np.median(dts))
    scale = wl.NsK / sample_nsk  # cost is linear in the sample count (entropy > 99 %)
    # threads actually used: process CPU time over wall time (NumPy's element-wise kernels, which
    # dominate this path exactly as in the reference, run on one thread whatever the core count)
    busy = (time.process_time() - cpu0) / max(time.perf_counter() - wall0, 1e-9)
    cores = max(1, int(round(busy)))
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count()
    return {
        "value": 1.0 / (dt * scale),
        "unit": "evals/s (Ns=1e6-equivalent)",
        "cores": cores,
        "kind": "port",
        "sample": f"{len(dts)} value+grad evals at NsK={sample_nsk} per component ({sample_nsk * wl.K} samples; "
                  f"{', '.join('%.2f' % t for t in dts)} s, median taken)"
                  + (f", scaled x{scale:.1f} to NsK={wl.NsK}" if scale != 1.0 else "")
                  + f"; NumPy default threading, {busy:.2f} threads busy on average of {avail} available",
    }


def spawn_ranks(a):
    """`python bench.py --gpus N` outside a launcher: start the N ranks ourselves (one process per GPU,
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, as torch.distributed.run would
    -- no torchrun needed), hand rank 0's JSON line through, exit with the worst return code."""
    import socket
    import subprocess

    from pyvbmc_amd import _lib

    n_dev = _lib.device_count()
    if n_dev < a.gpus:
        sys.stderr.write(f"bench.py: --gpus {a.gpus} needs {a.gpus} visible GPUs, found {n_dev}\n")
        sys.exit(2)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(a.gpus),
               TORCHELASTIC_RUN_ID=f"bench{os.getpid()}", HSA_ENABLE_IPC_MODE_LEGACY="0",
               VBMC_LAUNCH_NONCE=os.urandom(8).hex())
    argv = [sys.executable, str(Path(__file__).resolve())] + [x for x in sys.argv[1:] if x != "--spawn"]
    procs = [subprocess.Popen(argv, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True)
             for r in range(a.gpus)]
    out0 = procs[0].communicate()[0]
    rcs = [p.wait() for p in procs]
    sys.stdout.write(out0)
    sys.stdout.flush()
    sys.exit(max(abs(rc) for rc in rcs))


def main():
    a = parse()
    if "RANK" not in os.environ and (a.gpus > 1 or a.spawn):
        spawn_ranks(a)
    from pyvbmc_amd import VariationalPosterior, _lib, comm, synthetic
    from pyvbmc_amd import gp as gpm
    from pyvbmc_amd.variational_optimization import _neg_elcbo

    rank, world, local_rank = comm.env_rank_world()
    if world != a.gpus:
        a.gpus = world
    job_mode = a.job or a.config == 4
    if job_mode and a.config not in (4, 5):
        sys.exit("bench.py: --job applies to configs 4 and 5")
    # the CPU baseline first (rank 0, N = 1): the GPU legs then run back to back to the end of the
    # process, where a coarse activity sampler can see them
    cpu_res = None
    ns_gpu = (JOB_NS[a.config] // world) if job_mode else PER_GPU_NS[a.config]
    wl = synthetic.make_workload(a.config, Ns_total=ns_gpu, S=a.S)
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        # ~10-30 s of host work: the whole workload at configs 2 and 3, a bounded part of it otherwise
        div = {2: 1, 3: 1, 4: 8, 5: 5 if not job_mode else 40}[a.config]
        nsk_cpu = a.cpu_sample_nsk or max(2, (wl.NsK // div) // 2 * 2)
        cpu_res = cpu_baseline(wl, nsk_cpu, a.cpu_reps)

    ctx = _lib.Context(local_rank)
    _lib.set_default_context(ctx)
    comm.init_from_env(ctx)
    comm_rank, comm_world = ctx.comm_info()  # what RCCL itself reports

    D, K = wl.D, wl.K
    vp = VariationalPosterior(D, K)
    vp.mu, vp.sigma, vp.lambd = wl.mu.copy(), wl.sigma.reshape(1, -1), wl.lambd.reshape(-1, 1)
    vp.w, vp.eta = wl.w.reshape(1, -1), wl.eta.reshape(1, -1)
    gp = gpm.GP(D, gpm.SquaredExponential(), gpm.NegativeQuadratic(),
                gpm.GaussianNoise(constant_add=True, user_provided_add=wl.s2 is not None))
    gp.update(X_new=wl.X, y_new=wl.y, s2_new=wl.s2, hyp=wl.hyp)
    bnd = synthetic.default_theta_bnd(wl)
    theta = wl.theta.copy()

    # per-component samples of the whole job under either reading of "N GPUs":
    #   weak   -- every rank keeps the per-GPU share, the job grows with N (config 3 -> config 4's shape)
    #   strong -- the job is fixed (Ns = 1e6 at config 3; the config's job size with --job) and split N ways
    if job_mode:
        nsk_of = {"strong": synthetic.ns_per_component(JOB_NS[a.config], K)}
        scaling = "strong"
    else:
        nsk_of = {"weak": wl.NsK * world, "strong": wl.NsK}
        scaling = a.scaling if world > 1 else "weak"
    if any((n // 2) < world for n in nsk_of.values()):
        sys.exit("bench.py: fewer antithetic rows per component than ranks")

    def make_step(nsk_job, rng):
        step_no = [0]
        if rng == "philox":

            # a different parameter vector every call, as in Adam: seven of them, made outside the timed loop (the metric is
            # the evaluation, not the caller's arithmetic on theta; _neg_elcbo's in-place max-shift of the eta tail,
            # variational_optimization.py:1082-1085, is idempotent on them)
            ths = [theta + 1e-9 * i for i in range(7)]

            def step():
                step_no[0] += 1
                return _neg_elcbo(ths[step_no[0] % 7], gp, vp, 0.0, nsk_job, True, False, bnd, rng="philox", seed=step_no[0])

            return step
        # HBM-resident draws, uploaded once and reused by every evaluation: the fused
        # C-ABI entry is called directly so that nothing is re-uploaded per step.
        import ctypes as C

        vp._upload(ctx)
        gpm.upload_gp(gp, ctx)
        full_rows = nsk_job // 2
        r0, r1 = comm.shard_rows(full_rows, rank, world)
        big = np.zeros((K, full_rows, D))
        big[:, r0:r1, :] = np.random.default_rng(1000 + rank).standard_normal((K, r1 - r0, D))
        ctx.set_eps(big, r0, r1 - r0)
        del big
        opts = _lib.ElboOpts()
        opts.ns_per_comp, opts.eps_mode, opts.seed = nsk_job, _lib.EPS_RESIDENT, 0
        opts.compute_grad, opts.optimize_mask = 1, 15
        opts.row_begin, opts.row_count = 0, -1
        lb, ub = _lib.f64(bnd["lb"]), _lib.f64(bnd["ub"])
        opts.bnd_lb, opts.bnd_ub, opts.n_bnd = _lib.ptr(lb), _lib.ptr(ub), lb.size
        opts.tol_con, opts.weight_threshold, opts.weight_penalty = (
            bnd["tol_con"], bnd["weight_threshold"], bnd["weight_penalty"])
        Fc, Gc, Hc = C.c_double(), C.c_double(), C.c_double()
        dF = np.empty(theta.size)
        th = theta.copy()

        def step(_keep=(lb, ub)):  # (the bounds' buffers live as long as the closure)
            step_no[0] += 1
            th[:] = theta + 1e-9 * (step_no[0] % 7)  # a different parameter vector every call
            ctx.check(ctx._lib.vbmc_neg_elcbo(ctx._h, _lib.ptr(th), th.size, C.byref(opts),
                                              C.byref(Fc), _lib.ptr(dF), C.byref(Gc), C.byref(Hc),
                                              None, None, None, None, None))
            return Fc.value, dF, Gc.value, Hc.value, 0

        return step

    def measure(nsk_job, min_timed_s, rng):
        """W warm-up steps, a calibration, then the timed region: R x K steps back to back between
        barrier + synchronize on both sides, the maximum over the ranks."""
        step = make_step(nsk_job, rng)
        for _ in range(a.warmup):
            out = step()
        ctx.synchronize()
        # a short calibration run (after the warm-up, so first-call allocations are out of it) sizes
        # the timed region
        n_cal = max(3, min(20, a.steps))
        t_w = time.perf_counter()
        for _ in range(n_cal):
            out = step()
        ctx.synchronize()
        est = (time.perf_counter() - t_w) / n_cal
        # The timed region is R back-to-back repetitions of the K requested steps, R chosen so that it
        # lasts >= min_timed_s whatever K is (a 20-step region is 3 ms: too short to mean anything).
        # All ranks must agree on R: take the maximum estimate.
        est = ctx.comm_max(est)
        repeats = max(1, int(np.ceil(1.15 * min_timed_s / max(a.steps * est, 1e-9))))
        n_timed = repeats * a.steps
        ctx.comm_barrier()
        ctx.synchronize()
        # The main kernel's duration comes from HIP events carried by its own dispatch packet
        # (hipExtLaunchKernel: no barrier packet in the queue), read back on every SAMPLE_EVERY-th step
        # of the timed region together with the library's host-side breakdown; the other steps run
        # without any instrumentation call.
        SAMPLE_EVERY = 128  # (a sampled step is not armed, api_elbo.hip: every 32nd cost the mean 0.3 us)
        kern_ms = []
        host_us = np.zeros(5)
        n_host = 0
        t0 = time.perf_counter()
        for i in range(n_timed):
            if i % SAMPLE_EVERY <= 1:
                ctx.set_timing(i % SAMPLE_EVERY == 0)  # on for the sampled step, off again after it
            out = step()
            if i % SAMPLE_EVERY == 0:
                kern_ms.append(ctx.last_kernel_ms(0))
                host_us += ctx.last_host_us()
                n_host += 1
        ctx.set_timing(False)
        ctx.synchronize()
        ctx.comm_barrier()
        dt = time.perf_counter() - t0
        dt = ctx.comm_max(dt)
        if not np.isfinite(out[0]):
            sys.exit("non-finite objective")
        return {"dt": dt, "n_timed": n_timed, "repeats": repeats, "F": out[0], "kern_ms": float(np.mean(kern_ms)),
                "n_kern": len(kern_ms), "host_us": host_us / max(n_host, 1), "plan": ctx.last_entmc_plan(),
                "sample_every": SAMPLE_EVERY, "nsk_job": nsk_job}

    nsk_job = nsk_of[scaling]
    ns_job = nsk_job * K
    headline = a.config == 3 and not job_mode and a.S == 1
    min_timed = a.min_timed_s if a.min_timed_s is not None else (MIN_TIMED_S if headline else MIN_TIMED_SECONDARY_S)

    predict_roofline = adam_loop = reference_stream = None
    if not a.no_secondary:
        # Secondary roofline (third way): gp.predict at the acquisition batch size,
        # its GEMM flops against the FP64 matrix-core peak.  Timed with HIP events around the
        # variance product alone (`frac`) and around the three predict launches (K*, product, finish).
        M_pred = 8192
        xs_pred = np.random.default_rng(7).standard_normal((M_pred, D))
        gp.predict(xs_pred, separate_samples=True)
        pm, vm = [], []
        # two passes: the event pair around the variance product sits BETWEEN predict's launches (two records of
        # ~6 us each), so the interval around all of them is read with that pair off
        ctx.set_timing(1)
        for _ in range(9):
            gp.predict(xs_pred, separate_samples=True)
            pm.append(ctx.last_kernel_ms(3))
        ctx.set_timing(2)
        for _ in range(9):
            gp.predict(xs_pred, separate_samples=True)
            vm.append(ctx.last_kernel_ms(5))
        ctx.set_timing(False)
        nt = (wl.N + 63) // 64
        gemm_flops = 2.0 * M_pred * 64 * sum(min((c + 1) * 64, wl.N) for c in range(nt))  # triangular skip
        pred_ms, var_ms = float(np.median(pm)), float(np.median(vm))
        predict_roofline = {
            "kernel": "predict_var_dma_kernel (S=1, M=8192): T = (sW o K*) L^-1 on the FP64 matrix cores",
            "bound": "mfma", "achieved": gemm_flops / (var_ms * 1e-3) / 1e12, "peak": FP64_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": gemm_flops / (var_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
            "kernel_ms": var_ms, "gemm_flops": gemm_flops,
            "kernel_ms_from": "HIP events around that launch alone (vbmc_set_timing(2), vbmc_last_kernel_ms which=5; the "
                              "finish launch separate in that pass), median of 9",
            "all_launches_ms": pred_ms,
            "all_launches": "predict_kstar_mfma + predict_var_dma with the finish in its epilogue (two launches), HIP events "
                            "around the two and NO record between them (vbmc_set_timing(1): rounds 2-4 read this interval "
                            "with the inner pair recorded as well, which added ~5 us), median of 9",
            "frac_all_launches": gemm_flops / (pred_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
        }
        # Secondary figure (not `value`): the same evaluation inside the device-resident optimiser
        # loop -- no host round trip per evaluation; every rank runs it (the
        # all-reduce is in-stream).  Measured first: it also brings the GPU clocks up before the
        # W warm-up steps and the timed region.
        from pyvbmc_amd.minimize_adam import minimize_adam_elbo

        n_loop = max(200, min(a.steps, 400))
        kw = dict(max_iter=n_loop, use_early_stopping=False, seed=12345, rng="philox")
        minimize_adam_elbo(theta.copy(), gp, vp, nsk_job, bnd, **kw)  # warm-up
        dts_loop = []
        for _ in range(3):  # three whole optimisations (begin, n_loop iterations, end), the median reported
            ctx.comm_barrier()
            t1 = time.perf_counter()
            loop = minimize_adam_elbo(theta.copy(), gp, vp, nsk_job, bnd, **kw)
            ctx.comm_barrier()
            dts_loop.append(ctx.comm_max(time.perf_counter() - t1))
        dt_loop = float(np.median(dts_loop))
        adam_loop = {
            "iterations": n_loop,
            "runs_us_per_iteration": [1e6 * t / n_loop for t in dts_loop],
            "stat": "median of 3 runs of the whole optimisation (begin + iterations + end) after one warm-up run",
            "launches_per_iteration": 2 if ctx.last_entmc_plan().get("adam_tail") else 4,
            "us_per_iteration": 1e6 * dt_loop / n_loop,
            "evals_per_s": (n_loop / dt_loop) * (1.0 if job_mode else nsk_job * K / PER_GPU_NS[a.config]),
            "F_first_last": [float(loop[3][0]), float(loop[3][-1])],
        }
        if world == 1:
            # What that loop replaces: the reference's own minimize_adam (host NumPy update, stopping
            # rule off, vbmc/minimize_adam.py:84-137) around the same accelerated objective -- the step above plus the
            # optimiser's per-iteration host arithmetic on the 610-vector, which the headline's timed loop does not contain.
            from pyvbmc_amd.minimize_adam import minimize_adam

            seeds = iter(range(777, 10**9))

            def objective(t):
                r = _neg_elcbo(t, gp, vp, 0.0, nsk_job, True, False, bnd, rng="philox", seed=next(seeds))
                return r[0], r[1]

            hkw = dict(max_iter=n_loop, use_early_stopping=False)
            minimize_adam(objective, theta.copy(), **hkw)  # warm-up
            dts_host = []
            for _ in range(3):
                t1 = time.perf_counter()
                minimize_adam(objective, theta.copy(), **hkw)
                dts_host.append(time.perf_counter() - t1)
            adam_loop["host_driven_adam_loop"] = {
                "what": "minimize_adam (the reference's host loop: NumPy Adam update on the parameter vector per iteration) "
                        "around the accelerated _neg_elcbo, same iterations, fresh Philox draws per iteration",
                "runs_us_per_iteration": [1e6 * t / n_loop for t in dts_host],
                "us_per_iteration": 1e6 * float(np.median(dts_host)) / n_loop,
            }
        if world == 1 and not job_mode:
            # Secondary figure (never `value`): the drop-in's DEFAULT draw source, rng="numpy" -- the
            # reference's MT19937 stream drawn on the host cores (csrc/host_randn.hip) and shipped over
            # PCIe every evaluation (bit-identical inputs to the reference's).  The headline needs
            # rng="philox".
            _neg_elcbo(theta.copy(), gp, vp, 0.0, nsk_job, True, False, bnd, rng="numpy")
            n_ref = 15
            t_ref = []
            for _ in range(n_ref):
                t1 = time.perf_counter()
                _neg_elcbo(theta.copy(), gp, vp, 0.0, nsk_job, True, False, bnd, rng="numpy")
                t_ref.append(time.perf_counter() - t1)
            # the median: the generator runs on the host cores, which a GPU box shares with other jobs
            # (single evaluations of 15-50 ms occur); the spread is reported beside it
            dt_ref = float(np.median(t_ref))
            p10, p90 = (float(x) for x in np.percentile(t_ref, [10, 90]))
            reference_stream = {
                "evals_per_s": 1.0 / dt_ref, "ms_per_eval": 1e3 * dt_ref, "evals": n_ref, "stat": "median",
                "ms_per_eval_p10": 1e3 * p10, "ms_per_eval_p50": 1e3 * dt_ref, "ms_per_eval_p90": 1e3 * p90,
                "ms_per_eval_mean": 1e3 * float(np.mean(t_ref)), "ms_per_eval_max": 1e3 * float(np.max(t_ref)),
                "what": "rng='numpy' (the default): the reference's np.random.randn stream of K*NsK/2*D normals generated "
                        "ON THE DEVICE per evaluation (csrc/device_randn.hip: MT19937 with a GF(2) jump-ahead per workgroup, "
                        "polar method, prefix sum; words, accepted attempts and NumPy's state bit-identical, 99.9 % of the "
                        "values too, the rest within 3 ulp) -- no host generator, no PCIe; rounds 2-4 drew it on the host "
                        "cores and shipped 40 MB per evaluation (BENCH_r04: 106 evaluations/s)",
            }

    gp_samples = full_elcbo = None
    if not a.no_secondary and world == 1 and not job_mode and a.rng == "philox":
        # Secondary figures ("S = 1 (also S = 8)"): the same step with S = 4 (the reference's default at
        # N = 400) and S = 8 GP hyper-parameter samples -- step time, where the GP sums ran and when their word and
        # the entropy's reached the host (vbmc_last_step_marks: the GP word must arrive first, or the sums are
        # on the critical path)
        gp_samples = {}
        for S2 in (4, 8):
            if S2 == a.S:
                continue
            wl2 = synthetic.make_workload(a.config, Ns_total=ns_gpu, S=S2)
            gp2 = gpm.GP(D, gpm.SquaredExponential(), gpm.NegativeQuadratic(),
                         gpm.GaussianNoise(constant_add=True, user_provided_add=wl2.s2 is not None))
            gp2.update(X_new=wl2.X, y_new=wl2.y, s2_new=wl2.s2, hyp=wl2.hyp)
            th2 = wl2.theta.copy()

            def step2(i):
                return _neg_elcbo(th2 + 1e-9 * (i % 7), gp2, vp, 0.0, nsk_job, True, False, bnd, rng="philox", seed=50_000 + i)

            for i in range(30):
                step2(i)
            ctx.synchronize()
            n2 = max(200, int(0.4 / max(1e-6, 1.2e-4)))
            marks = []
            t1 = time.perf_counter()
            for i in range(n2):
                out2 = step2(30 + i)
                if i % 64 == 63:
                    marks.append(ctx.last_step_marks())
            ctx.synchronize()
            dt2 = (time.perf_counter() - t1) / n2
            gp_samples[f"S{S2}"] = {
                "ms_per_step": 1e3 * dt2, "evals_per_s": 1.0 / dt2, "steps": n2, "F": float(out2[0]),
                "gp_sums_in": marks[-1]["gp_sums_in"],
                "gp_word_us": float(np.median([m_["gp_word_us"] for m_ in marks])),
                "entropy_word_us": float(np.median([m_["entropy_word_us"] for m_ in marks])),
            }
        _neg_elcbo(theta.copy(), gp, vp, 0.0, nsk_job, True, False, bnd, rng="philox", seed=1)  # (the headline GP back on the device)
        # SURVEY 8d's secondary unit of work: `_eval_full_elcbo` (variational_optimization.py:428-500) -- value,
        # variance and the per-component terms (compute_var, separate_K) at ns_ent_fine = 2^12 per component
        nsk_full = 4096
        fe = lambda i: _neg_elcbo(theta.copy(), gp, vp, 0.0, nsk_full, False, True, None, 0.0, True, rng="philox", seed=90_000 + i)
        for i in range(5):
            r_full = fe(i)
        ctx.synchronize()
        t1 = time.perf_counter()
        n_full = 100
        for i in range(n_full):
            r_full = fe(5 + i)
        ctx.synchronize()
        dt_full = (time.perf_counter() - t1) / n_full
        full_elcbo = {"ms_per_eval": 1e3 * dt_full, "evals_per_s": 1.0 / dt_full, "NsK": nsk_full, "evals": n_full,
                      "F": float(r_full[0]), "varF": float(np.ravel(r_full[4])[0]),
                      "what": "_neg_elcbo(theta, gp, vp, 0, NsK=4096, compute_grad=False, compute_var=True, separate_K=True): "
                              "value + variance + per-component I_sk / J_sjk, the call _eval_full_elcbo makes"}

    # the other reading of "N GPUs" first (a short region), then the line's own
    other = None
    if world > 1 and not job_mode and a.rng == "philox":
        o = "strong" if scaling == "weak" else "weak"
        mo = measure(nsk_of[o], MIN_TIMED_SECONDARY_S, a.rng)
        ev = mo["n_timed"] / mo["dt"]
        other = {"scaling": o, "value": ev * (nsk_of[o] * K / PER_GPU_NS[a.config]), "evals_per_s_job": ev,
                 "unit": "evals/s (1e6-sample-equivalent)", "job_Ns": nsk_of[o] * K, "ms_per_step": 1e3 / ev,
                 "timed_steps": mo["n_timed"], "timed_region_s": mo["dt"], "entropy_kernel_ms": mo["kern_ms"],
                 "entropy_launch": mo["plan"],
                 "what": ("the SAME Ns=1e6 job split over the ranks (what `metric` literally says): per-rank entropy "
                          "kernel ~1/N of the single-GPU one, prep / finish / all-reduce / host turnaround unchanged"
                          if o == "strong" else "Ns = N x 1e6: every rank keeps config 3's per-GPU share (config 4's shape)")}
    m = measure(nsk_job, min_timed, a.rng)
    dt, n_timed, repeats, F, k_ms, plan = m["dt"], m["n_timed"], m["repeats"], m["F"], m["kern_ms"], m["plan"]
    host_us = m["host_us"]

    ms_per_step = 1e3 * dt / n_timed
    evals_per_s = n_timed / dt
    # `value`: the headline counts evaluations in units of the per-GPU share (1e6 samples at config 3),
    # so that N ranks at weak scaling report N x the single-GPU rate; job lines count whole-job evaluations
    value = evals_per_s * (1.0 if job_mode else ns_job / PER_GPU_NS[a.config])
    flops = algorithmic_flops(D, K, ns_job / world, grad=True)  # per launch (this rank's rows)
    achieved = flops / (k_ms * 1e-3) / 1e12
    achieved_e2e = flops / (ms_per_step * 1e-3) / 1e12
    # Algorithmic HBM bytes of the entropy launch: the antithetic half of the draws when they are INPUT
    # (--rng resident: parity mode), ~0 with the device generator.  What the step really moves in Philox mode is reported
    # beside it (step_hbm_bytes): the generator writes the draws into HBM one evaluation ahead and the kernel reads them
    # back -- traffic the implementation adds, hidden behind a compute-bound kernel but not algorithmic.
    draws_bytes = (ns_job / world / 2) * D * 8
    eps_bytes = draws_bytes if (plan["resident_draws"] and a.rng == "resident") else 0.0
    step_hbm_bytes = {"generator_writes": draws_bytes if a.rng == "philox" and plan["resident_draws"] else 0.0,
                      "entropy_kernel_reads": draws_bytes if plan["resident_draws"] else 0.0}
    traffic, traffic_from = None, None
    try:  # PMC-measured HBM bytes per launch (separate rocprofv3 --pmc passes, see profiles/README.md)
        tj = json.load(open(ROOT / "profiles" / "traffic.json"))
        ent = tj.get(f"config{a.config}" + ("_job" if job_mode and a.config == 5 else ""), {}).get(a.rng)
        if ent and world == 1:
            traffic = ent["hbm_bytes_per_launch"]
            traffic_from = {"file": "profiles/traffic.json", "source": ent.get("source"),
                            "collected": ent.get("collected"), "how": tj.get("_comment")}
    except (OSError, ValueError, KeyError):
        pass
    if headline:
        metric = "ELBO+entropy evals/sec at D=10, K=50, N=400, Ns=1e6 (1e6-sample-equivalent evals/s)"
    elif a.config == 3 and not job_mode:
        metric = f"ELBO+entropy evals/sec at D=10, K=50, N=400, Ns=1e6 with S={a.S} GP hyper-parameter samples (secondary line)"
    elif job_mode:
        metric = (f"ELBO+entropy evals/sec at D={D}, K={K}, N={wl.N}, Ns={JOB_NS[a.config]:.0e} (BASELINE config "
                  f"{a.config}'s whole job per evaluation, split over {world} GPU{'s' if world > 1 else ''}; secondary line)")
    else:
        metric = (f"ELBO+entropy evals/sec at D={D}, K={K}, N={wl.N}, Ns={PER_GPU_NS[a.config]:.0e} per GPU "
                  f"(BASELINE config {a.config}'s per-GPU share; secondary line)")
    res = {
        "metric": metric,
        "value": value,
        "unit": "evals/s",
        "n_gpus": world,
        "comm_world": comm_world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": ms_per_step,
        "timed_steps": n_timed,
        "timed_region_s": dt,
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"BASELINE config {a.config}{' job' if job_mode else ' per GPU'}: D={D} K={K} N={wl.N} S={a.S}, "
                        f"Ns={wl.Ns_total if not job_mode else ns_job // world} MC samples per GPU (job Ns={ns_job}), "
                        f"value+grad _neg_elcbo with soft bounds, eps={a.rng}",
            "evals_per_s_job": evals_per_s,
            "parallelism": f"sample-sharded x{world}, 1 RCCL all-reduce/eval" if world > 1 else "single GPU",
            "host_thread": (lambda b, n: f"{'narrowed to' if b else 'left on'} {n} CPUs "
                            f"({'the device-local NUMA node, by vbmc_ctx_create' if b else 'affinity untouched'})")(*ctx.host_affinity()),
            "timed_region": f"{repeats} x {a.steps} steps back to back (>= {min_timed} s), one barrier + "
                            f"synchronize on either side",
            "entropy_launch": plan,
        },
        "roofline": {
            "kernel": "entmc main kernel (%s)" % plan["kernel"],
            "bound": "mfma",
            "bound_detail": "FP64 arithmetic issue.  The kernel executes NO MFMA instruction: it is bound by the "
                            "float64 vector FMA pipe, whose peak on MI355X equals the dense FP64 matrix peak "
                            "(78.6 TFLOP/s) -- the label the contract offers for a compute bound; exp/log not "
                            "counted as flops",
            "achieved": achieved,
            "peak": FP64_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": achieved / FP64_PEAK_TFLOPS,
            "traffic": traffic,
            "traffic_from": traffic_from,
            "kernel_ms": k_ms,
            "kernel_ms_from": f"HIP events on the kernel's own dispatch, read on every {m['sample_every']}th of the {n_timed} "
                              f"timed steps ({m['n_kern']} launches)",
            "algorithmic_flops_per_launch": flops,
            "hbm_bytes_per_launch_algorithmic": eps_bytes,
            "step_hbm_bytes": step_hbm_bytes,
            "hbm_achieved_GBs": (eps_bytes / (k_ms * 1e-3) / 1e9) if eps_bytes else 0.0,
            "hbm_peak_GBs": HBM_PEAK_GBS,
        },
        "roofline_e2e": {
            "what": "the same algorithmic flops over the whole host-driven step (ms_per_step): upload, prep, "
                    "entropy, finish, host finalisation and Python included",
            "achieved": achieved_e2e, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved_e2e / FP64_PEAK_TFLOPS,
        },
        "predict_roofline": predict_roofline,
        "device_resident_adam_loop": adam_loop,
        "reference_stream": reference_stream,
        "gp_samples": gp_samples,
        "full_elcbo": full_elcbo,
        "F": F,
        "host_us_per_step": dict(zip(["pack_upload", "launch", "wait_device", "finalize", "c_total"],
                                     np.asarray(host_us).round(2).tolist())),
    }
    if other is not None:
        res[other["scaling"] + "_scaling"] = other
    if cpu_res is not None:
        res["cpu_baseline"] = cpu_res
        res["speedup_vs_cpu_baseline"] = value / cpu_res["value"]
    if rank == 0:
        print(json.dumps(res))
    ctx.close()


if __name__ == "__main__":
    main()
