#!/usr/bin/env python
"""Benchmark: ELBO+entropy evaluations per second at BASELINE.json's headline shape.

    python bench.py --gpus 1 --steps 50 --warmup 5
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
evals/s at Ns=1e6.

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
PER_GPU_NS = {2: 100_000, 3: 1_000_000, 5: 4_000_000 // 8}
MIN_TIMED_S = 0.5  # the timed region repeats its K steps until it is at least this long


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--config", type=int, default=3, choices=sorted(PER_GPU_NS),
                   help="BASELINE config whose per-GPU shape is run (3 = the headline metric's)")
    p.add_argument("--rng", choices=["philox", "resident"], default="philox",
                   help="philox: fresh in-kernel draws every eval; resident: HBM-resident eps reused")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-secondary", action="store_true",
                   help="skip the secondary figures (predict roofline, device-resident loop, reference stream)")
    p.add_argument("--cpu-sample-nsk", type=int, default=0,
                   help="per-component samples of the CPU-baseline run (0 = a sample sized for ~10-30 s of "
                        "host work: the whole workload at configs 2 and 3, a fifth of it at config 5)")
    p.add_argument("--cpu-reps", type=int, default=2)
    return p.parse_args()


def algorithmic_flops(D, K, ns_rows_total, grad=True):
    """entropy value Ns*K*(3D+4) flops (+ grads Ns*K*(3D+3));
    exp/log are NOT counted as flops.  ns_rows_total = samples this launch covers."""
    f = ns_rows_total * K * (3 * D + 4)
    if grad:
        f += ns_rows_total * K * (3 * D + 3)
    return float(f)


def cpu_baseline(wl, sample_nsk, reps=2):
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
    dt = min(dts)
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
                  f"{', '.join('%.2f' % t for t in dts)} s, best taken)"
                  + (f", scaled x{scale:.1f} to NsK={wl.NsK}" if scale != 1.0 else "")
                  + f"; NumPy default threading, {busy:.2f} threads busy on average of {avail} available",
    }


def main():
    a = parse()
    from pyvbmc_amd import VariationalPosterior, _lib, comm, synthetic
    from pyvbmc_amd import gp as gpm
    from pyvbmc_amd.variational_optimization import _neg_elcbo

    rank, world, local_rank = comm.env_rank_world()
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        a.gpus = world
    ctx = _lib.Context(local_rank)
    _lib.set_default_context(ctx)
    comm.init_from_env(ctx)
    comm_rank, comm_world = ctx.comm_info()  # what RCCL itself reports

    wl = synthetic.make_workload(a.config, Ns_total=PER_GPU_NS[a.config])
    D, K = wl.D, wl.K
    nsk_job = wl.NsK * world                   # per-component samples of the whole job
    ns_job = nsk_job * K
    vp = VariationalPosterior(D, K)
    vp.mu, vp.sigma, vp.lambd = wl.mu.copy(), wl.sigma.reshape(1, -1), wl.lambd.reshape(-1, 1)
    vp.w, vp.eta = wl.w.reshape(1, -1), wl.eta.reshape(1, -1)
    gp = gpm.GP(D, gpm.SquaredExponential(), gpm.NegativeQuadratic(),
                gpm.GaussianNoise(constant_add=True, user_provided_add=wl.s2 is not None))
    gp.update(X_new=wl.X, y_new=wl.y, s2_new=wl.s2, hyp=wl.hyp)
    bnd = synthetic.default_theta_bnd(wl)
    theta = wl.theta.copy()

    step_no = [0]
    if a.rng == "philox":

        def step():
            step_no[0] += 1
            th = theta + 1e-9 * (step_no[0] % 7)  # a different parameter vector every call, as in Adam
            return _neg_elcbo(th, gp, vp, 0.0, nsk_job, True, False, bnd, rng="philox", seed=step_no[0])

    else:
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

        def step():
            step_no[0] += 1
            th[:] = theta + 1e-9 * (step_no[0] % 7)  # a different parameter vector every call
            ctx.check(ctx._lib.vbmc_neg_elcbo(ctx._h, _lib.ptr(th), th.size, C.byref(opts),
                                              C.byref(Fc), _lib.ptr(dF), C.byref(Gc), C.byref(Hc),
                                              None, None, None, None, None))
            return Fc.value, dF, Gc.value, Hc.value, 0

    predict_roofline = adam_loop = reference_stream = None
    if not a.no_secondary:
        # Secondary roofline (third way): gp.predict at the acquisition batch size,
        # its GEMM flops against the FP64 matrix-core peak.  Timed with HIP events around the
        # variance product alone (`frac`) and around the three predict launches (K*, product, finish).
        M_pred = 8192
        xs_pred = np.random.default_rng(7).standard_normal((M_pred, D))
        gp.predict(xs_pred, separate_samples=True)
        pm, vm = [], []
        ctx.set_timing(True)
        for _ in range(9):
            gp.predict(xs_pred, separate_samples=True)
            pm.append(ctx.last_kernel_ms(3))
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
            "kernel_ms_from": "HIP events around that launch alone (vbmc_last_kernel_ms which=5), median of 9",
            "all_launches_ms": pred_ms,
            "all_launches": "predict_kstar_mfma + predict_var_dma + predict_finish, HIP events around the three",
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
        ctx.comm_barrier()
        t1 = time.perf_counter()
        loop = minimize_adam_elbo(theta.copy(), gp, vp, nsk_job, bnd, **kw)
        ctx.comm_barrier()
        dt_loop = ctx.comm_max(time.perf_counter() - t1)
        adam_loop = {
            "iterations": n_loop,
            "us_per_iteration": 1e6 * dt_loop / n_loop,
            "evals_per_s": (n_loop / dt_loop) * (ns_job / PER_GPU_NS[a.config]),
            "F_first_last": [float(loop[3][0]), float(loop[3][-1])],
        }
        if world == 1:
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
            # (single evaluations of 15-50 ms occur); the mean is reported beside it
            dt_ref = float(np.median(t_ref))
            reference_stream = {
                "evals_per_s": 1.0 / dt_ref, "ms_per_eval": 1e3 * dt_ref, "evals": n_ref, "stat": "median",
                "ms_per_eval_mean": 1e3 * float(np.mean(t_ref)), "ms_per_eval_max": 1e3 * float(np.max(t_ref)),
                "what": "rng='numpy' (the default): the reference's np.random.randn stream of K*NsK/2*D normals, "
                        "restated bit for bit on the host cores (vbmc_set_eps_numpy: MT19937 recurrence on one "
                        "thread, polar method on all) + H2D copy per evaluation, PCIe-inclusive; dominated by "
                        "the host generator",
            }
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
    # lasts >= MIN_TIMED_S whatever K is (a 20-step region is 3 ms: too short to mean anything).
    # All ranks must agree on R: take the maximum estimate.
    est = ctx.comm_max(est)
    repeats = max(1, int(np.ceil(1.3 * MIN_TIMED_S / max(a.steps * est, 1e-9))))
    n_timed = repeats * a.steps
    ctx.comm_barrier()
    ctx.synchronize()
    # The main kernel's duration comes from HIP events carried by its own dispatch packet
    # (hipExtLaunchKernel: no barrier packet in the queue), read back on every SAMPLE_EVERY-th step
    # of the timed region together with the library's host-side breakdown; the other steps run
    # without any instrumentation call.
    SAMPLE_EVERY = 32
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
    F = out[0]
    if not np.isfinite(F):
        sys.exit("non-finite objective")
    plan = ctx.last_entmc_plan()

    ms_per_step = 1e3 * dt / n_timed
    evals_per_s = n_timed / dt
    value = evals_per_s * (ns_job / PER_GPU_NS[a.config])
    k_ms = float(np.mean(kern_ms))
    flops = algorithmic_flops(D, K, ns_job / world, grad=True)  # per launch (this rank's rows)
    achieved = flops / (k_ms * 1e-3) / 1e12
    achieved_e2e = flops / (ms_per_step * 1e-3) / 1e12
    # the draws the entropy kernel reads: resident, or (Philox mode) generated into HBM ahead of it
    # -- unless VBMC_ELBO_PREGEN=0 keeps the generation inside the entropy kernel
    eps_bytes = (ns_job / world / 2) * D * 8 if plan["resident_draws"] else 0.0
    traffic, traffic_from = None, None
    try:  # PMC-measured HBM bytes per launch (separate rocprofv3 --pmc passes, see profiles/README.md)
        tj = json.load(open(ROOT / "profiles" / "traffic.json"))
        ent = tj.get(f"config{a.config}", {}).get(a.rng)
        if ent:
            traffic = ent["hbm_bytes_per_launch"]
            traffic_from = {"file": "profiles/traffic.json", "source": ent.get("source"),
                            "collected": ent.get("collected"), "how": tj.get("_comment")}
    except (OSError, ValueError, KeyError):
        pass
    if a.config == 3:
        metric = "ELBO+entropy evals/sec at D=10, K=50, N=400, Ns=1e6 (1e6-sample-equivalent evals/s)"
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
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"BASELINE config {a.config} per GPU: D={D} K={K} N={wl.N} S=1, "
                        f"Ns={wl.Ns_total} MC samples per GPU (job Ns={ns_job}), value+grad _neg_elcbo "
                        f"with soft bounds, eps={a.rng}",
            "evals_per_s_job": evals_per_s,
            "parallelism": f"sample-sharded x{world}, 1 RCCL all-reduce/eval" if world > 1 else "single GPU",
            "timed_region": f"{repeats} x {a.steps} steps back to back (>= {MIN_TIMED_S} s), one barrier + "
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
            "kernel_ms_from": f"HIP events on the kernel's own dispatch, read on every {SAMPLE_EVERY}th of the {n_timed} "
                              f"timed steps ({len(kern_ms)} launches)",
            "algorithmic_flops_per_launch": flops,
            "hbm_bytes_per_launch_algorithmic": eps_bytes,
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
        "F": F,
        "host_us_per_step": dict(zip(["pack_upload", "launch", "wait_device", "finalize", "c_total"],
                                     (host_us / max(n_host, 1)).round(2).tolist())),
    }
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        nsk_cpu = a.cpu_sample_nsk or (wl.NsK // 5 if a.config == 5 else wl.NsK)
        res["cpu_baseline"] = cpu_baseline(wl, nsk_cpu, a.cpu_reps)
        res["speedup_vs_cpu_baseline"] = value / res["cpu_baseline"]["value"]
    if rank == 0:
        print(json.dumps(res))
    ctx.close()


if __name__ == "__main__":
    main()
