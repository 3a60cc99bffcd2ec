#!/usr/bin/env python
"""Benchmark of the ETDRK4 hot path on MI355X (contract: task statement, section 4).

    python bench.py --gpus 1 --steps K --warmup W

One "step" = one full ETDRK4 time step (4 stages, 36 fused 2-D transforms + budgets) of
CoupledModel 4096^2 fp64 (BASELINE.json configs[2]: LambDipole q, uniform phi, filter on), with the
state resident in HBM.  Prints ONE JSON line.  `roofline` is for the dominant kernel (k_x_products),
timed live with HIP events on the context's stream; `cpu_baseline` times the numpy oracle in its
reference-faithful mode (104 c2c transforms per step, 1 thread) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F0, NB, L = 1e-4, 0.01, 2 * np.pi * 200e3
MZ = 2 * np.pi / 280.0
K0 = 10 * (2 * np.pi / L)
U0 = 0.1
TE = 1.0 / (U0 * K0)

# algorithmic bytes per grid point ("bytes"): SURVEY 8(d) canonical figure per step, and the
# bytes one k_x_products launch must move (4 half-spectrum + 2 full inputs, 2 half + 1 full outputs: the two phi
# tendency sources of the canonical count leave the kernel as ONE array)
CANONICAL_B_PER_PT_STEP = {"coupled": 3136, "uncoupled": 2240, "qg": 848}
X_PRODUCTS_B_PER_PT = {"coupled": 4 * 8 + 2 * 16 + 2 * 8 + 16, "uncoupled": 3 * 8 + 3 * 16 + 2 * 8 + 16,
                       "qg": 3 * 8 + 2 * 8}
# algorithmic bytes per grid point and LAUNCH of the other fused kernels of the CoupledModel step:
#  s_phi: tendency in 16, phi and phi_y out 32, ETDRK4 state + coefficient planes 80/80/112/128 in the four stages
#         (mean 100), start-of-stage phih for the budget projections 12 (3 of 4 stages)
#  x_wavepv: phi, phi_y rows in 32, two half-spectrum rows out 16;  s_q: 2 half rows in 16, state + coefficients 50;
#  s_invert: 2 half rows in 16, q-hat 8, filter 4, four half rows out 32, psi-hat and qw-hat stored in the last stage 4
KERNEL_B_PER_PT = {"coupled": {"x_products": 96.0, "s_phi": 160.0, "x_wavepv": 48.0, "s_q": 66.0, "s_invert": 64.0}}
KERNEL_SYMBOL = {"x_products": "k_x_products", "s_phi": "k_s_phi", "x_wavepv": "k_x_wavepv", "s_q": "k_s_q",
                 "s_invert": "k_s_invert", "y_A": "k_y_A"}
HBM_PEAK_GBS = 8000.0


def measured_traffic(kernel_prefix):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary
    (profiles/r01_pmc_summary.json: FETCH_SIZE/WRITE_SIZE in separate passes, FETCH doubled as the gfx950
    note in MI355X_MICROARCH.md prescribes; same workload, same command).  None if the file is absent."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_summary.json")))
        for name, v in d["kernels"].items():
            if name.startswith(kernel_prefix) and "hbm_bytes_per_launch" in v:
                return v["hbm_bytes_per_launch"]
    except Exception:
        pass
    return None


def c3_kwargs(nx, model):
    dt = 0.025 * TE * 128 / nx
    kw = dict(L=L, nx=nx, tmax=1e30, dt=dt, twrite=10 ** 9, tdiags=10 ** 9, use_filter=True, U=-U0)
    if model == "qg":
        kw.update(nu4=7.5e8 / 64 if nx == 2048 else 5e11 * (128.0 / nx) ** 4, dt=0.05 * TE * 128 / nx)
    else:
        kw.update(m=MZ, N=NB, f=F0, nu4=5e11 * (128.0 / nx) ** 4, nu4w=0.0, nu=20, nuw=50.0, mu=0.0, muw=0.0)
    return kw


def build_model(model, nx, device):
    import niwqg_amd
    from niwqg_amd import InitialConditions as ic
    mod = {"coupled": niwqg_amd.CoupledModel, "uncoupled": niwqg_amd.UnCoupledModel, "qg": niwqg_amd.QGModel}[model]
    m = mod.Model(device=device, **c3_kwargs(nx, model))
    if model == "qg" and nx == 2048:
        q = 1e-5 * np.random.default_rng(0).standard_normal((nx, nx))
    else:
        q = ic.LambDipole(m, U=U0, R=2 * np.pi / K0)
    m.set_q(q)
    if model != "qg":
        m.set_phi((np.ones((nx, nx)) + 1j) * (2 * U0) / np.sqrt(2))
    return m


class _SlabCtxView(object):
    """What bench.py needs from a context, on top of a SlabRank."""

    def __init__(self, rank):
        from niwqg_amd import _lib
        self.r, self.L, self.h = rank, rank.L, rank.h
        self.budgets_enabled = rank.budgets
        self.KERNEL_CLASSES = _lib.Context.KERNEL_CLASSES
        for name in ("sync", "timer_start", "timer_stop", "profile_enable", "profile_read", "profile_read_all",
                     "device_bytes", "_chk"):
            setattr(self, name, getattr(_lib.Context, name).__get__(self))


def build_slab(model, nx, grp, local_rank):
    """One slab-decomposed simulation over all ranks of `grp` (niwqg_amd.slab); every rank prepares the same
    initial condition on the host and uploads its own column slab of the spectra."""
    import niwqg_amd
    from niwqg_amd import _lib, slab, InitialConditions as ic
    kw = c3_kwargs(nx, model)

    class G(object):          # the grid attributes InitialConditions.LambDipole reads
        pass
    g = G()
    g.nx = nx
    cell = (np.arange(nx) + 0.5) / nx * L
    g.x, g.y = np.meshgrid(cell, cell)
    dk = 2 * np.pi / L
    ll = dk * np.append(np.arange(0., nx / 2), np.arange(-nx / 2, 0.))
    kk = ll.copy() if model != "qg" else dk * np.arange(0., nx // 2 + 1)
    dx = L / nx
    wvx = np.sqrt((kk[None, :] * dx) ** 2. + (ll[:, None] * dx) ** 2.)
    filtr = np.exp(-23.6 * (wvx - 0.65 * np.pi) ** 4.)
    filtr[wvx <= 0.65 * np.pi] = 1.
    mid = {"coupled": _lib.COUPLED, "uncoupled": _lib.UNCOUPLED, "qg": _lib.QG}[model]
    phys = dict(U=kw["U"], nu=kw.get("nu", 0.0), nu4=kw["nu4"], mu=kw.get("mu", 0.0))
    if model != "qg":
        kappa2 = (kw["m"] * kw["f"] / kw["N"]) ** 2
        phys.update(f=kw["f"], kappa2=kappa2, nuw=kw["nuw"], nu4w=kw["nu4w"], muw=kw["muw"])
    ranks = slab.make_ranks(mid, nx, kk, ll, filtr, kw["dt"], grp.world, device=local_rank, only_rank=grp.rank,
                            budgets=True, **phys)
    sim = slab.SlabSimulation(ranks, slab.TorchTransport(grp.dist, stage_via_host=(getattr(grp, "backend", "") == "gloo")))
    if model == "qg" and nx == 2048:
        q = 1e-5 * np.random.default_rng(0).standard_normal((nx, nx))
    else:
        q = ic.LambDipole(g, U=U0, R=2 * np.pi / K0)
    sim.set_q_spectrum(np.fft.rfft2(q))
    if model != "qg":
        sim.set_phi_spectrum(np.fft.fft2((np.ones((nx, nx)) + 1j) * (2 * U0) / np.sqrt(2)))
    sim.sync()
    return sim, _SlabCtxView(ranks[0])


def cpu_baseline(model, nx_target, budget_s=20.0):
    """Reference-faithful numpy oracle (oracle/niwqg_oracle.py), one thread, on a bounded sample:
    the same model at a grid that finishes in ~20 s; steps/s is scaled to nx_target with N^2 log2 N."""
    try:
        import threadpoolctl
        limiter = threadpoolctl.threadpool_limits(1)
    except Exception:
        limiter = None
    from oracle import niwqg_oracle as O
    nx = 512 if model != "qg" else 1024
    kw = c3_kwargs(nx, model)
    kw.pop("twrite"), kw.pop("tdiags")
    if model == "qg":
        m = O.QGOracle(twrite=10 ** 9, tdiags=10 ** 9, **kw)
    else:
        m = O.NIWQGOracle(model, twrite=10 ** 9, tdiags=10 ** 9, coeff_chunk=16, **kw)
    m.set_q(O.lamb_dipole(m.grid, U=U0, R=2 * np.pi / K0))
    if model != "qg":
        m.set_phi((np.ones((nx, nx)) + 1j) * (2 * U0) / np.sqrt(2))
    m._step_forward()                      # includes the one-off tc==0 diagnostics tick
    t0, n = time.perf_counter(), 0
    while True:
        m._step_forward()
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 200:
            break
    sps = n / el
    scale = (nx_target ** 2 * np.log2(nx_target)) / (nx ** 2 * np.log2(nx))
    if limiter is not None:
        limiter.unregister() if hasattr(limiter, "unregister") else None
    return {"value": sps / scale, "unit": "steps/s", "cores": 1, "kind": "port",
            "sample": "%s oracle (numpy.fft, %d transforms/step) %d^2: %d steps in %.1f s = %.3f steps/s; "
                      "scaled by N^2 log2 N to %d^2" % (model, sum(m.fft_calls) // (n + 1), nx, n, el, sps, nx_target),
            "measured_steps_per_s_at_sample": sps, "sample_nx": nx, "host_cores_available": os.cpu_count()}


def bench_ensemble(args, grp, rank, world, local_rank):
    """BASELINE config 5: members sharded over the ranks, every member its own context and stream."""
    import torch
    from niwqg_amd import ensemble
    from niwqg_amd.distributed import aggregate_throughput
    nx = 1024 if args.nx == 4096 else args.nx
    ens = ensemble.Ensemble(lambda j: ensemble.config5_member(j, nx=nx, device=local_rank), args.members * world, rank, world)
    ens.step(args.warmup)
    grp.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ens.step(args.steps)
    grp.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    sps, wall = aggregate_throughput(grp, args.steps * len(ens.members), wall)
    if rank == 0:
        npts = float(nx) ** 2
        print(json.dumps({
            "metric": "member-steps/sec, ensemble of independent UnCoupledModel %d^2 members (BASELINE config 5)" % nx,
            "value": sps, "unit": "member-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * wall / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%d x UnCoupledModel %d^2 fp64, %d members per GPU, ETDRK4, filter on, budgets on"
                                   % (args.members * world, nx, args.members),
                       "parallelism": "members sharded over ranks, one HIP stream per member, no collective"},
            "roofline": {"bound": "hbm", "kernel": "whole step", "achieved": CANONICAL_B_PER_PT_STEP["uncoupled"] * npts * sps / 1e9,
                         "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                         "frac": CANONICAL_B_PER_PT_STEP["uncoupled"] * npts * sps / 1e9 / (HBM_PEAK_GBS * world), "traffic": None}}))
    grp.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)      # >= 100 timed, >= 20 warm-up steps
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--nx", type=int, default=4096)
    ap.add_argument("--model", default="coupled", choices=["coupled", "uncoupled", "qg"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-slab", action="store_true", help="use the slab path (and its collectives) even with one rank")
    ap.add_argument("--members", type=int, default=0, help="BASELINE config 5 instead of the headline: this many "
                    "independent UnCoupledModel 1024^2 members PER GPU (8 in the config), no collective")
    ap.add_argument("--replicas", action="store_true", help="with --gpus N > 1: N independent replicas instead of one "
                                                            "slab-decomposed simulation")
    args = ap.parse_args()

    from niwqg_amd.distributed import Group, aggregate_throughput
    import torch
    grp = Group(force=args.force_slab)   # nccl (= RCCL) when launched with WORLD_SIZE > 1
    rank, world, local_rank = grp.rank, grp.world, grp.local_rank
    local_rank %= max(torch.cuda.device_count(), 1)      # gloo rehearsal: several ranks share the one GPU of the box
    torch.cuda.set_device(local_rank)

    if args.members > 0:
        return bench_ensemble(args, grp, rank, world, local_rank)

    mode = "single GPU"
    slab_error = None
    sim = None
    if (world > 1 or (args.force_slab and grp.dist is not None)) and not args.replicas:
        # ONE simulation, slab-decomposed over the ranks: 4 all_to_all_single per stage over RCCL
        try:
            sim, ctx = build_slab(args.model, args.nx, grp, local_rank)
            sim.step(1)                             # exercises every collective once
            sim.sync()
            mode = "slab x%d, one all_to_all per transition (16 per step)" % world
        except Exception as e:                      # never lose the whole scaling run to a transport problem
            slab_error = "%s: %s" % (type(e).__name__, e)
            sim = None
        ok = grp.sum([1.0 if sim is not None else 0.0])[0]
        if ok < world:                              # all ranks take the same path
            if slab_error is None:
                slab_error = "another rank failed to set up the slab path"
            sim = None
    if sim is None:
        m = build_model(args.model, args.nx, local_rank)
        ctx = m._ctx
        if world > 1:
            mode = "replicas x%d (one full problem per GPU)" % world

    def advance(n):
        if sim is not None:
            sim.step(n)
        else:
            ctx.step(n)

    advance(args.warmup)
    ctx.sync()

    def barrier():
        grp.barrier()
        torch.cuda.synchronize()
        ctx.sync()

    per_class = hasattr(ctx, "profile_read_all")
    ctx.profile_enable(-2 if per_class else ctx.KERNEL_CLASSES["x_products"])      # HIP events around every launch
    barrier()
    t0 = time.perf_counter()
    ctx.timer_start()
    advance(args.steps)
    dev_ms = ctx.timer_stop()
    barrier()
    wall = time.perf_counter() - t0
    classes = None
    if per_class:
        classes = ctx.profile_read_all()
        launches, kms = classes["x_products"]
    else:
        launches, kms = ctx.profile_read()
    ctx.profile_enable(-1)
    if sim is not None:
        # strong scaling: all ranks advance the SAME simulation; whole-job steps/s = steps / slowest rank
        wall = grp.max(wall)
        sps = args.steps / wall
    else:
        sps, wall = aggregate_throughput(grp, args.steps, wall)   # all ranks' steps / max-over-ranks time

    extra = {}
    if sim is not None:
        # volume this rank hands to the all-to-alls per step (off-rank part), for the xGMI arithmetic in DESIGN.md 9
        per_stage = sum(t.numel() * 16 for t in sim.ranks[0].gx if t is not None)
        extra["exchange_GB_sent_per_rank_per_step"] = 4 * per_stage * (world - 1) / max(world, 1) / 1e9
        if world > 1:
            # the other way to use N GPUs (config 5 style): N independent simulations, no collective.  Timed AFTER
            # and OUTSIDE the K-step region above; reported as context only, never as `value`.
            try:
                sim.sync()
                m2 = build_model(args.model, args.nx, local_rank)
                m2._ctx.step(2)
                m2._ctx.sync()
                grp.barrier()
                t1 = time.perf_counter()
                m2._ctx.step(args.steps)
                m2._ctx.sync()
                grp.barrier()
                extra["replicas_aggregate_steps_per_s"] = aggregate_throughput(grp, args.steps, time.perf_counter() - t1)[0]
                del m2
            except Exception as e:
                extra["replicas_aggregate_steps_per_s"] = None
                extra["replicas_error"] = "%s: %s" % (type(e).__name__, e)

    if rank == 0:
        npts = float(args.nx) ** 2
        # the roofline object describes the DOMINANT kernel of the timed region: the class with the largest total time
        # among those whose algorithmic bytes are tabulated (the A sub-passes are separate launches of 0.1-0.2 ms each)
        dom = "x_products"
        table = KERNEL_B_PER_PT.get(args.model, {})
        if classes:
            cands = [k for k in classes if k in table and classes[k][0] > 0]
            if cands:
                dom = max(cands, key=lambda k: classes[k][1])
                launches, kms = classes[dom]
        k_ms = kms / max(launches, 1)
        k_bytes = (table[dom] if dom in table else X_PRODUCTS_B_PER_PT[args.model]) * npts
        if sim is not None:
            k_bytes /= world                       # each rank's launch covers nx/world rows
        achieved = k_bytes / (k_ms * 1e-3) / 1e9
        step_bytes = CANONICAL_B_PER_PT_STEP[args.model] * npts
        out = {
            "metric": "time-steps/sec, %sModel %d^2 fp64 (achieved HBM GB/s in roofline)" % (
                {"coupled": "Coupled", "uncoupled": "UnCoupled", "qg": "QG"}[args.model], args.nx),
            "value": sps, "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * wall / args.steps, "higher_is_better": True,
            "scaling": "strong" if sim is not None else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%sModel LambDipole %d^2 fp64, ETDRK4, filter on, budgets %s"
                                   % (args.model, args.nx, "on" if ctx.budgets_enabled else "off"),
                       "parallelism": mode, "slab_fallback_reason": slab_error,
                       "device_ms_per_step_hip_events": dev_ms / args.steps,
                       "device_bytes": ctx.device_bytes(), **extra},
            "roofline": {"bound": "hbm", "kernel": KERNEL_SYMBOL[dom], "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": measured_traffic(KERNEL_SYMBOL[dom] + "<") if (ctx.budgets_enabled and sim is None and args.nx == 4096) else None,
                         "launches": launches, "avg_launch_ms": k_ms, "algorithmic_bytes_per_launch": k_bytes,
                         "per_kernel_ms_per_step": ({k: round(v[1] / args.steps, 4) for k, v in classes.items()} if classes else None),
                         "per_kernel_frac_of_peak": ({k: round(table[k] * npts / (world if sim is not None else 1) / (classes[k][1] / classes[k][0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                                                      for k in table if classes and classes[k][0] > 0} if classes else None),
                         "step_canonical_bytes": step_bytes,
                         "step_achieved_GBs": step_bytes / (wall / args.steps) / 1e9,
                         "step_frac_of_peak": step_bytes / (wall / args.steps) / 1e9 / (HBM_PEAK_GBS * (world if sim is not None else 1))},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.model, args.nx)
        print(json.dumps(out))
    grp.close()


if __name__ == "__main__":
    main()
