#!/usr/bin/env python
"""Benchmark of the ETDRK4 hot path on MI355X (contract: task statement, section 4).

    python bench.py --gpus N --steps K --warmup W

One "step" = one full ETDRK4 time step (4 stages; 32 fused 2-D transforms + in-step budgets) of CoupledModel 4096^2
fp64 (BASELINE.json configs[2]: LambDipole q, uniform phi, filter on), state resident in HBM.  Prints ONE JSON line.

* `value` = K steps / wall time of the region bracketed by barrier + synchronize on both sides (max over ranks).
  The K steps are issued as 5 blocks with a HIP event between blocks (no synchronisation inside the region):
  `blocks_ms_per_step` / `median_block_steps_per_s` are SURVEY 8d's "median of 5".
* `roofline` describes the DOMINANT kernel class (largest total time among all six classes in an untimed pass with every
  launch bracketed by HIP events on the context's stream); INSIDE the timed region every 7th launch of that class is
  bracketed again (`--region-stride`; an event pair costs the stream ~9 us: all 44 launches of a step 4.6 % of it, all 20 of
  the dominant class ~1 %, and 10 % of a rank's step on eight slab ranks; 7 is coprime with the 20 launches of a step, so every
  position is sampled equally often), and achieved = algorithmic bytes per launch / the average duration of those launches.  `traffic` = HBM bytes per launch from the committed rocprofv3 PMC summary, used
  only if that summary was taken from the very sources that are running (sha256 of niwqg_amd/csrc + include/ stamped
  into it), else null.  `peak_measured_copy` = a 1r + 1w stream copy timed in the untimed part of the same run.
* `cpu_baseline` times the numpy oracle in its reference-faithful mode (104 c2c numpy.fft transforms per step, one thread)
  AT THE TARGET GRID in this run: one `_step_etdrk4` at 4096^2 (about 60 s; `--cpu-baseline-nx 2048` for a shorter, scaled
  sample).  `value` is always this run's own number; the round-2 recording rides along under its own key.
With --gpus N > 1 and no WORLD_SIZE in the environment the script starts N ranks itself (torch.distributed.run, child
process, before any GPU call); with WORLD_SIZE set it must equal --gpus.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
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

# canonical algorithmic bytes per grid point and STEP (2 passes x (read + write) per required transform,
# 64 B/pt complex, 32 B/pt real, + ETDRK4 state/coefficient streaming)
# (YBJModel is not in the survey's table: same convention, complex x3 inverse (phi, phix, phiy) + x2 forward per stage =
# 320 -> 1280, + the phi equation's 416 of ETDRK4 streaming)
CANONICAL_B_PER_PT_STEP = {"coupled": 3136, "uncoupled": 2240, "qg": 848, "ybj": 1696}
# Algorithmic bytes per grid point and LAUNCH of the six kernel classes as this design runs them.
#  x_products: 4 half-spectrum + 2 full rows in, 2 half + 1 full out (the phi tendency leaves as ONE array)
#  y_A       : in-place radix-S2 sub-pass, 32 B/pt per complex-plane equivalent (16 read + 16 written); per stage
#              W (1) + uq,vq (2 half = 1) + phi,phiy (2) + a,b (1) + u,psi,q,qw (4 half = 2) = 7 planes in 5 launches
#  s_phi     : tendency in 16, phi and phi_y out 32, ETDRK4 state + coefficient planes 80/80/112/128 in the four stages
#              (mean 100), start-of-stage phih for the budget projections 12 (3 of 4 stages); round 4: rows l and N - l share one
#              coefficient row (mirrored planes), i.e. the coefficient part (32/32/32/64, mean 40) counts half: 160 -> 140
#  x_wavepv  : phi, phi_y rows in 32, two half-spectrum rows out 16;  s_q: 2 half rows in 16, state + coefficients 50
#              (coefficients 20 of them: 66 -> 56 with mirrored rows)
#  s_invert  : 2 half rows in 16, q-hat 8, filter 4, four half rows out 32, psi-hat and qw-hat stored in the last stage 4
KERNEL_B_PER_PT = {
    "coupled": {"x_products": 96.0, "s_phi": 140.0, "x_wavepv": 48.0, "s_q": 56.0, "s_invert": 64.0, "y_A": 7 * 32.0 / 5},
    "uncoupled": {"x_products": 3 * 8 + 3 * 16 + 2 * 8 + 16, "s_phi": 140.0, "s_q": 56.0, "s_invert": 44.0,
                  "y_A": (1 + 1 + 2 + 1.5) * 32.0 / 4},
    "qg": {"x_products": 3 * 8 + 2 * 8, "s_q": 56.0, "s_invert": 44.0, "y_A": (1 + 1.5) * 32.0 / 2},
    "ybj": {"x_products": 3 * 8 + 3 * 16 + 2 * 8 + 16, "s_phi": 128.0, "y_A": (1 + 2) * 32.0 / 2},
}
KERNEL_SYMBOL = {"x_products": "k_x_products", "s_phi": "k_s_phi", "x_wavepv": "k_x_wavepv", "s_q": "k_s_q",
                 "s_invert": "k_s_invert", "y_A": "k_y_A"}
HBM_PEAK_GBS = 8000.0
GUIDE_COPY_GBS = 6290.0          # MI355X_MICROARCH.md: measured float4 copy, 79 % of the 8 TB/s data-sheet peak
PMC_SUMMARY = os.path.join(ROOT, "profiles", "pmc_summary.json")


def source_hash():
    """sha256 over the device sources and the C header: ties a PMC summary to the build it was measured on."""
    h = hashlib.sha256()
    files = [os.path.join(ROOT, "include", "niwqg_amd.h")]
    d = os.path.join(ROOT, "niwqg_amd", "csrc")
    files += sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith((".hip", ".hpp", ".h")))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def measured_traffic(symbol):
    """(bytes per launch of `symbol`, total bytes per step, note) from profiles/pmc_summary.json (rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE in separate passes, corrected as MI355X_MICROARCH.md prescribes; written by
    tools/pmc_summary.py), or (None, None, why) when the file is absent or was taken from other sources."""
    try:
        d = json.load(open(PMC_SUMMARY))
    except Exception:
        return None, None, "no profiles/pmc_summary.json"
    if d.get("source_sha256") != source_hash():
        return None, None, "profiles/pmc_summary.json was measured on other sources (hash mismatch)"
    per, step = None, d.get("step_hbm_bytes")
    tot, n = 0.0, 0
    for name, v in d.get("kernels", {}).items():
        if name.startswith(symbol + "<") and "hbm_bytes_per_launch" in v:
            w = v.get("launches_sampled", 1)
            tot += v["hbm_bytes_per_launch"] * w
            n += w
    if n:
        per = tot / n
    return per, step, "committed profile of these sources: " + d.get("note", "")[:120]


def c3_kwargs(nx, model):
    """SURVEY.md 8(d) "synthetic inputs per BASELINE config": C3 / C4 for the Kernel family at any nx; for QGModel C1 at 256^2
    (ref examples/LambDipole_qg.py:21-45: dt = 0.05 Te, nu4 = 7.5e8, use_filter=False) and C2 at 2048^2 (dt = 0.05 Te / 8,
    nu4 = 7.5e8 / 64, filter on); at other sizes the dipole workload with the round-1 scaling"""
    dt = 0.025 * TE * 128 / nx
    kw = dict(L=L, nx=nx, tmax=1e30, dt=dt, twrite=10 ** 9, tdiags=10 ** 9, use_filter=True, U=-U0)
    if model == "qg":
        if nx == 256:
            kw.update(nu4=7.5e8, dt=0.05 * TE, use_filter=False)
        elif nx == 2048:
            kw.update(nu4=7.5e8 / 64, dt=0.05 * TE / 8)
        else:
            kw.update(nu4=5e11 * (128.0 / nx) ** 4, dt=0.05 * TE * 128 / nx)
    else:
        kw.update(m=MZ, N=NB, f=F0, nu4=5e11 * (128.0 / nx) ** 4, nu4w=0.0, nu=20, nuw=50.0, mu=0.0, muw=0.0)
    return kw


def initial_fields(model, nx, grid):
    from niwqg_amd import InitialConditions as ic
    if model == "qg" and nx == 2048:                    # BASELINE config 2
        q = 1e-5 * np.random.default_rng(0).standard_normal((nx, nx))
    else:
        q = ic.LambDipole(grid, U=U0, R=2 * np.pi / K0)
    phi = None if model == "qg" else (np.ones((nx, nx)) + 1j) * (2 * U0) / np.sqrt(2)
    return q, phi


def lamb_dipole_rows(xs, ys, nx, U=U0, R=2 * np.pi / K0):
    """rows `ys` of InitialConditions.LambDipole (ref: niwqg/InitialConditions.py:77-114; same arithmetic per point) without
    ever forming the full plane: what a slab rank needs of the initial condition"""
    from scipy import special
    x, y = np.meshgrid(xs, ys)
    x0, y0 = xs[nx // 2], xs[nx // 2]
    r = np.sqrt((x - x0) ** 2 + (y - y0) ** 2)
    s = np.zeros_like(r)
    away = r != 0.
    s[away] = (y[away] - y0) / r[away]
    lam = 3.8317 / R
    C = -(2. * U * lam) / special.j0(lam * R)
    q = np.zeros_like(r)
    inside = r <= R
    q[inside] = C * special.j1(lam * r[inside]) * s[inside]
    return q


def build_model(model, nx, device, kind=None):
    import niwqg_amd
    mod = {"coupled": niwqg_amd.CoupledModel, "uncoupled": niwqg_amd.UnCoupledModel, "qg": niwqg_amd.QGModel,
           "ybj": niwqg_amd.YBJModel}[kind or model]
    m = mod.Model(device=device, slab=False, **c3_kwargs(nx, model))     # one whole problem on this GPU
    q, phi = initial_fields(model, nx, m)
    m.set_q(q)
    if phi is not None:
        m.set_phi(phi)
    return m


class _SlabCtxView(object):
    """What bench.py needs from a context, on top of a SlabRank."""

    def __init__(self, rank):
        from niwqg_amd import _lib
        self.r, self.L, self.h = rank, rank.L, rank.h
        self.budgets_enabled = rank.budgets
        self.KERNEL_CLASSES = _lib.Context.KERNEL_CLASSES
        for name in ("sync", "timer_start", "timer_stop", "profile_enable", "profile_stride", "profile_read", "profile_read_all",
                     "device_bytes", "event_record", "event_elapsed", "_chk"):
            setattr(self, name, getattr(_lib.Context, name).__get__(self))


def build_slab(model, nx, grp, local_rank, nchunks=2, kind=None, rank_of=0, peers=0):
    """One slab-decomposed simulation over all ranks of `grp` (niwqg_amd.slab).  Two stages so that the ranks can agree
    that every one of them got its memory BEFORE the first collective: allocate() then initialise().
    rank_of = P > 0: rank 0 of a P-rank decomposition ALONE on this GPU, with the library's null link (nq_slab_set_null_link):
    every launch, stream, event and row chunk of a real rank, nothing on the wire -- the compute term of DESIGN.md section 9.
    peers = P > 1: ALL P ranks in this one process, rank r on device r (modulo the devices there are): the blocks cross by peer
    copies over xGMI (SDMA, no CUs) instead of RCCL's send/recv kernels -- the A/B for the first real multi-GPU run."""
    nranks, myrank = (rank_of, 0) if rank_of else ((peers, None) if peers else (grp.world, grp.rank))
    from niwqg_amd import _lib, slab
    kw = c3_kwargs(nx, model)
    dk = 2 * np.pi / L
    ll = dk * np.append(np.arange(0., nx / 2), np.arange(-nx / 2, 0.))
    kk = ll.copy() if model != "qg" else dk * np.arange(0., nx // 2 + 1)
    dx = L / nx
    wvx = np.sqrt((kk[None, :] * dx) ** 2. + (ll[:, None] * dx) ** 2.)
    filtr = np.exp(-23.6 * (wvx - 0.65 * np.pi) ** 4.)
    filtr[wvx <= 0.65 * np.pi] = 1.
    if not kw["use_filter"]:
        filtr = np.ones_like(wvx)
    mid = {"coupled": _lib.COUPLED, "uncoupled": _lib.UNCOUPLED, "qg": _lib.QG}[model]
    budgets = True
    if kind == "ybj":                 # YBJModel: the UnCoupled workload, only phi is stepped, no budgets in the step
        mid, budgets = _lib.YBJ, False
    phys = dict(U=kw["U"], nu=kw.get("nu", 0.0), nu4=kw["nu4"], mu=kw.get("mu", 0.0))
    if model != "qg":
        kappa2 = (kw["m"] * kw["f"] / kw["N"]) ** 2
        phys.update(f=kw["f"], kappa2=kappa2, nuw=kw["nuw"], nu4w=kw["nu4w"], muw=kw["muw"])

    def allocate():
        dev = local_rank
        if peers:
            import torch
            ndev = max(torch.cuda.device_count(), 1)
            dev = [r % ndev for r in range(peers)]
        ranks = slab.make_ranks(mid, nx, kk, ll, filtr, kw["dt"], nranks, device=dev, only_rank=myrank,
                                budgets=budgets, torch_buffers=not (rank_of or peers), **phys)
        return ranks

    def initialise(ranks):
        # RCCL issued by the library itself (grouped send/recv per row chunk); with gloo the library calls back into
        # Python at every exchange
        if rank_of:
            sim = slab.SlabSimulation(ranks, "null", nchunks=nchunks)
        elif peers:
            sim = slab.SlabSimulation(ranks, "peers", nchunks=nchunks)
        else:
            sim = slab.connect(ranks, grp.dist, nchunks)

        nloc = ranks[0].nloc
        cell = (np.arange(nx) + 0.5) / nx * L
        noise = 1e-5 * np.random.default_rng(0).standard_normal((nx, nx)) if (model == "qg" and nx == 2048) else None

        class Rows(object):       # what SlabSimulation.set_q indexes: a global row range -> those rows, formed on demand
            def __init__(self, make):
                self.make = make

            def __getitem__(self, sl):
                assert (sl.stop - sl.start) == nloc and sl.start % nloc == 0
                return self.make(sl.start, sl.stop)
        if noise is not None:                                    # BASELINE config 2: seeded white noise, row by row
            sim.set_q(Rows(lambda a, b: noise[a:b]))
        else:
            sim.set_q(Rows(lambda a, b: lamb_dipole_rows(cell, cell[a:b], nx)))
        if model != "qg":
            sim.set_phi(Rows(lambda a, b: (np.ones((b - a, nx)) + 1j) * (2 * U0) / np.sqrt(2)))
        sim.sync()
        return sim, _SlabCtxView(ranks[0])

    return allocate, initialise


def _oracle_for(model, nx, table_workers):
    from oracle import niwqg_oracle as O
    kw = c3_kwargs(nx, model)
    kw.pop("twrite"), kw.pop("tdiags")
    if model == "qg":
        m = O.QGOracle(twrite=10 ** 9, tdiags=10 ** 9, coeff_chunk=8, table_workers=table_workers, **kw)
    else:
        m = O.NIWQGOracle(model, twrite=10 ** 9, tdiags=10 ** 9, coeff_chunk=8, table_workers=table_workers, **kw)
    if model == "qg" and nx == 2048:
        m.set_q(1e-5 * np.random.default_rng(0).standard_normal((nx, nx)))
    else:
        m.set_q(O.lamb_dipole(m.grid, U=U0, R=2 * np.pi / K0))
    if model != "qg":
        m.set_phi((np.ones((nx, nx)) + 1j) * (2 * U0) / np.sqrt(2))
    return m


def _time_oracle(m, budget_s, min_steps, max_steps=200):
    """times `_step_etdrk4` -- the hot path the GPU side times, without the diagnostics tick and the status line of
    `_step_forward` (negligible at tdiags = twrite = 1e9 anyway, except for the one-off tick at tc == 0)"""
    f0 = sum(m.fft_calls)
    t0, n = time.perf_counter(), 0
    while True:
        m._step_etdrk4()
        n += 1
        el = time.perf_counter() - t0
        if (el > budget_s and n >= min_steps) or n >= max_steps:
            break
    return n, el, (sum(m.fft_calls) - f0) // n


def cpu_baseline(model, nx_target, nx_sample=None, budget_s=20.0):
    """Reference-faithful numpy oracle (oracle/niwqg_oracle.py: numpy.fft, the reference's 104/72/33 transforms per step, ONE
    thread for the timed steps) MEASURED IN THIS RUN, on this box's host cores, AT THE TARGET GRID by default: at 4096^2 one
    `_step_etdrk4` is about 60 s, so the sample is exactly one step there (>= 2 steps or 20 s on smaller grids).  `value` is
    always what this run measured (scaled by N^2 log2 N only when --cpu-baseline-nx asks for a smaller sample grid).  The
    untimed constructor uses a thread pool for the coefficient tables.  Cross-checks that ride along under their own keys: a
    short 512^2 run against the N^2 log2 N law (`fit_check`), and the 4096^2 measurement recorded in round 2 on another box
    (`recorded_measurement_at_target_nx`, with its file)."""
    try:
        import threadpoolctl
        threadpoolctl.threadpool_limits(1)
    except Exception:
        pass
    cores = os.cpu_count() or 1
    tw = max(1, min(16, cores - 1))
    nx = nx_sample or nx_target

    def cost(n):
        return n ** 2 * np.log2(n)

    t_build = time.perf_counter()
    m = _oracle_for(model, nx, tw)
    t_build = time.perf_counter() - t_build
    n, el, nfft = _time_oracle(m, budget_s, 1 if nx >= 4096 else 2)
    del m
    sps = n / el
    out = {"value": sps * cost(nx) / cost(nx_target), "unit": "steps/s", "cores": 1, "kind": "port",
           "value_source": "measured in this run" if nx == nx_target else "measured in this run at %d^2, scaled by N^2 log2 N" % nx,
           "sample": "%s oracle (numpy.fft, %d transforms/step, 1 thread) MEASURED at %d^2 in this run: %d x _step_etdrk4 in %.1f s = %.4f steps/s%s"
                     % (model, nfft, nx, n, el, sps,
                        "" if nx == nx_target else "; scaled by N^2 log2 N (x%.2f) to %d^2" % (cost(nx_target) / cost(nx), nx_target)),
           "measured_steps_per_s_at_sample": sps, "sample_nx": nx, "sample_steps": n, "host_cores_available": cores,
           "untimed_setup_s": t_build}
    rec = os.path.join(ROOT, "profiles", "r02_bench_line_cpu_baseline_measured_at_4096.json")
    if model == "coupled" and nx_target == 4096 and os.path.exists(rec):
        try:
            r = json.load(open(rec))["cpu_baseline"]
            out["recorded_measurement_at_target_nx"] = {"steps_per_s": r["measured_steps_per_s_at_sample"], "file": os.path.relpath(rec, ROOT),
                                                        "host": "another box, round 2", "sample": r["sample"]}
        except Exception:
            pass
    if nx > 512:
        m = _oracle_for(model, 512, tw)
        n5, el5, _ = _time_oracle(m, 4.0, 2)
        s5 = n5 / el5
        out["fit_check"] = {"nx": 512, "measured_steps_per_s": s5, "predicted_at_sample_nx": s5 * cost(512) / cost(nx),
                            "measured_over_predicted_at_sample_nx": sps / (s5 * cost(512) / cost(nx))}
    return out


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


def bench_any_size(args, m, grp, rank, world):
    """Grids without a fused plan (niwqg_amd/_anysize.py: the reference's whole-plane sequence on the device, Bluestein / four-step
    transforms): wall-clock steps/s of `_step_etdrk4`, the whole step against the canonical bytes.  No per-kernel table: a step is
    hundreds to thousands of small library calls."""
    import torch
    from niwqg_amd.distributed import aggregate_throughput
    for _ in range(max(1, min(args.warmup, 3))):
        m._step_etdrk4()
    m._ctx.sync()
    grp.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        m._step_etdrk4()
    m._ctx.sync()
    grp.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    sps, wall = aggregate_throughput(grp, args.steps, wall)
    if rank == 0:
        npts = float(args.nx) ** 2
        step_bytes = CANONICAL_B_PER_PT_STEP[args.model] * npts
        gbs = step_bytes * sps / world / 1e9
        print(json.dumps({
            "metric": "time-steps/sec, %sModel %d^2 fp64, ANY-SIZE path (no fused plan for this grid)" % (
                {"coupled": "Coupled", "uncoupled": "UnCoupled", "qg": "QG", "ybj": "YBJ"}[args.model], args.nx),
            "value": sps, "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * wall / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%sModel %d^2 fp64, ETDRK4, the reference's whole-plane sequence on the device" % (args.model, args.nx),
                       "parallelism": "single GPU" if world == 1 else "replicas x%d" % world, "device_bytes": m._ctx.device_bytes()},
            "roofline": {"bound": "hbm", "kernel": "whole step (canonical bytes of SURVEY 8d)", "achieved": gbs, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None}}))
    grp.close()


def _lib_copy(ctx):
    """nq_stream_copy_gbs on a context view that does not wrap it (a slab rank)"""
    import ctypes
    v = ctypes.c_double()
    ctx._chk(ctx.L.nq_stream_copy_gbs(ctx.h, 512 << 20, 5, ctypes.byref(v)), "nq_stream_copy_gbs")
    return v.value


def relaunch_as_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD job (this process has not touched the
    GPU yet and never will) and pass rank 0's JSON line through."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


_watchdog = {"timer": None, "phase": None, "arm": None}


def watchdog(phase):
    """(re-)arm the multi-rank watchdog for the next phase of the run; None disarms it"""
    if _watchdog["arm"] is not None:
        _watchdog["arm"](phase)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)      # >= 100 timed, >= 20 warm-up steps
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--nx", type=int, default=4096)
    ap.add_argument("--model", default="coupled", choices=["coupled", "uncoupled", "qg", "ybj"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-nx", type=int, default=0, help="grid of the MEASURED CPU sample (default: nx itself)")
    ap.add_argument("--force-slab", action="store_true", help="use the slab path (and its collectives) even with one rank")
    ap.add_argument("--members", type=int, default=0, help="BASELINE config 5 instead of the headline: this many "
                    "independent UnCoupledModel 1024^2 members PER GPU (8 in the config), no collective")
    ap.add_argument("--chunks", type=int, default=2, help="row chunks per exchange of the slab path (1, 2, 4, 8)")
    ap.add_argument("--watchdog-seconds", type=int, default=600, help="multi-rank runs: abort (exit 124) when one phase "
                    "of the run makes no progress for this long (0 = off)")
    ap.add_argument("--rank-of", type=int, default=0, help="measure ONE rank of a P-rank slab decomposition alone on this GPU "
                    "(P = 2, 4, 8; no exchange, nq_slab_set_null_link): the per-rank compute term of the strong-scaling "
                    "arithmetic, next to the single-GPU step of the same run")
    ap.add_argument("--link", default="rccl", choices=["rccl", "peers"], help="with --gpus N > 1: 'rccl' = one process per GPU, grouped "
                    "ncclSend/ncclRecv issued by the library (the driver's launch); 'peers' = ONE process, rank r on device r, the "
                    "blocks cross by peer copies (SDMA over xGMI, no CUs): no torchrun, no RCCL")
    ap.add_argument("--rank-only", action="store_true", help="with --rank-of: skip the single-GPU step of the same run (so that a "
                    "kernel trace of the run holds the slab instantiations only)")
    ap.add_argument("--region-stride", type=int, default=7, help="inside the timed region bracket every n-th launch of the "
                    "dominant kernel class with HIP events (1 = every launch, as rounds 1-3 did)")
    ap.add_argument("--replicas", action="store_true", help="with --gpus N > 1: N independent replicas instead of one "
                                                            "slab-decomposed simulation")
    args = ap.parse_args()

    env_world = os.environ.get("WORLD_SIZE")
    peers_n = args.gpus if (args.link == "peers" and args.gpus > 1) else 0
    if peers_n and env_world is not None:
        sys.exit("bench.py: --link peers runs all ranks in ONE process: start it without a launcher")
    if env_world is None and args.gpus > 1 and not peers_n:
        sys.exit(relaunch_as_ranks(args.gpus))
    if env_world is not None and int(env_world) != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%s: launch with torch.distributed.run --nproc-per-node %d, or "
                 "run `python bench.py --gpus %d` without a launcher" % (args.gpus, env_world, args.gpus, args.gpus))

    if args.gpus > 1 and args.watchdog_seconds > 0:
        # A rank that waits for a peer which died (or an RCCL exchange that never completes) would sit there until the
        # launcher's own limit: give up loudly instead.  The timer is re-armed at every phase of the run.
        import threading

        def expired():
            sys.stderr.write("bench.py rank %s: no progress for %d s in phase '%s' -- giving up\n"
                             % (os.environ.get("RANK", "?"), args.watchdog_seconds, _watchdog["phase"]))
            sys.stderr.flush()
            os._exit(124)

        def arm(phase):
            if _watchdog["timer"] is not None:
                _watchdog["timer"].cancel()
            _watchdog["phase"] = phase
            if phase is not None:
                t = threading.Timer(args.watchdog_seconds, expired)
                t.daemon = True
                t.start()
                _watchdog["timer"] = t
        _watchdog["arm"] = arm
    watchdog("set-up")

    from niwqg_amd.distributed import Group, aggregate_throughput
    import torch
    grp = Group(force=args.force_slab)   # nccl (= RCCL) when launched with WORLD_SIZE > 1
    rank, world, local_rank = grp.rank, grp.world, grp.local_rank
    local_rank %= max(torch.cuda.device_count(), 1)      # gloo rehearsal: several ranks share the one GPU of the box
    torch.cuda.set_device(local_rank)

    if args.members > 0:
        return bench_ensemble(args, grp, rank, world, local_rank)

    phys_model = "uncoupled" if args.model == "ybj" else args.model       # YBJ: the UnCoupled workload, phi-only stepping
    mode = "single GPU"
    sim = None
    if args.rank_of and world > 1:
        sys.exit("bench.py: --rank-of measures one rank alone: run it with --gpus 1")
    if args.rank_of or peers_n or ((world > 1 or (args.force_slab and grp.dist is not None)) and not args.replicas):
        # ONE simulation, slab-decomposed over the ranks.  Allocation is the only step allowed to fail
        # softly: the ranks agree on it BEFORE the first collective; from then on any error is fatal (a rank that
        # dropped out of a collective sequence cannot be recovered from inside the job).
        allocate, initialise = build_slab(phys_model, args.nx, grp, local_rank, args.chunks, kind=args.model, rank_of=args.rank_of, peers=peers_n)
        err, ranks = None, None
        try:
            ranks = allocate()
        except (RuntimeError, MemoryError) as e:
            err = "%s: %s" % (type(e).__name__, e)
        ok = grp.sum([1.0 if ranks is not None else 0.0])[0]
        if ok < world:
            sys.stderr.write("bench.py rank %d: slab allocation failed on %d of %d ranks (%s)\n"
                             % (rank, world - int(ok), world, err or "this rank was fine"))
            grp.close()
            sys.exit(3)
        sim, ctx = initialise(ranks)
        mode = "slab x%d: %s" % (args.rank_of or peers_n or world, sim.describe())
    if sim is None:
        m = build_model(phys_model, args.nx, local_rank, kind=args.model)
        ctx = m._ctx
        if world > 1:
            mode = "replicas x%d (one full problem per GPU)" % world
        if getattr(m, "_any_size", False):
            return bench_any_size(args, m, grp, rank, world)

    def advance(n):
        if sim is not None:
            sim.step(n)
        else:
            ctx.step(n)

    watchdog("warm-up")
    advance(args.warmup)
    ctx.sync()
    # what a plain 1-read + 1-write copy kernel reaches on THIS device (512 MiB, best of 5): the second denominator of the
    # roofline fractions; untimed part of the run
    copy_gbs = None
    try:
        copy_gbs = ctx.stream_copy_gbs(512 << 20, 5) if hasattr(ctx, "stream_copy_gbs") else _lib_copy(ctx)
    except Exception as e:                                  # pragma: no cover - e.g. not enough free memory for 1 GiB
        sys.stderr.write("bench.py: stream-copy measurement skipped (%s)\n" % e)
    watchdog("timed region")

    def barrier():
        # the library's streams first (nq_sync drains the compute AND the exchange stream): torch's communicator must never
        # start a collective while the library's own still has sends / receives queued on the same device
        (sim.sync if sim is not None else ctx.sync)()       # (every rank this process drives: all of them with --link peers)
        torch.cuda.synchronize()
        grp.barrier()
        torch.cuda.synchronize()
        (sim.sync if sim is not None else ctx.sync)()

    NBLK = 5 if args.steps >= 5 else 1
    blocks = [args.steps // NBLK + (1 if i < args.steps % NBLK else 0) for i in range(NBLK)]
    # HIP events around kernel launches.  Bracketing EVERY launch of the six classes inside the timed region costs 4.6 % of a
    # 4096^2 step (88 event records per step at ~4.5 us each: measured 101.8 against 106.5 steps/s on one box, round 3), a
    # quarter of a 256^2 step.  So: nx >= 4096 -- an untimed pass brackets all six classes (the per-class table, and which
    # class is dominant), then INSIDE the timed region only the launches of that dominant class are bracketed: the roofline's
    # kernel is measured live over the timed region (20 of the 44 launches of a step at most); small grids -- everything in
    # a separate pass right after the timed region.
    events_in_region = args.nx >= 4096
    classes_all, ksteps_all = None, 0
    if events_in_region:
        ksteps_all = min(args.steps, 5)
        ctx.profile_enable(-2)
        advance(ksteps_all)                                 # creates the event pool
        ctx.profile_read_all()
        advance(ksteps_all)                                 # untimed, all six classes bracketed: the per-class table
        classes_all = ctx.profile_read_all()
        live = [k for k in classes_all if classes_all[k][0] > 0]
        dom_class = max(live, key=lambda k: classes_all[k][1])
        ctx.profile_enable(ctx.KERNEL_CLASSES[dom_class])   # timed region: the dominant class only ...
        # ... and of its launches every REGION_STRIDE-th: an event pair costs the stream ~9 us, i.e. 2 % of a 4096^2 step with
        # all 20 A sub-passes of a step bracketed, 10 % of a rank's step on eight slab ranks (round 4).  7 is coprime with the 5
        # launches per stage / 20 per step of that class, so every position of the step is sampled equally often.
        ctx.profile_stride(args.region_stride)
    xt = None
    if sim is not None:
        # the exchange stream's own time (two event records per exchange) is taken in a short UNTIMED pass, not in the region
        sim.counters(reset=2)
        advance(min(args.steps, 5))
        sim.sync()
        xt = sim.counters()
        sim.counters(reset=1)                               # the region itself: count host calls / exchange chunks / bytes only
    barrier()
    t0 = time.perf_counter()
    ctx.event_record(0)
    for i, nb in enumerate(blocks):
        advance(nb)
        ctx.event_record(i + 1)
    barrier()
    wall = time.perf_counter() - t0
    block_ms = [ctx.event_elapsed(i, i + 1) / nb for i, nb in enumerate(blocks)]
    dev_ms = ctx.event_elapsed(0, NBLK)
    cnt = sim.counters() if sim is not None else None
    ksteps = args.steps
    if not events_in_region:
        ksteps = max(1, min(args.steps, 20))
        ctx.profile_enable(-2)
        advance(ksteps)
        ctx.profile_read_all()                              # first pass creates the events, second one is read
        advance(ksteps)
    classes = ctx.profile_read_all()
    ctx.profile_enable(-1)
    ctx.profile_stride(1)
    dom_timed = None
    if events_in_region:                                    # `classes` holds the dominant class over the timed region only
        dom_timed = (dom_class, classes[dom_class])
        classes, ksteps = classes_all, ksteps_all
    if sim is not None:
        # strong scaling: all ranks advance the SAME simulation; whole-job steps/s = steps / slowest rank
        wall = grp.max(wall)
        sps = args.steps / wall
    else:
        sps, wall = aggregate_throughput(grp, args.steps, wall)   # all ranks' steps / max-over-ranks time

    extra = {}
    if sim is not None:
        nst = max(cnt["steps"], 1)
        # rank 0's step, taken apart with HIP events on the two streams: kernels on the compute stream (every launch of the
        # six classes bracketed), the rest of the compute stream's time = it waited for an exchange (or a launch gap), and
        # what the exchange stream itself spent in exchanges / all-reduces (these overlap with the kernels when the
        # chunking works: exchange_ms > exposed_exchange_ms is the hidden part)
        compute_ms = sum(v[1] for v in classes.values()) / ksteps
        extra.update(host_dispatches_per_step=cnt["host_calls"] / nst,
                     exchange_chunks_per_step=cnt["exchange_chunks"] / nst,
                     exchange_ms_per_step=xt["exchange_ms"] / max(xt["steps"], 1),          # (untimed pass right before the region)
                     allreduce_ms_per_step=xt["allreduce_ms"] / max(xt["steps"], 1),
                     compute_ms_per_step=compute_ms,
                     exposed_exchange_ms_per_step=dev_ms / args.steps - compute_ms,
                     exchange_GB_sent_per_rank_per_step=cnt["bytes_sent"] / nst / 1e9)
        if world > 1:
            # the other way to use N GPUs (config 5 style): N independent simulations, no collective.  Timed AFTER
            # and OUTSIDE the K-step region above; reported as context only, never as `value`.
            sim.sync()
            watchdog("replica context")
            m2 = build_model(phys_model, args.nx, local_rank, kind=args.model)
            m2._ctx.step(2)
            m2._ctx.sync()
            grp.barrier()
            t1 = time.perf_counter()
            m2._ctx.step(args.steps)
            m2._ctx.sync()
            grp.barrier()
            extra["replicas_aggregate_steps_per_s"] = aggregate_throughput(grp, args.steps, time.perf_counter() - t1)[0]
            del m2

    if sim is not None and args.rank_of and args.rank_only:
        extra.update(rank_of=args.rank_of, rank_compute_ms_per_step=dev_ms / args.steps)
    elif sim is not None and args.rank_of:
        # the single-GPU step on the same box in the same run: what the rank's compute is held against (gate: <= 1.15 x step / P)
        sim.sync()
        m1 = build_model(phys_model, args.nx, local_rank, kind=args.model)
        m1._ctx.step(3)
        m1._ctx.sync()
        t1 = time.perf_counter()
        n1 = max(5, min(args.steps, 20))
        m1._ctx.step(n1)
        m1._ctx.sync()
        single_ms = 1e3 * (time.perf_counter() - t1) / n1
        rank_ms = dev_ms / args.steps
        extra.update(rank_of=args.rank_of, single_gpu_ms_per_step_same_run=single_ms,
                     rank_compute_ms_per_step=rank_ms, ideal_ms_per_step=single_ms / args.rank_of,
                     rank_compute_over_ideal=rank_ms / (single_ms / args.rank_of))
        del m1

    if rank == 0:
        npts = float(args.nx) ** 2
        share = (args.rank_of or peers_n or world) if sim is not None else 1     # a slab rank's launch covers 1/P of the grid
        table = KERNEL_B_PER_PT[args.model]
        cands = [k for k in classes if k in table and classes[k][0] > 0]
        dom = max(cands, key=lambda k: classes[k][1])
        launches, kms = classes[dom]
        if dom_timed is not None and dom_timed[0] in table and dom_timed[1][0] > 0:
            dom, (launches, kms) = dom_timed                # measured live inside the timed region
        k_ms = kms / launches
        k_bytes = table[dom] * npts / share
        achieved = k_bytes / (k_ms * 1e-3) / 1e9
        step_bytes = CANONICAL_B_PER_PT_STEP[args.model] * npts
        design_bytes = sum(table[k] * npts / share * classes[k][0] for k in cands) / ksteps * share
        traffic, pmc_step, pmc_note = (None, None, "PMC summary is for the default workload only")
        if sim is None and args.nx == 4096 and args.model == "coupled" and ctx.budgets_enabled:
            traffic, pmc_step, pmc_note = measured_traffic(KERNEL_SYMBOL[dom])
        real_bytes = pmc_step if pmc_step else design_bytes
        s_per_step = wall / args.steps
        peak = HBM_PEAK_GBS * share
        out = {
            "metric": ("rank-compute steps/sec of ONE rank of %d alone (no exchange; NOT a simulation rate), " % args.rank_of if args.rank_of else "")
                      + "time-steps/sec, %sModel %d^2 fp64 (achieved HBM GB/s in roofline)" % (
                {"coupled": "Coupled", "uncoupled": "UnCoupled", "qg": "QG", "ybj": "YBJ"}[args.model], args.nx),
            "value": sps, "unit": "steps/s", "n_gpus": peers_n or world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * s_per_step, "higher_is_better": True,
            # --gpus N runs ONE simulation of the same grid on N slab ranks: the series 1, 2, 4, 8 is strong scaling (total work fixed),
            # its N = 1 member included; only --replicas (one full problem per GPU) is weak
            "scaling": "weak" if (args.replicas and world > 1) else "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%sModel %s %d^2 fp64, ETDRK4, filter %s, budgets %s"
                                   % (args.model, "random-q" if (args.model == "qg" and args.nx == 2048) else "LambDipole",
                                      args.nx, "on" if c3_kwargs(args.nx, phys_model)["use_filter"] else "off",
                                      "on" if ctx.budgets_enabled else "off"),
                       "parallelism": mode,
                       "device_ms_per_step_hip_events": dev_ms / args.steps,
                       "blocks_ms_per_step": [round(b, 5) for b in block_ms],
                       "median_block_steps_per_s": 1e3 / float(np.median(block_ms)),
                       "device_bytes": ctx.device_bytes(), **extra},
            "roofline": {"bound": "hbm", "kernel": KERNEL_SYMBOL[dom], "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": pmc_note,
                         "launches": launches, "avg_launch_ms": k_ms, "algorithmic_bytes_per_launch": k_bytes,
                         "kernel_events": ("every %d-th launch of the dominant class bracketed inside the timed region (%d launches sampled, all "
                                           "positions of the step equally often); per-class table from an untimed pass of %d steps right before "
                                           "it (an event pair costs the stream ~9 us: all six classes in the region 4.6 %% of the step, all launches "
                                           "of the dominant class 2 %%)" % (args.region_stride, launches, ksteps)) if events_in_region
                                          else "separate pass of %d steps after the timed region" % ksteps,
                         "per_kernel_ms_per_step": {k: round(v[1] / ksteps, 4) for k, v in classes.items()},
                         "per_kernel_frac_of_peak": {k: round(table[k] * npts / share / (classes[k][1] / classes[k][0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                                                     for k in cands},
                         "step_canonical_bytes": step_bytes,
                         "step_achieved_GBs": step_bytes / s_per_step / 1e9,
                         "step_frac_of_peak": step_bytes / s_per_step / 1e9 / peak,
                         "step_real_bytes": real_bytes,
                         "step_real_source": "rocprofv3 PMC (profiles/pmc_summary.json)" if pmc_step else "design table x launches",
                         "step_real_GBs": real_bytes / s_per_step / 1e9,
                         "step_real_frac": real_bytes / s_per_step / 1e9 / peak,
                         # second denominator: the 1r + 1w stream-copy rate measured on this device in this run
                         "peak_guide_copy": GUIDE_COPY_GBS,
                         "frac_of_guide_copy": achieved / GUIDE_COPY_GBS,
                         "step_frac_of_guide_copy": step_bytes / s_per_step / 1e9 / (GUIDE_COPY_GBS * share),
                         "step_real_frac_of_guide_copy": real_bytes / s_per_step / 1e9 / (GUIDE_COPY_GBS * share),
                         "peak_measured_copy": copy_gbs,
                         "frac_of_copy": (achieved / copy_gbs) if copy_gbs else None,
                         "step_frac_of_copy": (step_bytes / s_per_step / 1e9 / (copy_gbs * share)) if copy_gbs else None,
                         "step_real_frac_of_copy": (real_bytes / s_per_step / 1e9 / (copy_gbs * share)) if copy_gbs else None},
        }
        if world == 1 and not args.no_cpu_baseline and not args.rank_of and not peers_n:
            out["cpu_baseline"] = cpu_baseline(phys_model, args.nx, nx_sample=args.cpu_baseline_nx or None)
        print(json.dumps(out))
    watchdog("shutdown")
    grp.close()
    watchdog(None)


if __name__ == "__main__":
    main()
