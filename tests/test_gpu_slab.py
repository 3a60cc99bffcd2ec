"""Slab decomposition on ONE GPU: P peer ranks in one process (the library's "peers" link: device copies for the wire).
Everything except the wire itself is the code that runs with one process per GPU: the step inside the library
(nq_slab_step: phases, chunked exchanges on a second stream, events), blocked-row addressing in the row kernels,
column-slab geometry in the spectral kernels, set_q / set_phi from each rank's own rows, the cross-rank sums of the
budget integrals and of the diagnostics tick."""
import numpy as np
import pytest

from oracle import niwqg_oracle as O
from test_oracle_golden import notebook_kwargs, rel, L, K0, U0

pytestmark = pytest.mark.gpu


def setup_case(kind, nx, use_filter=True):
    from niwqg_amd import _lib
    kw = notebook_kwargs(nx, use_filter)
    if kind == "qg":
        o = O.QGOracle(L=L, nx=nx, tmax=1e30, dt=kw["dt"], twrite=10 ** 9, nu4=kw["nu4"], nu=5.0, mu=1e-8,
                       use_filter=use_filter, U=-U0, tdiags=10 ** 9, beta=2e-11)
        phys = dict(U=-U0, nu=5.0, nu4=kw["nu4"], mu=1e-8, beta=2e-11)
        model = _lib.QG
    else:
        o = O.NIWQGOracle(kind, **kw)
        phys = dict(U=kw["U"], f=kw["f"], kappa2=o.kappa2, nu=kw["nu"], nu4=kw["nu4"], mu=kw["mu"], nuw=kw["nuw"],
                    nu4w=kw["nu4w"], muw=kw["muw"])
        model = {"coupled": _lib.COUPLED, "uncoupled": _lib.UNCOUPLED}[kind]
    q0 = O.lamb_dipole(o.grid, U=U0, R=2 * np.pi / K0)
    phi0 = None if kind == "qg" else 0.2 * O.wave_packet(o.grid, k=2 * K0, l=K0, R=L / 6, x0=L / 2, y0=L / 2)
    return model, o, kw["dt"], phys, q0, phi0


@pytest.mark.parametrize("kind", ["coupled", "uncoupled", "qg"])
@pytest.mark.parametrize("nranks,nchunks", [(2, 1), (2, 2), (4, 4), (8, 2)])
def test_virtual_ranks_match_single_context(kind, nranks, nchunks):
    from niwqg_amd import _lib, slab
    nx, nsteps = (256 if nranks < 8 else 512), 3        # a rank needs at least 64 columns
    model, o, dt, phys, q0, phi0 = setup_case(kind, nx)
    # single context
    one = _lib.Context(model, nx, o.kk, o.ll, o.filtr, dt, budgets=True, **phys)
    one.set_q(q0)
    if phi0 is not None:
        one.set_phi(phi0)
    one.take_budget_increments() if model != _lib.QG else one.scalar(_lib.S_KE)
    one.step(nsteps)
    qh1 = one.field(_lib.F_QH)
    phih1 = one.field(_lib.F_PHIH) if phi0 is not None else None
    inc1 = one.take_budget_increments() if model != _lib.QG else (one.scalar(_lib.S_KE),)
    # P virtual ranks
    ranks = slab.make_ranks(model, nx, o.kk, o.ll, o.filtr, dt, nranks, budgets=True, **phys)
    sim = slab.SlabSimulation(ranks, "peers", nchunks=nchunks)
    assert sim.counters()["nchunks"] == nchunks
    sim.set_q(q0)                          # physical fields: every rank transforms its own rows on the device
    if phi0 is not None:
        sim.set_phi(phi0)
    for r in ranks:
        r.budget_increments()
    sim.counters(reset=1)
    sim.step(2)
    sim.step(nsteps - 2)                   # two host calls for three steps
    sim.sync()
    cnt = sim.counters()
    assert cnt["host_calls"] == 2 and cnt["steps"] == nsteps
    groups = {"coupled": 4, "uncoupled": 3, "qg": 2}[kind]
    assert cnt["exchange_chunks"] == nsteps * 4 * groups * nchunks
    qhP = sim.gather_qh()
    assert qhP.shape == qh1.shape
    assert rel(qhP, qh1) < 1e-13
    if phi0 is not None:
        assert rel(sim.gather_phih(), phih1) < 1e-13
    for r in ranks:                      # every rank ends up with the global increments
        inc = r.budget_increments()
        assert np.allclose(inc[:len(inc1)], inc1, rtol=1e-10, atol=1e-30), (r.rank, inc, inc1)
    # physical rows of every rank against the single context's fields
    assert rel(sim.gather_rows(_lib.F_Q), one.field(_lib.F_Q)) < 1e-13
    assert rel(sim.gather_rows(_lib.F_U), one.field(_lib.F_U)) < 1e-12
    assert rel(sim.gather_rows(_lib.F_V), one.field(_lib.F_V)) < 1e-12
    assert rel(sim.gather_rows(_lib.F_P), one.field(_lib.F_P)) < 1e-12
    assert abs(sim.cfl_max() - one.scalar(_lib.S_CFL)) < 1e-12 * one.scalar(_lib.S_CFL)
    if phi0 is not None:
        assert rel(sim.gather_rows(_lib.F_PHI), one.field(_lib.F_PHI)) < 1e-13
        assert rel(sim.gather_rows(_lib.F_PHIX), one.field(_lib.F_PHIX)) < 1e-12
        assert rel(sim.gather_rows(_lib.F_PHIY), one.field(_lib.F_PHIY)) < 1e-12
        d1, dP = one.diagnostic_sums(), sim.diagnostics()
        assert np.allclose(dP, d1, rtol=1e-10, atol=1e-13 * np.abs(d1).max()), (dP, d1)
    if kind == "coupled":
        assert rel(sim.gather_rows(_lib.F_QW), one.field(_lib.F_QW)) < 1e-11


def test_virtual_ranks_against_the_oracle_with_quirk_q2():
    """set_q before set_phi (wave-free psi in the first stage), then compare with the reference-pinned oracle."""
    from niwqg_amd import slab
    nx, nranks = 256, 4
    model, o, dt, phys, q0, phi0 = setup_case("coupled", nx, use_filter=False)
    o.set_q(q0)
    o.set_phi(phi0)
    for _ in range(3):
        o._step_forward()
    ranks = slab.make_ranks(model, nx, o.kk, o.ll, o.filtr, dt, nranks, budgets=False, **phys)
    sim = slab.SlabSimulation(ranks, "peers", nchunks=2)
    sim.set_q(q0)
    sim.set_phi(phi0)
    sim.step(3)
    sim.sync()
    assert rel(sim.gather_phih(), o.phih) < 1e-12
    qh = sim.gather_qh()
    assert rel(np.fft.irfft2(qh), o.q) < 1e-12


def test_rccl_link_with_one_rank():
    """The RCCL link end to end with the only world size a one-GPU box can host: unique id, ncclCommInitRank, the
    grouped send/recv path (no peers: the own block crosses by a device copy), ncclAllReduce of the budget sums, the
    exchange stream and its events.  A one-rank slab context keeps SEPARATE x-side and y-side buffers, so every exchange
    really copies."""
    import os
    import torch
    import torch.distributed as dist
    from conftest import free_port
    from niwqg_amd import _lib, slab
    nx, nsteps = 256, 3
    model, o, dt, phys, q0, phi0 = setup_case("coupled", nx)
    one = _lib.Context(model, nx, o.kk, o.ll, o.filtr, dt, budgets=True, **phys)
    one.set_q(q0)
    one.set_phi(phi0)
    one.take_budget_increments()
    one.step(nsteps)
    created = not dist.is_initialized()
    if created:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ["MASTER_PORT"] = str(free_port())
        dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        ranks = slab.make_ranks(model, nx, o.kk, o.ll, o.filtr, dt, 1, only_rank=0, budgets=True, **phys)
        sim = slab.SlabSimulation(ranks, "rccl", dist=dist, nchunks=2)
        sim.set_q(q0)
        sim.set_phi(phi0)
        ranks[0].budget_increments()
        sim.counters(reset=2)                  # with exchange timing
        sim.step(nsteps)
        sim.sync()
        cnt = sim.counters()
        assert cnt["exchange_chunks"] == nsteps * 16 * 2 and cnt["exchange_ms"] > 0 and cnt["bytes_sent"] == 0
        assert rel(sim.gather_qh(), one.field(_lib.F_QH)) < 1e-13
        assert rel(sim.gather_phih(), one.field(_lib.F_PHIH)) < 1e-13
        assert np.allclose(ranks[0].budget_increments(), one.take_budget_increments(), rtol=1e-10, atol=1e-30)
    finally:
        if created:
            dist.destroy_process_group()


TWO_PROCESS_WORKER = """
import sys
sys.path.insert(0, %r)
sys.path.insert(0, %r)
import numpy as np
import torch
from niwqg_amd.distributed import Group
from niwqg_amd import _lib, slab
from test_gpu_slab import setup_case, rel

g = Group(backend="gloo")                      # both processes share GPU 0; the wire is host-staged gloo
nx, nsteps = 256, 3
model, o, dt, phys, q0, phi0 = setup_case("coupled", nx)
ranks = slab.make_ranks(model, nx, o.kk, o.ll, o.filtr, dt, g.world, device=0, only_rank=g.rank, budgets=True,
                        torch_buffers=True, **phys)
sim = slab.SlabSimulation(ranks, "callback", dist=g.dist, stage_via_host=True)
sim.set_q(q0)
sim.set_phi(phi0)
ranks[0].budget_increments()
sim.step(nsteps)
sim.sync()
inc = ranks[0].budget_increments()
dsum = sim.diagnostics()
cfl = sim.cfl_max()
mine = [torch.from_numpy(ranks[0].download(0)), torch.from_numpy(ranks[0].download(1))]
parts = [None] * g.world
g.dist.all_gather_object(parts, (g.rank, mine[0].numpy(), mine[1].numpy(), inc))
if g.rank == 0:
    parts.sort(key=lambda t: t[0])
    qh = np.concatenate([t[1] for t in parts], axis=1)
    phih = np.concatenate([t[2] for t in parts], axis=1)
    one = _lib.Context(model, nx, o.kk, o.ll, o.filtr, dt, budgets=True, **phys)
    one.set_q(q0)
    one.set_phi(phi0)
    one.take_budget_increments()
    one.step(nsteps)
    assert rel(qh, one.field(_lib.F_QH)) < 1e-13, rel(qh, one.field(_lib.F_QH))
    assert rel(phih, one.field(_lib.F_PHIH)) < 1e-13
    inc1 = one.take_budget_increments()
    for t in parts:
        assert np.allclose(t[3], inc1, rtol=1e-10, atol=1e-30), (t[0], t[3], inc1)
    d1 = one.diagnostic_sums()
    assert np.allclose(dsum, d1, rtol=1e-10, atol=1e-13 * np.abs(d1).max()), (dsum, d1)
    assert abs(cfl - one.scalar(_lib.S_CFL)) < 1e-12 * cfl
    print("two processes agree with one context")
g.close()
"""


def test_two_processes_one_gpu_host_staged_collectives(tmp_path):
    """The real multi-process driver (one SlabRank per process, the step inside the library) with two processes sharing
    the one GPU of the test box; RCCL refuses two ranks on one device, so the library calls back into Python at every
    exchange and gloo carries the buffers through host memory (the "callback" link)."""
    from conftest import free_port
    port = free_port()
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "two_proc.py"
    script.write_text(TWO_PROCESS_WORKER % (root, os.path.join(root, "tests")))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert "two processes agree with one context" in out.stdout


COPY_PATH_WORKER = r"""
import sys
sys.path.insert(0, %r)
sys.path.insert(0, %r)
import numpy as np
from niwqg_amd import _lib, slab
from test_gpu_slab import setup_case
from test_oracle_golden import rel
worst = 0.0
for kind, nranks, nchunks in (("coupled", 4, 2), ("uncoupled", 2, 1), ("qg", 4, 1)):
    model, o, dt, phys, q0, phi0 = setup_case(kind, 256)
    one = _lib.Context(model, 256, o.kk, o.ll, o.filtr, dt, budgets=True, **phys)
    ranks = slab.make_ranks(model, 256, o.kk, o.ll, o.filtr, dt, nranks, budgets=True, **phys)
    sim = slab.SlabSimulation(ranks, "peers", nchunks=nchunks)
    for x in (one, sim):
        x.set_q(q0)
        if phi0 is not None:
            x.set_phi(phi0)
    one.step(3)
    sim.step(3)
    sim.sync()
    worst = max(worst, rel(sim.gather_qh(), one.field(_lib.F_QH)))
    if phi0 is not None:
        worst = max(worst, rel(sim.gather_phih(), one.field(_lib.F_PHIH)))
print("own blocks copied across: worst relative difference %%.2e" %% worst)
assert worst < 1e-13
print("copy path agrees with one context")
"""


def test_own_block_copy_path_still_agrees(tmp_path):
    """NIWQG_AMD_SLAB_OWN_REDIRECT=0: the rank's own block is copied across at every exchange as in rounds 2-3 (by default the A
    sub-passes read / write it on the row side and nothing moves it: ArrayListR).  The switch is read once per process, hence a
    child process; every other slab test runs the default."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "copy_path.py"
    script.write_text(COPY_PATH_WORKER % (root, os.path.join(root, "tests")))
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, NIWQG_AMD_SLAB_OWN_REDIRECT="0"))
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert "copy path agrees with one context" in out.stdout


@pytest.mark.parametrize("nranks,nchunks", [(8, 4), (2, 1), (4, 2)])
def test_config4_8192_on_virtual_ranks_against_the_oracle(nranks, nchunks):
    """BASELINE config 4 (CoupledModel 8192^2 slab-decomposed over 8 ranks; also over 2 and 4), all ranks on this one GPU.
    Truth: the reference-pinned oracle at 128^2 on the same band-limited state (the step is exact at any resolution
    that holds the band; see test_gpu_models.test_full_size_parity_through_resolution_independence)."""
    from niwqg_amd import _lib, slab
    from test_gpu_models import _band_limited_state, _low_modes
    nx, nsteps = 8192, 2
    kw = notebook_kwargs(64, False)
    kw.update(nx=128)
    o = O.NIWQGOracle("coupled", **kw)
    q0, phi0 = _band_limited_state(o.grid)
    o.set_q(q0)
    o.set_phi(phi0)
    for _ in range(nsteps):
        o._step_forward()
    kw.update(nx=nx)
    big = O.SpectralGrid(nx, L)
    q1, phi1 = _band_limited_state(big)
    phys = dict(U=kw["U"], f=kw["f"], kappa2=o.kappa2, nu=kw["nu"], nu4=kw["nu4"], mu=kw["mu"], nuw=kw["nuw"],
                nu4w=kw["nu4w"], muw=kw["muw"])
    filtr = np.ones((nx, nx))
    ranks = slab.make_ranks(_lib.COUPLED, nx, big.kk, big.ll, filtr, kw["dt"], nranks, budgets=True, **phys)
    sim = slab.SlabSimulation(ranks, "peers", nchunks=nchunks)
    sim.set_q(q1)
    sim.set_phi(phi1)
    del q1, phi1
    for r in ranks:
        r.budget_increments()
    sim.step(nsteps)
    sim.sync()
    x0, y0 = big.x.ravel()[0], big.y.ravel()[0]
    ref = _low_modes(o.phih, o.kk, o.ll, o.grid.x.ravel()[0], o.grid.y.ravel()[0], 128)
    got = _low_modes(sim.gather_phih(), big.kk, big.ll, x0, y0, nx)
    assert np.abs(got - ref).max() < 1e-11 * np.abs(ref).max()
    qh = sim.gather_qh()                                     # half spectrum (ny, nx/2+1): columns 0..12 of the band
    M = 12
    rows = np.r_[0:M + 1, nx - M:nx]
    refq = _low_modes(o.qh, o.kk, o.ll, o.grid.x.ravel()[0], o.grid.y.ravel()[0], 128)[:, :M + 1]
    gotq = qh[np.ix_(rows, np.arange(M + 1))] / nx ** 2 * np.exp(-1j * (big.kk[:M + 1][None, :] * x0
                                                                           + big.ll[rows][:, None] * y0))
    assert np.abs(gotq - refq).max() < 1e-11 * np.abs(refq).max()
    inc = ranks[nranks - 1].budget_increments()
    o0 = O.NIWQGOracle("coupled", **dict(kw, nx=128))
    o0.set_q(q0)
    o0.set_phi(phi0)
    assert np.allclose(inc, [o.Ke - o0.Ke, o.Pw - o0.Pw, o.Kw - o0.Kw], rtol=1e-8, atol=1e-30)


@pytest.mark.parametrize("model", ["coupled", "ybj"])
def test_bench_contract_with_two_ranks_rehearsed_over_gloo(model):
    """`bench.py --gpus 2` exactly as the driver launches it (torch.distributed.run, one process per rank), with both
    ranks on the one GPU of the test box and the collectives staged through gloo (NIWQG_AMD_DIST_BACKEND): the slab
    set-up, the all-ranks agreement, barrier + max-over-ranks timing and the single JSON line of rank 0."""
    from conftest import free_port
    port = free_port()
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NIWQG_AMD_DIST_BACKEND="gloo")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
                          "--gpus", "2", "--steps", "5", "--warmup", "1", "--nx", "256", "--model", model],
                         capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines                      # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["warmup"] == 1 and d["scaling"] == "strong"
    assert d["config"]["host_dispatches_per_step"] <= 1.0 and d["config"]["exchange_GB_sent_per_rank_per_step"] > 0
    # the per-step breakdown a first real multi-GPU run reads (HIP events on the two streams of rank 0)
    for key in ("compute_ms_per_step", "exposed_exchange_ms_per_step", "exchange_ms_per_step", "allreduce_ms_per_step"):
        assert key in d["config"] and d["config"][key] is not None, key
    assert d["config"]["compute_ms_per_step"] > 0
    assert d["roofline"]["peak_measured_copy"] > 1000.0           # the stream-copy denominator, GB/s
    assert "slab x2" in d["config"]["parallelism"]        # no silent fallback exists any more: a failed slab run exits non-zero
    assert d["value"] > 0 and abs(d["value"] * d["ms_per_step"] / 1e3 - 1.0) < 1e-6
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"])
    assert "cpu_baseline" not in d                     # rank 0 at N = 1 only


def test_bench_watchdog_ends_a_multi_rank_run_that_makes_no_progress():
    """a rank whose peer never shows up must not sit there until the launcher's limit: bench.py gives up with exit 124"""
    from conftest import free_port
    import os
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NIWQG_AMD_DIST_BACKEND="gloo", RANK="0", LOCAL_RANK="0", WORLD_SIZE="2",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--nx", "256", "--watchdog-seconds", "5"], capture_output=True, text=True, timeout=300, env=env,
                         cwd=root)
    assert out.returncode == 124, (out.returncode, out.stderr[-2000:])
    assert "no progress for 5 s in phase 'set-up'" in out.stderr and time.time() - t0 < 120


THREAD_RANKS_WORKER = """
import ctypes, os, sys, threading
sys.path.insert(0, %(root)r)
sys.path.insert(0, %(tests)r)
import numpy as np
from niwqg_amd import _lib, slab
from test_gpu_slab import setup_case
from test_oracle_golden import rel

def on_all(ranks, fn):
    # one host thread per rank, as one process per GPU would run them; ctypes releases the GIL inside the library
    out, err = [None] * len(ranks), []
    def run(i, r):
        try:
            out[i] = fn(r)
        except Exception as e:
            err.append((i, repr(e)))
    th = [threading.Thread(target=run, args=(i, r)) for i, r in enumerate(ranks)]
    for t in th: t.start()
    for t in th: t.join(300)
    assert not any(t.is_alive() for t in th), "a rank thread hangs"
    assert not err, err
    return out

mock = ctypes.CDLL(os.environ["NIWQG_AMD_RCCL_LIB"])
for kind, nranks, nchunks, nx in (("coupled", 2, 2, 256), ("coupled", 4, 4, 256), ("uncoupled", 4, 1, 256), ("qg", 2, 2, 256),
                                  ("ybj", 4, 2, 256), ("coupled", 8, 2, 512)):
    model, o, dt, phys, q0, phi0 = setup_case("uncoupled" if kind == "ybj" else kind, nx)
    if kind == "ybj":
        model = _lib.YBJ          # the stage results cross in a fifth group; no budgets in YBJModel's step
    nsteps = 3
    one = _lib.Context(model, nx, o.kk, o.ll, o.filtr, dt, budgets=True, **phys)
    one.set_q(q0)
    if phi0 is not None:
        one.set_phi(phi0)
    bud = kind != "ybj"
    if bud:
        one.take_budget_increments() if model != _lib.QG else one.scalar(_lib.S_KE)
    one.step(nsteps)
    if bud:
        inc1 = one.take_budget_increments() if model != _lib.QG else (one.scalar(_lib.S_KE),)
    ranks = slab.make_ranks(model, nx, o.kk, o.ll, o.filtr, dt, nranks, budgets=True, **phys)
    L = ranks[0].L
    uid = (ctypes.c_ubyte * 128)()
    assert L.nq_comm_unique_id(uid) == 0, L.nq_last_error(None)
    def chk(r, rc, what):
        assert rc == 0, (what, r.rank, L.nq_last_error(r.h))
    on_all(ranks, lambda r: chk(r, L.nq_comm_init(r.h, uid, nranks, r.rank), "nq_comm_init"))
    for r in ranks:
        chk(r, L.nq_slab_config(r.h, nchunks), "nq_slab_config")
        r.put_rows(0, q0[r.rank * r.nloc:(r.rank + 1) * r.nloc])
    on_all(ranks, lambda r: chk(r, L.nq_slab_commit(r.h, 0), "commit q"))
    if phi0 is not None:
        for r in ranks:
            r.put_rows(1, phi0[r.rank * r.nloc:(r.rank + 1) * r.nloc])
        on_all(ranks, lambda r: chk(r, L.nq_slab_commit(r.h, 1), "commit phi"))
    for r in ranks:
        if bud:
            r.budget_increments()
    before = (ctypes.c_longlong * 4)()
    mock.mock_rccl_counters(before)
    on_all(ranks, lambda r: chk(r, L.nq_slab_step(r.h, 2), "step"))
    on_all(ranks, lambda r: chk(r, L.nq_slab_step(r.h, nsteps - 2), "step"))
    after = (ctypes.c_longlong * 4)()
    mock.mock_rccl_counters(after)
    groups = {"coupled": 4, "uncoupled": 3, "qg": 2, "ybj": 2}[kind]
    sends = after[0] - before[0]
    assert sends == after[1] - before[1] == nsteps * 4 * groups * nchunks * nranks * (nranks - 1), (kind, nranks, sends)
    assert after[2] - before[2] == (0 if kind == "ybj" else nsteps), "one all-reduce of the budget sums per step"
    qh = np.concatenate([r.download(0) for r in ranks], axis=1)
    assert rel(qh, one.field(_lib.F_QH)) < 1e-13, (kind, nranks)
    if phi0 is not None:
        assert rel(np.concatenate([r.download(1) for r in ranks], axis=1), one.field(_lib.F_PHIH)) < 1e-13
        assert rel(np.concatenate([r.get_rows(_lib.F_PHI) for r in ranks], axis=0), one.field(_lib.F_PHI)) < 1e-13
    assert rel(np.concatenate([r.get_rows(_lib.F_Q) for r in ranks], axis=0), one.field(_lib.F_Q)) < 1e-13
    assert rel(np.concatenate([r.get_rows(_lib.F_U) for r in ranks], axis=0), one.field(_lib.F_U)) < 1e-12
    for r in ranks:
        if bud:
            inc = r.budget_increments()
            assert np.allclose(inc[:len(inc1)], inc1, rtol=1e-10, atol=1e-30), (r.rank, inc, inc1)
    if phi0 is not None:                                 # the tick's two all-reduces, every rank calling for itself
        d1 = one.diagnostic_sums()
        def diag(r):
            out = np.zeros(32)
            chk(r, L.nq_slab_diagnostics(r.h, _lib._dptr(out)), "diagnostics")
            return out
        for dP in on_all(ranks, diag):
            assert np.allclose(dP, d1, rtol=1e-10, atol=1e-13 * np.abs(d1).max())
    for r in ranks:
        r.close()
    one.close()
    print("ok", kind, nranks, nchunks, flush=True)
print("rank threads over the mock RCCL agree with the single context")
"""


def test_rccl_link_with_rank_threads_over_a_mock_librccl(tmp_path):
    """The RCCL link with MORE than one rank: P host threads, one per rank context, drive nq_comm_init / nq_slab_commit /
    nq_slab_step / nq_slab_diagnostics concurrently, exactly as P processes would, against tests/mock_rccl -- a test
    double of the nine librccl entry points that matches sends and receives per rank pair in posting order and refuses
    mismatched counts.  Pins what the one-rank test with the real librccl cannot: peer indexing and block offsets of the
    grouped send/recv, the chunk choreography under independently running ranks, the all-reduce points."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "tests", "mock_rccl", "libmock_rccl.so")
    src = os.path.join(root, "tests", "mock_rccl", "mock_rccl.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["hipcc", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, src, "-lpthread"], check=True)
    script = tmp_path / "thread_ranks.py"
    script.write_text(THREAD_RANKS_WORKER % dict(root=root, tests=os.path.join(root, "tests")))
    env = dict(os.environ, NIWQG_AMD_RCCL_LIB=so, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=900, env=env)
    if out.returncode != 0:
        print(out.stdout[-3000:])
        print(out.stderr[-6000:])
    assert out.returncode == 0
    assert "rank threads over the mock RCCL agree with the single context" in out.stdout


def test_bench_contract_on_one_gpu():
    """`python bench.py` (N = 1): the JSON line's contract fields, the roofline object with both denominators, and the CPU-baseline
    object (a small grid so that the oracle leg takes seconds)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "20", "--warmup", "5", "--nx", "128",
                          "--model", "coupled"], capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic", "peak_measured_copy", "frac_of_copy")) <= set(r)
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["peak_measured_copy"] > 1000.0
    cb = d["cpu_baseline"]
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(cb) and cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0


def test_bench_peers_link_and_rank_of_on_one_gpu():
    """`bench.py --gpus 2 --link peers`: ONE process drives both ranks (on a multi-GPU node rank r sits on device r and the blocks
    cross by peer copies; here both share the one device) -- no launcher, no RCCL; `bench.py --rank-of 4`: one rank of the
    decomposition alone with the null link, next to the single-GPU step of the same run; `bench.py --nx 96`: the any-size path."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}

    def line(*args):
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "10", "--warmup", "3", "--no-cpu-baseline"] + list(args),
                             capture_output=True, text=True, timeout=600, cwd=root, env=env)
        assert out.returncode == 0, (args, out.stdout[-1500:], out.stderr[-3000:])
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, lines
        return json.loads(lines[0])
    d = line("--nx", "256", "--gpus", "2", "--link", "peers")
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and "peer ranks in one process" in d["config"]["parallelism"]
    assert d["config"]["exchange_chunks_per_step"] > 0 and d["value"] > 0
    d = line("--nx", "512", "--rank-of", "4", "--chunks", "1")
    c = d["config"]
    assert d["n_gpus"] == 1 and c["rank_of"] == 4 and c["rank_compute_ms_per_step"] > 0 and c["single_gpu_ms_per_step_same_run"] > 0
    assert "NOT a simulation rate" in d["metric"] and c["exchange_GB_sent_per_rank_per_step"] == 0
    d = line("--nx", "96")
    assert "ANY-SIZE" in d["metric"] and d["value"] > 0 and d["roofline"]["bound"] == "hbm"
