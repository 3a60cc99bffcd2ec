"""The reference's class API on a slab-decomposed simulation: ``Model(..., slab=P)`` runs ONE simulation on P peer ranks
(all on the one GPU of the test box; under torch.distributed.run the same classes take one rank per process), with the
constructor / set_q / set_phi / run() surface of niwqg.Kernel (ref: niwqg/Kernel.py:70-98, :183-203, :520-551), the status
line and CFL of :568-598 / :660-662 reduced over the ranks, the diagnostics tick of Diagnostics.py:41-58 from per-rank
partial sums, and m.q / m.phi / m.qh gathered on demand.  Expected values: the reference's own goldens and logged lines."""
import os
import re

import numpy as np
import pytest

from test_oracle_golden import notebook_kwargs, rel, L, K0, U0, TE, F0, NB, MZ
from test_gpu_models import steps

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("P", [2, 4])
def test_notebook_run_through_the_model_api_on_slabs(golden, P):
    """examples/LambDipole_CoupledModel.ipynb: the 33 logged status lines (g5) and the diagnostics series (g6) of the
    reference, through CoupledModel.Model(...).run() on P ranks."""
    import niwqg_amd
    from niwqg_amd import InitialConditions as ic
    g = golden("g6_notebook_diags.npz")
    dt = 0.025 * TE
    m = niwqg_amd.CoupledModel.Model(L=L, nx=128, tmax=10 * TE, dt=dt, m=MZ, N=NB, f=F0,
                                     twrite=int((2 * np.pi / F0) / dt), nu4=5e11, nu4w=0e10, nu=20, nuw=50e0,
                                     mu=0.e-7, muw=0e-7, use_filter=False, U=-U0, tdiags=1,
                                     save_to_disk=False, dealias=False, slab=P, nchunks=2)
    lines = []

    class Grab(object):
        def info(self, fmt, *a):
            lines.append("INFO: " + fmt % a)

        def error(self, *a):
            return "error"

    m.logger = Grab()
    m.set_q(ic.LambDipole(m, U=U0, R=2 * np.pi / K0))
    m.set_phi((np.ones((128, 128)) + 1j) * (2 * U0) / np.sqrt(2))
    m.run()
    logged = open(os.path.join(os.path.dirname(__file__), "golden", "g5_notebook_cell9_log.txt")).read()
    logged = [re.sub(r"\s+$", "", s) for s in logged.splitlines() if s.strip()]
    assert lines == logged
    for name in ("time", "Ke", "Pw", "Kw", "ke_qg", "ke_niw", "pe_niw", "gamma_r", "gamma_a", "xi_r", "xi_a",
                 "ep_psi", "chi_phi", "ep_phi", "pi", "ens", "ke_qg_q", "ke_qg_w", "ke_qg_qw", "chi_q"):
        assert np.allclose(m.diagnostics[name]['value'], g[name], rtol=1e-8, atol=1e-22), name
    assert rel(m.q, g["final_q"]) < 1e-11 and rel(m.phi, g["final_phi"]) < 1e-11


@pytest.mark.parametrize("use_filter", [False, True])
@pytest.mark.parametrize("nx,P", [(64, 2), (128, 2), (128, 4)])
def test_coupled_goldens_through_the_model_api_on_slabs(golden, nx, P, use_filter):
    import niwqg_amd
    g = golden("g2_coupled_%d_%s.npz" % (nx, "filter" if use_filter else "nofilter"))
    m = niwqg_amd.CoupledModel.Model(slab=P, **notebook_kwargs(nx, use_filter))
    m.set_q(g["q0"])
    m.set_phi(g["phi0"])
    for n in g["snaps"]:
        m.tmax = (int(n) - 0.5) * m.dt
        m.run()
        assert m.tc == n
        assert rel(m.q, g["q_%d" % n]) < 1e-12
        assert rel(m.phi, g["phi_%d" % n]) < 1e-12
        if "phih_%d" % n in g.files:
            assert rel(m.phih, g["phih_%d" % n]) < 1e-12
            assert rel(m.ph, g["ph_%d" % n]) < 1e-12
            assert rel(m.qh, g["qh_%d" % n]) < 1e-12
        assert np.allclose([m.Ke, m.Pw, m.Kw], g["budgets_%d" % n], rtol=1e-9)


def test_quirks_q1_q2_through_the_model_api_on_slabs(golden):
    """g4: UnCoupledModel's stale phix / phiy (tdiags 1 vs inf) and the set_q / set_phi order dependence, on 2 ranks."""
    import niwqg_amd
    g = golden("g4_quirks_64.npz")
    res = {}
    for tag, td in (("td1", 1), ("tdinf", 10 ** 9)):
        m = niwqg_amd.UnCoupledModel.Model(slab=2, **notebook_kwargs(64, True, tdiags=td))
        m.set_q(g["unc_q0"])
        m.set_phi(g["unc_phi0"])
        m.tmax = 19.5 * m.dt
        m.run()
        assert m.tc == 20
        assert rel(m.phi, g["unc_phi_" + tag]) < 1e-12
        assert rel(m.q, g["unc_q_" + tag]) < 1e-12
        assert np.allclose([m.Ke, m.Pw, m.Kw], g["unc_budgets_" + tag], rtol=1e-9)
        res[tag] = m.phi
    assert rel(res["td1"], res["tdinf"]) > 1e-3
    for tag in ("q_then_phi", "phi_then_q"):
        m = niwqg_amd.CoupledModel.Model(slab=2, **notebook_kwargs(64, True))
        if tag == "q_then_phi":
            m.set_q(g["order_q0"]); m.set_phi(g["order_phi0"])
        else:
            m.set_phi(g["order_phi0"]); m.set_q(g["order_q0"])
        assert rel(m.ph, g["order_ph0_" + tag]) < 1e-13
        steps(m, 1)
        assert rel(m.q, g["order_q_" + tag]) < 1e-13
        assert rel(m.phi, g["order_phi_" + tag]) < 1e-12


def test_qgmodel_golden_through_the_model_api_on_slabs(golden):
    import niwqg_amd
    g = golden("g3_qg_256.npz")
    m = niwqg_amd.QGModel.Model(L=L, nx=256, tmax=1e30, dt=float(g["dt"]), twrite=10 ** 9, nu4=7.5e8, use_filter=False,
                                U=-U0, tdiags=10 ** 9, beta=0.0, slab=4)
    m.set_q(g["q0"])
    n = int(g["snaps"][-1])
    m.tmax = (n - 0.5) * m.dt
    m.run()
    assert m.tc == n
    assert rel(m.q, g["q_%d" % n]) < 1e-11 and rel(m.qh, g["qh_%d" % n]) < 1e-11
    assert abs(m.Ke - float(g["Ke_%d" % n])) < 1e-9 * abs(float(g["Ke_%d" % n]))


def test_whole_plane_calls_of_the_class_api_on_slabs(golden):
    """Kernel.fft / ifft and the three Jacobians on a slab model: global arrays in and out, computed on the slabs.  Checked
    against numpy, against the reference's own per-function vectors (golden g1) and against the whole-plane model."""
    import niwqg_amd
    from niwqg_amd import InitialConditions as ic
    rng = np.random.default_rng(3)
    g1 = golden("g1_functions_64.npz")
    m = niwqg_amd.CoupledModel.Model(slab=2, **notebook_kwargs(64, True))
    a = rng.standard_normal((64, 64)) + 1j * rng.standard_normal((64, 64))
    assert rel(m.fft(a), np.fft.fft2(a)) < 2e-15 and rel(m.ifft(a), np.fft.ifft2(a)) < 2e-15
    r0 = m._ctx.sim.ranks[0]                      # a full-width result cannot be read as a half-spectrum slab
    buf = np.zeros((64, r0.wf), np.complex128)
    assert r0.L.nq_slab_spectral_read(r0.h, 1, niwqg_amd._lib._dptr(buf.view(np.float64))) != 0
    assert b"full-width" in r0.L.nq_last_error(r0.h)
    m.set_q(g1["q0"])
    m.set_phi(g1["phi0"])
    m._invert()
    assert rel(m.jacobian_psi_q(), g1["jac_psi_q"]) < 1e-13
    assert rel(m.jacobian_psi_phi(), g1["jac_psi_phi"]) < 1e-13
    assert rel(m.jacobian_phic_phi(), g1["jac_phic_phi"]) < 1e-13
    assert rel(m._ctx.refraction(), g1["refraction"]) < 1e-13
    assert m.jacobian_psi_q()[0, 0] == 0 and m.jacobian_psi_phi()[0, 0] == 0 and m.jacobian_phic_phi()[0, 0] == 0
    steps(m, 2)                                   # the calls left the state alone
    w = niwqg_amd.CoupledModel.Model(slab=False, **notebook_kwargs(64, True))
    w.set_q(g1["q0"])
    w.set_phi(g1["phi0"])
    w._invert()
    steps(w, 2)
    assert rel(m.q, w.q) < 1e-13 and rel(m.phi, w.phi) < 1e-13
    assert rel(m.jacobian_psi_q(), w.jacobian_psi_q()) < 1e-13 and rel(m.jacobian_phic_phi(), w.jacobian_phic_phi()) < 1e-13
    # 4 ranks, 256^2, the random initial conditions that go through the FFT seam (same seeds as the whole-plane model)
    kw = notebook_kwargs(256, True)
    s4 = niwqg_amd.UnCoupledModel.Model(slab=4, **kw)
    w4 = niwqg_amd.UnCoupledModel.Model(slab=False, **kw)
    fields = []
    for x in (s4, w4):
        np.random.seed(11)
        fields.append(ic.McWilliams1984(x, k0=6 * 2 * np.pi / L, E=0.5 * U0 ** 2))
    b = rng.standard_normal((256, 256)) + 1j * rng.standard_normal((256, 256))
    assert rel(s4.fft(b), np.fft.fft2(b)) < 2e-15 and rel(s4.ifft(b), np.fft.ifft2(b)) < 2e-15
    assert rel(fields[0], fields[1]) < 1e-12        # the generator normalises by a ratio of spectral sums
    for x in (s4, w4):
        x.set_q(fields[1])
        x.set_phi(ic.WavePacket(x, k=3 * K0, l=K0, R=L / 6, x0=L / 3, y0=L / 2))
    assert rel(s4.jacobian_psi_phi(), w4.jacobian_psi_phi()) < 1e-13
    assert rel(s4.jacobian_psi_q(), w4.jacobian_psi_q()) < 1e-13
    with pytest.raises(NotImplementedError):
        s4._ctx.coeff(0, 0)
    # QGModel: rfft2 / irfft2 semantics, half-plane Jacobian with [0,0] kept, the passive scalar's tick Jacobian
    kwq = dict(L=L, nx=128, tmax=1e30, dt=2000.0, twrite=10 ** 9, nu4=7.5e8, use_filter=True, U=-U0, tdiags=10 ** 9,
               passive_scalar=True)
    sq = niwqg_amd.QGModel.Model(slab=2, **kwq)
    wq = niwqg_amd.QGModel.Model(slab=False, **kwq)
    r = rng.standard_normal((128, 128))
    h = np.fft.rfft2(r) + 0.3j * rng.standard_normal((128, 65))        # not Hermitian on the self-mirrored columns
    assert rel(sq.fft(r), np.fft.rfft2(r)) < 2e-15 and rel(sq.ifft(h), np.fft.irfft2(h)) < 2e-15
    q0 = 1e-5 * rng.standard_normal((128, 128))
    for x in (sq, wq):
        x.set_q(q0)
        x.set_c(r)
    assert rel(sq.jacobian_psi_q(), wq.jacobian_psi_q()) < 1e-13
    for x in (sq, wq):
        x.tmax = 2.5 * x.dt
        x.run()
    assert rel(sq.jacobian_psi_c(), wq.jacobian_psi_c()) < 1e-12


@pytest.mark.parametrize("P", [2, 4])
def test_dual_copy_q_equation_on_slabs(golden, P):
    """dealias=True (the reference's 2/3 mask is not mirror-symmetric, g4) and exact_qh=True (g2) keep two copies of
    the q equation per column slab; the full-plane qh assembled from both must equal the reference's."""
    import niwqg_amd
    g = golden("g4_quirks_64.npz")
    nx = 64 * (P // 2)
    if nx == 64:
        kw = notebook_kwargs(64, False)
        kw.update(dealias=True, nu4w=1e10, mu=1e-8, muw=2e-8)
        m = niwqg_amd.CoupledModel.Model(slab=P, **kw)
        m.set_q(g["rough_q0"])
        m.set_phi(g["rough_phi0"])
        steps(m, 5)
        assert rel(m.q, g["rough_q"]) < 1e-11 and rel(m.phi, g["rough_phi"]) < 1e-11
        assert rel(m.phih, g["rough_phih"]) < 1e-11
        assert rel(m.qh, g["rough_qh"]) < 1e-11
        assert np.allclose([m.Ke, m.Pw, m.Kw], g["rough_budgets"], rtol=1e-8)
    g = golden("g2_coupled_%d_nofilter.npz" % nx)
    m = niwqg_amd.CoupledModel.Model(slab=P, exact_qh=True, **notebook_kwargs(nx, False))
    m.set_q(g["q0"])
    m.set_phi(g["phi0"])
    n = 10 if nx == 64 else 100
    steps(m, n)
    assert rel(m.qh, g["qh_%d" % n]) < 1e-12             # no row excluded
    assert rel(m.q, g["q_%d" % n]) < 1e-12 and rel(m.phi, g["phi_%d" % n]) < 1e-12
    assert np.allclose([m.Ke, m.Pw, m.Kw], g["budgets_%d" % n], rtol=1e-9)
    # and the whole-plane dual model is the same computation
    kw = notebook_kwargs(nx, False)
    kw.update(dealias=True)
    w = niwqg_amd.UnCoupledModel.Model(slab=False, **kw)
    s = niwqg_amd.UnCoupledModel.Model(slab=P, **kw)
    for x in (w, s):
        x.set_q(g["q0"])
        x.set_phi(g["phi0"])
        steps(x, 7)
    assert rel(s.qh, w.qh) < 1e-13 and rel(s.phi, w.phi) < 1e-13 and rel(s.q, w.q) < 1e-13
    assert np.allclose([s.Ke, s.Pw, s.Kw], [w.Ke, w.Pw, w.Kw], rtol=1e-11)


MODEL_WORKER = """
import sys
sys.path.insert(0, %r)
sys.path.insert(0, %r)
import numpy as np
import niwqg_amd
from test_oracle_golden import notebook_kwargs, rel

g = np.load(%r, allow_pickle=False)
m = niwqg_amd.CoupledModel.Model(**notebook_kwargs(128, True))        # WORLD_SIZE = 2: slab-decomposed by itself
assert type(m._ctx).__name__ == "SlabContext" and m._ctx.sim.nranks == 2
m.set_q(g["q0"])
m.set_phi(g["phi0"])
m.tmax = 99.5 * m.dt
m.run()
assert m.tc == 100
eq, ep = rel(m.q, g["q_100"]), rel(m.phi, g["phi_100"])
assert eq < 1e-12 and ep < 1e-12, (eq, ep)
assert np.allclose([m.Ke, m.Pw, m.Kw], g["budgets_100"], rtol=1e-9)
assert abs(m._calc_cfl() - float(np.max([np.abs(m.u).max(), np.abs(m.v).max(), np.abs(m.phi).max()])) * m.dt / m.dx) < 1e-12
# YBJModel: its fifth exchange group crosses through the same callbacks (golden g8)
g8 = np.load(%r, allow_pickle=False)
kw = notebook_kwargs(64, True, tdiags=10 ** 9)
kw.update(nu4w=3e9, muw=1e-7)
y = niwqg_amd.YBJModel.Model(**kw)
y.set_q(g8["q0"])
y.set_phi(g8["phi0"])
y.tmax = 19.5 * y.dt
y.run()
assert rel(y.phi, g8["phi_tdinf_filter"]) < 1e-12 and rel(y.phix, g8["phix_tdinf_filter"]) < 1e-12
# dual-copy q equation (exact_qh) on two processes
g2 = np.load(%r, allow_pickle=False)
d = niwqg_amd.CoupledModel.Model(exact_qh=True, **notebook_kwargs(64, False))
d.set_q(g2["q0"])
d.set_phi(g2["phi0"])
while d.tc < 10:
    d._step_forward()
assert rel(d.qh, g2["qh_10"]) < 1e-12 and rel(d.phi, g2["phi_10"]) < 1e-12
# save_to_disk on a model spread over two processes: both gather, rank 0 alone writes (NpzWriter in place of h5py)
import os
from niwqg_amd import Saving
Saving.set_writer(Saving.NpzWriter)
out = %r
kw = notebook_kwargs(64, True, tdiags=4)
kw.update(tmax=11.5 * kw["dt"], save_to_disk=True, tsave_snapshots=6, path=out)
w = niwqg_amd.CoupledModel.Model(**kw)
w.set_q(g2["q0"])
w.set_phi(g2["phi0"])
w.run()
w._ctx.group.barrier()
names = sorted(os.listdir(out + "/snapshots"))
assert names == ['{:015.0f}.h5'.format(n * w.dt) for n in (0, 6, 12)], names
last = np.load(out + "/snapshots/" + names[-1], allow_pickle=False)
assert rel(last["q"], w.q) < 1e-15 and rel(last["phi"], w.phi) < 1e-15 and float(last["t"]) == w.t
assert os.path.exists(out + "/setup.h5") and os.path.exists(out + "/diagnostics.h5")
if m._ctx.group.rank == 0:
    print("two-process model agrees with the reference golden")
m._ctx.group.close()
"""


def test_model_api_in_two_processes_over_gloo(tmp_path):
    """`torch.distributed.run --nproc-per-node 2 script.py` where the script just builds CoupledModel.Model(...) and calls
    run(): every rank executes the reference's API, the model is slab-decomposed over the two processes (both on the one
    GPU of the test box, callbacks + gloo for the wire), and the 100-step golden of the reference is reproduced."""
    import subprocess
    import sys
    from conftest import free_port, GOLDEN
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "model_worker.py"
    script.write_text(MODEL_WORKER % (root, os.path.join(root, "tests"), os.path.join(GOLDEN, "g2_coupled_128_filter.npz"),
                                      os.path.join(GOLDEN, "g8_ybj_64.npz"), os.path.join(GOLDEN, "g2_coupled_64_nofilter.npz"),
                                      str(tmp_path / "saved")))
    env = dict(os.environ, NIWQG_AMD_DIST_BACKEND="gloo")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(script)],
                         capture_output=True, text=True, timeout=600, env=env)
    if out.returncode != 0:
        print(out.stdout[-3000:])
        print(out.stderr[-6000:])
    assert out.returncode == 0
    assert "two-process model agrees with the reference golden" in out.stdout


@pytest.mark.parametrize("td_tag,td", [("td1", 1), ("tdinf", 10 ** 9)])
@pytest.mark.parametrize("use_filter", [True, False])
def test_ybj_model_on_slabs_against_the_reference(golden, td_tag, td, use_filter):
    """golden g8 (niwqg.YBJModel run by the reference) through Model(slab=2): the stage results 0..2 cross y -> x in an
    exchange group of their own, so the stale phix / phiy a step leaves behind are the reference's as well."""
    import niwqg_amd
    g = golden("g8_ybj_64.npz")
    key = "%s_%s" % (td_tag, "filter" if use_filter else "nofilter")
    kw = notebook_kwargs(64, use_filter, tdiags=td)
    kw.update(nu4w=3e9, muw=1e-7)
    m = niwqg_amd.YBJModel.Model(slab=2, **kw)
    m.set_q(g["q0"])
    m.set_phi(g["phi0"])
    m.tmax = 19.5 * m.dt
    m.run()
    assert m.tc == 20
    assert rel(m.phi, g["phi_" + key]) < 1e-12 and rel(m.phih, g["phih_" + key]) < 1e-12
    assert rel(m.phix, g["phix_" + key]) < 1e-12 and rel(m.phiy, g["phiy_" + key]) < 1e-12
    assert np.allclose([m.Ke, m.Pw, m.Kw, m._calc_ke_niw(), m._calc_ke_qg()], g["scalars_" + key], rtol=1e-11)
    if td == 1:
        for name in m.diagnostics:
            ref = g["diag_%s_%s" % (name, key)]
            got = np.asarray(m.diagnostics[name]['value'])
            assert np.allclose(got, ref, rtol=1e-8, atol=1e-12 if name in ("skew", "conc_niw") else 1e-30), name


def test_ybj_model_on_four_slabs_equals_the_whole_plane_model():
    import niwqg_amd
    from niwqg_amd import InitialConditions as ic
    kw = notebook_kwargs(256, True)
    kw.update(nu4w=3e9, muw=1e-7)
    w = niwqg_amd.YBJModel.Model(slab=False, **kw)
    q0 = ic.LambDipole(w, U=U0, R=2 * np.pi / K0)
    phi0 = ic.WavePacket(w, k=3 * K0, l=K0, R=L / 6, x0=L / 3, y0=L / 2)
    for nch in (1, 2):
        s = niwqg_amd.YBJModel.Model(slab=4, nchunks=nch, **kw)
        for x in ((w, s) if nch == 1 else (s,)):
            x.set_q(q0)
            x.set_phi(phi0)
            steps(x, 12)
        assert rel(s.phi, w.phi) < 1e-13 and rel(s.phix, w.phix) < 1e-13 and rel(s.phiy, w.phiy) < 1e-13
        assert rel(s.q, w.q) < 1e-13 and rel(s.u, w.u) < 1e-13


@pytest.mark.parametrize("use_filter", [True, False])
def test_qg_passive_scalar_on_slabs_against_the_reference(golden, use_filter):
    """golden g10 (QGModel with passive_scalar=True, run by the reference) through Model(slab=2): the scalar rides in the
    spare slots of exchange groups 0 and 3; trajectory, variance budget and every diagnostics series of the tick."""
    import niwqg_amd
    g = golden("g10_qg_passive_64.npz")
    key = "filter" if use_filter else "nofilter"
    m = niwqg_amd.QGModel.Model(L=L, nx=64, tmax=19.5 * float(g["dt"]), dt=float(g["dt"]), twrite=10 ** 9,
                                nu4=7.5e8 * 16, nu=5.0, mu=1e-8, use_filter=use_filter, U=-U0, tdiags=1, beta=2e-11,
                                passive_scalar=True, nu4c=3e9, nuc=2.0, muc=1e-8, slab=2)
    m.set_q(g["q0"])
    m.set_c(g["c0"])
    m.run()
    assert m.tc == 20
    assert rel(m.q, g["q_" + key]) < 1e-11 and rel(m.qh, g["qh_" + key]) < 1e-11
    assert rel(m.c, g["c_" + key]) < 1e-12 and rel(m.ch, g["ch_" + key]) < 1e-12
    m._calc_derived_fields()
    assert np.allclose([m.Ke, m.cvar, m.C2, m.gradC2], g["scalars_" + key], rtol=1e-10)
    for name in m.diagnostics:
        ref = g["diag_%s_%s" % (name, key)]
        tol = 1e-7 if name == "Gamma_c" else 1e-8
        assert np.allclose(np.asarray(m.diagnostics[name]['value']), ref, rtol=tol, atol=1e-30), name


def test_qg_passive_scalar_on_four_slabs_equals_the_whole_plane_model():
    import niwqg_amd
    rng = np.random.default_rng(5)
    kw = dict(L=L, nx=256, tmax=1e30, dt=2000.0, twrite=10 ** 9, nu4=7.5e8, use_filter=True, U=-U0, tdiags=3, beta=1e-11,
              passive_scalar=True, nu4c=3e9, nuc=2.0, muc=1e-8)
    w = niwqg_amd.QGModel.Model(slab=False, **kw)
    s = niwqg_amd.QGModel.Model(slab=4, **kw)
    x = np.linspace(0, 2 * np.pi, 256, endpoint=False)
    q0 = 1e-5 * (np.sin(3 * x)[None, :] * np.cos(2 * x)[:, None] + 0.1 * rng.standard_normal((256, 256)))
    c0 = np.cos(x)[None, :] * np.sin(4 * x)[:, None] + 0.05 * rng.standard_normal((256, 256))
    for m in (w, s):
        m.set_q(q0)
        m.set_c(c0)
        m.tmax = 11.5 * m.dt
        m.run()
    assert rel(s.q, w.q) < 1e-13 and rel(s.c, w.c) < 1e-13 and rel(s.ch, w.ch) < 1e-13
    assert np.allclose([s.Ke, s.cvar], [w.Ke, w.cvar], rtol=1e-12)
    for name in w.diagnostics:
        a, b = np.asarray(s.diagnostics[name]['value']), np.asarray(w.diagnostics[name]['value'])
        assert np.allclose(a, b, rtol=1e-7 if name == "Gamma_c" else 1e-10, atol=1e-30), name


@pytest.mark.parametrize("P", [2, 4])
def test_contour_adjacent_entries_are_patched_on_every_slab_rank(golden, P):
    """Golden g13 (the REAL reference, CoupledModel 256^2 with the 2/3 mask, U = 0: c dt within 2e-6 of the ETDRK4 contour) on P
    slab ranks: each rank lists and patches its own columns (nq_coeff_near_contour with global column indices), all of the
    list is covered once, and six steps from white noise match the reference as the single context does."""
    import niwqg_amd
    from test_oracle_golden import G13_COUPLED, g13_half_plane_q_values
    g = golden("g13_contour_entries.npz")
    m = niwqg_amd.CoupledModel.Model(slab=P, **G13_COUPLED)
    li, _, _ = g13_half_plane_q_values(g, np.asarray(m.filtr), 256)
    counts = [r.contour_patched for r in m._ctx.sim.ranks]
    assert sum(c[0] for c in counts) == len(li) and sum(c[1] for c in counts) == len(g["cw_l"])
    rng = np.random.default_rng(14)
    m.set_q(1e-5 * rng.standard_normal((256, 256)))
    m.set_phi(0.05 * (rng.standard_normal((256, 256)) + 1j * rng.standard_normal((256, 256))))
    for _ in range(6):
        m._step_forward()
    assert rel(m.q, g["c_q6"]) < 1e-11 and rel(m.phi, g["c_phi6"]) < 1e-11
