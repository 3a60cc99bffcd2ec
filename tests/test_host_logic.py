"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol the header
declares, the layout helpers, the run-loop batching logic, and the loud failure without a GPU."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import niwqg_amd
    niwqg_amd.build()
    from niwqg_amd import _lib
    return _lib


def test_library_exports_every_declared_symbol(built):
    header = open(os.path.join(ROOT, "include", "niwqg_amd.h")).read()
    declared = sorted(set(re.findall(r"\b(nq_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 25
    L = built.lib()
    missing = [name for name in declared if not hasattr(L, name)]
    assert not missing, missing
    assert sorted(built.EXPORTS) == declared


def test_params_struct_matches_header(built):
    header = open(os.path.join(ROOT, "include", "niwqg_amd.h")).read()
    body = header[header.index("typedef struct nq_params {"):header.index("} nq_params;")]
    fields = re.findall(r"\b(?:int|double)\s+([^;]+);", body)
    names = [n.strip() for f in fields for n in f.split(",")]
    assert names == [n for n, _ in built.Params._fields_]


def test_no_gpu_means_loud_failure(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import niwqg_amd
    with pytest.raises(RuntimeError, match="no HIP device|hip"):
        niwqg_amd.CoupledModel.Model(nx=64)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "niwqg_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src, f


def test_hermitian_expansion_matches_numpy():
    from niwqg_amd.Kernel import hermitian_full, project_self_mirrored_columns
    rng = np.random.default_rng(0)
    a = rng.standard_normal((16, 16))
    assert np.allclose(hermitian_full(np.fft.rfft2(a)), np.fft.fft2(a), rtol=1e-13, atol=1e-13)
    h = np.fft.rfft2(a) * (1 + 0.3j)
    p = project_self_mirrored_columns(h)
    assert np.allclose(np.fft.irfft2(p), np.fft.irfft2(h), atol=1e-13)
    assert np.allclose(p[:, 0], np.fft.fft(np.fft.ifft(p[:, 0]).real), atol=1e-13)


class _FakeCtx(object):
    budgets_enabled = False

    def __init__(self):
        self.calls = []

    def step(self, n):
        self.calls.append(n)
        self.done = getattr(self, "done", 0) + n

    def request_stage4_max(self):
        self.requests = getattr(self, "requests", []) + [getattr(self, "done", 0) + 1]     # the step it will be recorded in

    def tick_snapshot(self):
        self.snapshots = getattr(self, "snapshots", []) + [getattr(self, "done", 0)]       # after how many steps

    def qh_passenger(self):
        return np.zeros(3, complex)


def _bare_kernel(tdiags, twrite, dt, tmax):
    from niwqg_amd import Kernel
    k = object.__new__(Kernel.Kernel)
    k.__dict__.update(tdiags=tdiags, twrite=twrite, dt=dt, tmax=tmax, t=0, tc=0, _cache={}, _user={},
                      _ctx=_FakeCtx(), diagnostics={}, save_to_disk=False, tsnaps=10, _pending_snapshots=[], _dual=False)
    k.ticks, k.status = [], []
    k._calc_derived_fields = lambda: k.ticks.append(k.tc)
    k._print_status_orig = Kernel.Kernel._print_status
    k._calc_ke_qg = k._calc_ke_niw = k._calc_pe_niw = lambda: 0.0
    k._calc_cfl = k._status_cfl = lambda: 0.0
    k.cflmax = 1.0

    class Log(object):
        def info(self, *a):
            k.status.append(k.tc)

        def error(self, *a):
            return ""
    k.logger = Log()
    return k


@pytest.mark.parametrize("tdiags,twrite,nsteps", [(10, 1000., 25), (1, 7, 20), (10 ** 9, 10 ** 9, 33), (4, 6., 30)])
def test_run_loop_batches_steps_but_keeps_the_reference_event_sequence(tdiags, twrite, nsteps):
    """run() may batch quiet steps into one nq_step call, but diagnostics ticks must fire after the steps
    with tc_before % tdiags == 0 (Diagnostics.py:43) and status lines when tc % twrite == 0 (Kernel.py:590)."""
    dt = 0.1
    k = _bare_kernel(tdiags, twrite, dt, (nsteps - 0.5) * dt)
    k.run()
    assert k.tc == nsteps and sum(k._ctx.calls) == nsteps
    assert k.ticks == [n for n in range(nsteps) if n % tdiags == 0]
    assert k.status == [n for n in range(1, nsteps + 1) if n % twrite == 0]
    if tdiags > nsteps and twrite > nsteps:
        assert len(k._ctx.calls) <= 3          # really batched
    # the fourth stage's max |u|, |v| are asked for exactly in the steps that end with a status line and have no tick before
    # it (ref Kernel.py:594 with :364-368; a tick recomputes u, v from the new psi: Kernel.py:681)
    want = [n for n in range(1, nsteps + 1) if n % twrite == 0 and (n - 1) % tdiags != 0]
    assert getattr(k._ctx, "requests", []) == want


def test_float_clock_is_the_references():
    """`while t < tmax: t += dt` (Kernel.py:198,:588) decides the step count, not round(tmax/dt)."""
    k = _bare_kernel(10 ** 9, 10 ** 9, 0.1, 0.3)      # 0.1+0.1+0.1 = 0.30000000000000004 > 0.3 -> 3 steps
    k.run()
    t, n = 0, 0
    while t < 0.3:
        t += 0.1
        n += 1
    assert k.tc == n


def test_enumerations_match_header(built):
    """Field, scalar and model ids of the ctypes layer are the header's."""
    header = open(os.path.join(ROOT, "include", "niwqg_amd.h")).read()
    ids = {name: int(val) for name, val in re.findall(r"\b(NQ_[A-Z0-9_]+)\s*=\s*(\d+)", header)}
    for py, c in (("F_Q", "NQ_F_Q"), ("F_QH", "NQ_F_QH"), ("F_P", "NQ_F_P"), ("F_PH", "NQ_F_PH"), ("F_PHI", "NQ_F_PHI"),
                  ("F_PHIH", "NQ_F_PHIH"), ("F_U", "NQ_F_U"), ("F_V", "NQ_F_V"), ("F_QPSI", "NQ_F_QPSI"),
                  ("F_QW", "NQ_F_QW"), ("F_QWH", "NQ_F_QWH"), ("F_PHIX", "NQ_F_PHIX"), ("F_PHIY", "NQ_F_PHIY"),
                  ("F_QH_MINUS", "NQ_F_QH_MINUS"), ("F_C", "NQ_F_C"), ("F_CH", "NQ_F_CH"),
                  ("F_QH_STAGE4", "NQ_F_QH_STAGE4"), ("S_KE", "NQ_S_KE"), ("S_PW", "NQ_S_PW"), ("S_KW", "NQ_S_KW"),
                  ("S_KE_QG", "NQ_S_KE_QG"), ("S_KE_NIW", "NQ_S_KE_NIW"), ("S_PE_NIW", "NQ_S_PE_NIW"),
                  ("S_CFL", "NQ_S_CFL"), ("COUPLED", "NQ_MODEL_COUPLED"), ("UNCOUPLED", "NQ_MODEL_UNCOUPLED"),
                  ("QG", "NQ_MODEL_QG"), ("YBJ", "NQ_MODEL_YBJ")):
        assert getattr(built, py) == ids[c], (py, c)
    from niwqg_amd import slab
    for i, name in enumerate(("PRODUCTS", "UPDATE", "WAVEPV", "INVERT", "EMIT_PHI", "INVERT_NOW", "BUDGET_SUMS",
                              "BUDGET_FINISH")):
        assert ids["NQ_PH_" + name] == i == getattr(slab, "PH_" + name)


def test_exchange_group_sizes_without_a_gpu(built):
    """nq_group_elems is pure host arithmetic: the four exchange groups of the slab decomposition carry 7 complex-plane
    equivalents per stage for CoupledModel, fewer for the other models, and split evenly."""
    import ctypes
    L = built.lib()

    def planes(model, nx, P, **kw):
        p = built.Params(model=model, nx=nx, budgets=1, dual_q=0, dt=1.0, U=0, f=1e-4, kappa2=1, nu=0, nu4=0, mu=0, nuw=0,
                         nu4w=0, muw=0, beta=0, passive_scalar=kw.get("passive", 0), nu4c=0, nuc=0, muc=0)
        n = [L.nq_group_elems(ctypes.byref(p), P, g) for g in range(4)]
        assert all(x >= 0 and x % P == 0 for x in n), n
        return [x * P / float(nx) ** 2 for x in n]

    for P in (1, 2, 4, 8):
        g = planes(built.COUPLED, 4096, P)
        assert abs(g[0] - 2) < 0.05 and abs(g[1] - 2) < 1e-12 and abs(g[2] - 1) < 0.05 and abs(g[3] - 2) < 0.07
        assert 7.0 < sum(g) < 7.2                                   # half spectra carry N/2+1 columns plus padding
    g = planes(built.UNCOUPLED, 1024, 2)
    assert g[2] == 0 and abs(g[3] - 1.5) < 0.1                     # no wave-PV group, three half-spectrum arrays back
    g = planes(built.QG, 2048, 4)
    assert g[1] == 0 and g[2] == 0 and abs(g[0] - 1) < 0.05        # uq, vq only
    assert planes(built.QG, 2048, 1, passive=1)[0] > 1.9           # + uc, vc
    p = built.Params(model=0, nx=4096, budgets=1, dual_q=0, dt=1.0, U=0, f=1e-4, kappa2=1, nu=0, nu4=0, mu=0, nuw=0, nu4w=0,
                     muw=0, beta=0, passive_scalar=0, nu4c=0, nuc=0, muc=0)
    assert L.nq_group_elems(ctypes.byref(p), 3, 0) < 0             # 4096 rows do not split over 3 ranks


class _Recorder(object):
    """stand-in for h5py.File(fno, 'w'): records what would have been written"""
    files = {}

    def __init__(self, fno):
        self.fno, self.data = fno, {}

    def create_dataset(self, name, data=None, dtype=None):
        self.data[name] = np.array(data)

    def close(self):
        open(self.fno, "w").write("stub")
        _Recorder.files[self.fno] = self.data


def test_saving_layout_names_and_overwrite_rules(tmp_path):
    """niwqg_amd/Saving.py against ref niwqg/Saving.py:6-101 with a recording writer: directory layout, the %015.0f snapshot
    names, dataset names, the tc % tsnaps rule, overwrite=False raising IOError, and loud failure without any writer."""
    from niwqg_amd import Saving

    class M(object):
        pass
    m = M()
    m.save_to_disk, m.overwrite, m.tsnaps = True, True, 5
    m.nx, m.kk, m.ll = 4, np.arange(4.), np.arange(4.)
    m.x = m.y = m.wv = np.zeros((4, 4))
    m._ctx = object()                                  # no snapshot_begin: fields are read synchronously
    m.q, m.phi = np.ones((4, 4)), 1j * np.ones((4, 4))
    m.diagnostics = {"Ke": {"value": np.arange(3.)}, "time": {"value": np.arange(3.)}}
    Saving.set_writer(None)
    if not Saving.writer_available():
        with pytest.raises(NotImplementedError):
            Saving.initialize_save_snapshots(m, str(tmp_path / "out0"))
    Saving.set_writer(_Recorder)
    try:
        path = str(tmp_path / "out")
        Saving.initialize_save_snapshots(m, path)
        assert os.path.isdir(path + "/snapshots")
        Saving.save_setup(m)
        assert set(_Recorder.files[path + "/setup.h5"]) == {"grid/nx", "grid/x", "grid/y", "grid/wv", "grid/k", "grid/l"}
        for tc, t in ((0, 0.0), (3, 300.0), (5, 500.0), (10, 123456.7)):
            m.tc, m.t = tc, t
            Saving.save_snapshots(m, fields=['t', 'q', 'phi'])
            # outside run() the file exists when the call returns, as in the reference (Saving.py:59-86)
            assert m._pending_snapshots == []
            assert os.path.exists(path + '/snapshots/{:015.0f}.h5'.format(t)) == (tc % 5 == 0)
        names = sorted(os.listdir(path + "/snapshots"))
        assert names == ["000000000000000.h5", "000000000000500.h5", "000000000123457.h5"]      # tc = 3 is no snapshot step
        snap = _Recorder.files[path + "/snapshots/000000000000500.h5"]
        assert set(snap) == {"t", "q", "phi"} and float(snap["t"]) == 500.0 and np.array_equal(snap["phi"], m.phi)
        Saving.save_diagnostics(m)
        assert set(_Recorder.files[path + "/diagnostics.h5"]) == {"Ke", "time"}
        m.overwrite = False
        with pytest.raises(IOError):
            Saving.save_setup(m)
        w = Saving.NpzWriter(str(tmp_path / "x.h5"))
        w.create_dataset("grid/nx", data=4, dtype=int)
        w.close()
        assert int(np.load(str(tmp_path / "x.h5"))["grid__nx"]) == 4
        # a model spread over several processes: every rank reads its fields (a collective gather), rank 0 alone writes
        class Ctx(object):
            def __init__(self, rank):
                self.group = type("G", (), {"rank": rank})()
        m.overwrite, m._ctx = True, Ctx(1)
        p1 = str(tmp_path / "out_rank1")
        Saving.initialize_save_snapshots(m, p1)
        Saving.save_setup(m)
        m.tc, m.t = 5, 500.0
        m._defer_snapshots = True                             # as inside run(): written by the next flush
        Saving.save_snapshots(m, fields=['t', 'q', 'phi'])
        assert len(m._pending_snapshots) == 1                 # the fields WERE read on this rank
        Saving.flush_snapshots(m)
        m._defer_snapshots = False
        Saving.save_diagnostics(m)
        assert not os.path.exists(p1) and m._pending_snapshots == []
        m._ctx = Ctx(0)
        Saving.initialize_save_snapshots(m, p1)
        Saving.save_snapshots(m, fields=['t', 'q', 'phi'])
        Saving.flush_snapshots(m)
        assert os.listdir(p1 + "/snapshots") == ["000000000000500.h5"]
    finally:
        Saving.set_writer(None)


def test_package_surface_mirrors_the_references():
    """``from niwqg import Diagnostics, InitialConditions, Saving`` (ref niwqg/__init__.py:3-5) and the model modules users import
    by name (examples/LambDipole.py:16-17, niwqg/tests/test_fft.py:4-5), each with a ``Model`` class; the auxiliary modules with
    the reference's function names."""
    import niwqg_amd
    for mod in ("Diagnostics", "InitialConditions", "Saving", "Kernel", "CoupledModel", "UnCoupledModel", "QGModel", "YBJModel"):
        assert hasattr(niwqg_amd, mod), mod
    for mod in ("CoupledModel", "UnCoupledModel", "QGModel", "YBJModel"):
        assert hasattr(getattr(niwqg_amd, mod), "Model"), mod
    assert issubclass(niwqg_amd.CoupledModel.Model, niwqg_amd.Kernel.Kernel)
    for name in ("get_diagnostic", "add_diagnostic", "describe_diagnostics", "_set_active_diagnostics", "increment_diagnostics"):
        assert callable(getattr(niwqg_amd.Diagnostics, name)), name
    for name in ("McWilliams1984", "Danioux2015", "LambDipole", "WavePacket", "PlaneWave"):
        assert callable(getattr(niwqg_amd.InitialConditions, name)), name
    for name in ("initialize_save_snapshots", "file_exist", "save_setup", "save_snapshots", "save_diagnostics"):
        assert callable(getattr(niwqg_amd.Saving, name)), name


def test_constructor_signatures_are_the_references():
    """Positional order, names and defaults of the two constructors, transcribed from ref niwqg/Kernel.py:70-98 and
    niwqg/QGModel.py:65-91; what this package adds (device, budgets, exact_qh, slab, nchunks) comes after them."""
    import inspect
    import niwqg_amd
    kernel = [('nx', 128), ('ny', None), ('L', 5e5), ('dt', 10000.), ('twrite', 1000.), ('tmax', 250000.), ('use_filter', True),
              ('cflmax', 0.8), ('U', .0), ('f', 1e-4), ('N', 0.01), ('m', 0.025), ('g', 9.81), ('nu4', 0), ('nu4w', 0), ('nu', 20),
              ('nuw', 50.), ('mu', 0), ('muw', 0), ('dealias', False), ('save_to_disk', False), ('overwrite', True),
              ('tsave_snapshots', 10), ('tdiags', 10), ('path', 'output/'), ('use_mkl', False), ('nthreads', 1)]
    qg = [('nx', 128), ('ny', None), ('L', 5e5), ('dt', 10000.), ('twrite', 1000), ('tswrite', 10), ('tmax', 250000.),
          ('use_filter', True), ('U', .0), ('nu4', 5.e9), ('nu', 0), ('mu', 0), ('beta', 0), ('passive_scalar', False), ('nu4c', 5.e9),
          ('nuc', 0), ('muc', 0), ('dealias', False), ('save_to_disk', False), ('overwrite', True), ('tsave_snapshots', 10),
          ('tdiags', 10), ('path', 'output/'), ('use_mkl', False), ('nthreads', 1)]
    for cls, want in ((niwqg_amd.Kernel.Kernel, kernel), (niwqg_amd.QGModel.Model, qg)):
        got = [(k, v.default) for k, v in inspect.signature(cls.__init__).parameters.items() if k != "self"]
        assert got[:len(want)] == want, cls
        assert set(k for k, _ in got[len(want):]) <= {"device", "budgets", "exact_qh", "slab", "nchunks"}, got[len(want):]


def test_any_size_path_host_logic(built):
    """Grids without a fused plan (niwqg_amd/_anysize.py): which sizes it takes, the loud refusals (before any device is touched),
    the class specialisation (same name and module, isinstance against the public class holds), and the op / reduction codes of the
    ctypes layer against the header."""
    import niwqg_amd
    from niwqg_amd import _anysize, _lib
    assert all(_lib.has_fused_plan(n) for n in (64, 128, 4096, 8192)) and not any(_lib.has_fused_plan(n) for n in (32, 96, 16384))
    assert all(_anysize.supported(n) for n in (4, 6, 16, 32, 96, 100, 1000, 3072, 4098, 5000, 6144, 8190, 16384))
    assert not any(_anysize.supported(n) for n in (2, 3, 97, 8194, 12288, 32768, 96.0))
    for nx in (97, 2, 8194, 32768):
        with pytest.raises(RuntimeError, match="any-size"):
            niwqg_amd.CoupledModel.Model(nx=nx)
        with pytest.raises(RuntimeError, match="any-size"):
            niwqg_amd.QGModel.Model(nx=nx)
    for pub, mix in ((niwqg_amd.CoupledModel.Model, _anysize.KernelFamily), (niwqg_amd.YBJModel.Model, _anysize.KernelFamily),
                     (niwqg_amd.QGModel.Model, _anysize.QGFamily)):
        cls = _anysize.specialise(pub, mix)
        assert cls is _anysize.specialise(pub, mix) and issubclass(cls, pub) and issubclass(cls, mix)
        assert cls.__name__ == pub.__name__ and cls.__module__ == pub.__module__ and cls.__mro__[1] is mix
    header = open(os.path.join(ROOT, "include", "niwqg_amd.h")).read()
    ids = {name: int(val) for name, val in re.findall(r"\b(NQ_[A-Z0-9_]+)\s*=\s*(\d+)", header)}
    for name in ("COPY", "MUL", "MULCONJ", "AXPBY", "AXPBYPCZ", "REAL", "ABS2", "SCALE", "CONJ", "ADDS", "IMAG", "MULADD"):
        assert ids["NQ_EW_" + name] == getattr(_anysize, "EW_" + name), name
    for name in ("SUM", "SUMABS2", "DOT", "DOTC", "MAXABS", "WSUMABS2", "MAXABSRE"):
        assert ids["NQ_RD_" + name] == getattr(_anysize, "RD_" + name), name
    assert ids["NQ_S_MAX_PHI"] == _lib.S_MAX_PHI
    src = open(os.path.join(ROOT, "niwqg_amd", "csrc", "nq_anysize.hpp")).read()
    for name, val in re.findall(r"\b(EW_[A-Z0-9]+|RD_[A-Z0-9]+) = (\d+)", src):
        assert ids["NQ_" + name] == int(val), name
