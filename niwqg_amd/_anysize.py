"""Grids without a fused plan: the reference's own sequence of whole-plane operations, on the device.

The reference takes any ``nx`` (ref: niwqg/Kernel.py:100-103, numpy.fft transforms any length, :562-566; QGModel.py:93-96,
:551-552).  The fused ETDRK4 kernels exist for powers of two in [64, 8192].  For every other EVEN ``nx`` in [4, 8192], and for 16384, the model
classes are specialised with the mix-ins below (``Kernel.Kernel.__new__`` / ``QGModel.Model.__new__`` pick them): state and
constants live on the device as ``Plane`` objects and every operation of the reference's time step -- whole-plane transforms of
any length (Bluestein on the power-of-two row engine), products, the ETDRK4 updates, domain means -- is one call into the
any-size engine of the library (include/niwqg_amd.h: ``nq_any_*``; csrc/nq_anysize.hpp).  Nothing is computed on the host
except what the fused path computes there too (wavenumbers, the filter, the contour-adjacent ETDRK4 entries).

The sequence of operations is the reference's, literally: 104 c2c transforms per CoupledModel step, its ``.real``
projections, its quirks (stale ``phix, phiy`` in UnCoupledModel, ``set_phi`` that does not re-invert, ``u, v`` of the fourth
stage after a step).  This path is for generality, not speed: a step is a few hundred small launches issued from Python.
"""
import ctypes

import numpy as np

from . import _etdrk4, _lib

(EW_COPY, EW_MUL, EW_MULCONJ, EW_AXPBY, EW_AXPBYPCZ, EW_REAL, EW_ABS2, EW_SCALE, EW_CONJ, EW_ADDS, EW_IMAG,
 EW_MULADD, EW_FILL) = range(13)
RD_SUM, RD_SUMABS2, RD_DOT, RD_DOTC, RD_MAXABS, RD_WSUMABS2, RD_MAXABSRE = range(7)

NX_MAX = 8192


def supported(nx):
    """even grid sizes the any-size engine takes: every even nx up to 8192 (Bluestein work rows of 2 nx - 1 points rounded up to a
    power of two <= 16384, the longest as a four-step 128 x 128 transform) and 16384 itself (four-step, no chirp)"""
    return isinstance(nx, (int, np.integer)) and nx % 2 == 0 and (4 <= nx <= NX_MAX or nx == 16384)


class Engine(object):
    """One ``nq_any`` engine of the library and a pool of its planes (freed planes are handed out again: after the first step
    nothing is allocated any more; everything runs in order on the engine's stream, so reuse is safe)."""

    def __init__(self, device=0):
        self.L = _lib.lib()
        h = ctypes.c_void_p()
        rc = self.L.nq_any_create(int(device), ctypes.byref(h))
        if rc != 0:
            raise RuntimeError("nq_any_create failed (%d): %s" % (rc, self.L.nq_any_last_error(None).decode()))
        self.h = h
        self.pool = {}
        self.sc = (ctypes.c_double * 6)()          # the scalars of the next element-wise call (one engine, one host thread)
        self._ew = self.L.nq_any_ew

    def chk(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed (%d): %s" % (what, rc, self.L.nq_any_last_error(self.h).decode()))

    def take(self, elems):
        free = self.pool.get(elems)
        if free:
            return free.pop()
        p = ctypes.c_void_p()
        self.chk(self.L.nq_any_alloc(self.h, elems, ctypes.byref(p)), "nq_any_alloc")
        return p.value

    def give(self, ptr, elems):
        self.pool.setdefault(elems, []).append(ptr)

    def sync(self):
        self.chk(self.L.nq_any_sync(self.h), "nq_any_sync")

    def device_bytes(self):
        return int(self.L.nq_any_device_bytes(self.h))

    def close(self):
        if getattr(self, "h", None):
            self.L.nq_any_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- planes from / to the host ------------------------------------------------------------------------------------
    def plane(self, array, real=None):
        a = np.asarray(array)
        if a.ndim != 2:
            raise ValueError("a plane is two-dimensional")
        isreal = (not np.iscomplexobj(a)) if real is None else bool(real)
        buf = np.ascontiguousarray(a, np.complex128)
        p = Plane(self, (int(buf.shape[0]), int(buf.shape[1])), isreal)
        self.chk(self.L.nq_any_upload(self.h, p.ptr, _lib._dptr(buf.view(np.float64)), buf.size), "nq_any_upload")
        return p

    def zeros(self, shape, real=False):
        p = Plane(self, (int(shape[0]), int(shape[1])), bool(real))
        p._ew(EW_FILL, p, s0=0.0)              # (not 0 * x: a pooled plane may hold anything, NaN included)
        return p


def _set_scalar(sc, i, v):
    if v.__class__ is float or v.__class__ is int:
        sc[i] = v
        sc[i + 1] = 0.0
    else:
        v = complex(v)
        sc[i] = v.real
        sc[i + 1] = v.imag


class Plane(object):
    """A (rows, cols) complex128 array on the device with the handful of numpy operations the reference's step uses.  ``real``
    marks planes whose imaginary part is zero by construction (numpy's float64 arrays: the results of ``.real``, ``abs()**2``):
    they download as float64 and their means are floats."""

    __array_priority__ = 1000        # numpy scalars and arrays defer to the reflected operators below

    __slots__ = ("eng", "shape", "isreal", "size", "ptr")

    def __init__(self, eng, shape, real=False):
        self.eng, self.shape, self.isreal = eng, shape, real
        self.size = size = shape[0] * shape[1]
        free = eng.pool.get(size)
        self.ptr = free.pop() if free else eng.take(size)

    def __del__(self):
        # back to the engine's pool (everything runs in order on the engine's stream: a plane handed out again is only written by
        # work queued after every reader of its previous life)
        try:
            eng = self.eng
            if eng.h:
                eng.pool.setdefault(self.size, []).append(self.ptr)
        except Exception:
            pass

    # ---- plumbing (a Coupled step is ~600 of these calls: kept lean -- the per-call Python cost is what bounds small grids)
    def _ew(self, op, a, b=None, c=None, s0=1.0, s1=0.0, s2=0.0):
        e = self.eng
        sc = e.sc
        _set_scalar(sc, 0, s0)
        _set_scalar(sc, 2, s1)
        _set_scalar(sc, 4, s2)
        rc = e._ew(e.h, op, self.ptr, a.ptr, None if b is None else b.ptr, None if c is None else c.ptr, self.size, sc)
        if rc:
            e.chk(rc, "nq_any_ew(%d)" % op)
        return self

    def _new(self, real=False):
        return Plane(self.eng, self.shape, real)

    def _reduce(self, op, other=None):
        out = np.zeros(2)
        e = self.eng
        e.chk(e.L.nq_any_reduce(e.h, op, self.ptr, None if other is None else other.ptr, self.size, _lib._dptr(out)),
              "nq_any_reduce(%d)" % op)
        return out

    def _same(self, o):
        if o.shape != self.shape:
            raise ValueError("planes of shapes %s and %s" % (self.shape, o.shape))

    def get(self):
        buf = np.empty(self.shape, np.complex128)
        e = self.eng
        e.chk(e.L.nq_any_download(e.h, self.ptr, _lib._dptr(buf.view(np.float64)), self.size), "nq_any_download")
        return np.ascontiguousarray(buf.real) if self.isreal else buf

    def copy(self):
        return self._new(self.isreal)._ew(EW_COPY, self)

    # ---- arithmetic
    @staticmethod
    def _scalar(x):
        return isinstance(x, (int, float, complex, np.integer, np.floating, np.complexfloating))

    def __mul__(self, o):
        if isinstance(o, Plane):
            self._same(o)
            return self._new(self.isreal and o.isreal)._ew(EW_MUL, self, o)
        if Plane._scalar(o):
            return self._new(self.isreal and not isinstance(o, (complex, np.complexfloating)))._ew(EW_SCALE, self, s0=o)
        return NotImplemented
    __rmul__ = __mul__

    def __truediv__(self, o):
        if Plane._scalar(o):
            return self * (1.0 / o)
        return NotImplemented

    def __neg__(self):
        return self._new(self.isreal)._ew(EW_SCALE, self, s0=-1.0)

    def _lin(self, o, sa, sb):
        if isinstance(o, Plane):
            self._same(o)
            return self._new(self.isreal and o.isreal)._ew(EW_AXPBY, self, o, s0=sa, s1=sb)
        if Plane._scalar(o):
            r = self if sa == 1.0 else self * sa
            return r._new(self.isreal and not isinstance(o, (complex, np.complexfloating)))._ew(EW_ADDS, r, s0=sb * o)
        return NotImplemented

    def __add__(self, o):
        return self._lin(o, 1.0, 1.0)
    __radd__ = __add__

    def __sub__(self, o):
        return self._lin(o, 1.0, -1.0)

    def __rsub__(self, o):
        return self._lin(o, -1.0, 1.0)

    def __pow__(self, n):
        if n == 2:
            return self * self
        if n == 3:
            return self * self * self
        return NotImplemented

    def muladd(self, b, c, s0=1.0, s1=1.0):
        """s0 * self * b + s1 * c in one pass"""
        return self._new(False)._ew(EW_MULADD, self, b, c, s0=s0, s1=s1)

    @property
    def real(self):
        return self._new(True)._ew(EW_REAL, self)

    @property
    def imag(self):
        return self._new(True)._ew(EW_IMAG, self)

    def conj(self):
        return self._new(self.isreal)._ew(EW_CONJ, self)

    def abs2(self):
        """numpy's abs(x)**2"""
        return self._new(True)._ew(EW_ABS2, self)

    def set_item(self, l, k, value):
        e = self.eng
        v = complex(value)
        e.chk(e.L.nq_any_set_elem(e.h, self.ptr, l * self.shape[1] + k, v.real, v.imag), "nq_any_set_elem")

    # ---- reductions (deterministic on the device)
    def sum(self):
        s = self._reduce(RD_SUM)
        return float(s[0]) if self.isreal else complex(s[0], s[1])

    def mean(self):
        return self.sum() / self.size

    def sumabs2(self):
        return float(self._reduce(RD_SUMABS2)[0])

    def dot(self, o):
        """sum(self * o)"""
        self._same(o)
        s = self._reduce(RD_DOT, o)
        return float(s[0]) if (self.isreal and o.isreal) else complex(s[0], s[1])

    def absmax(self):
        return float(self._reduce(RD_MAXABSRE if self.isreal else RD_MAXABS)[0])

    def std(self):
        m = self.mean()
        return float(np.sqrt((self - m).sumabs2() / self.size))


# =====================================================================================================================
def _coefficient_planes(eng, model, eq, nx, kk, ll, dt, prm, shape):
    """E, Eh, Q, f0, fab, fc of one equation as planes: device evaluation of the 32-point contour means (k_etdrk4_coeffs) and the
    entries within ``_etdrk4.DELTA`` of the contour recomputed on the host with the reference's own numpy expression, as for the
    fused path (ref: niwqg/Kernel.py:417-454; QGModel.py:426-466).  No filter folded in: this path multiplies by ``filtr``."""
    planes = [Plane(eng, shape) for _ in range(6)]
    p = _lib.Params(model=model, nx=nx, budgets=0, dual_q=0, dt=dt, U=prm.get("U", 0.0), f=prm.get("f", 1e-4),
                    kappa2=prm.get("kappa2", 1.0), nu=prm.get("nu", 0.0), nu4=prm.get("nu4", 0.0), mu=prm.get("mu", 0.0),
                    nuw=prm.get("nuw", 0.0), nu4w=prm.get("nu4w", 0.0), muw=prm.get("muw", 0.0), beta=prm.get("beta", 0.0),
                    passive_scalar=0, nu4c=prm.get("nu4c", 0.0), nuc=prm.get("nuc", 0.0), muc=prm.get("muc", 0.0))
    dev_eq = {(False, 0): 0, (False, 1): 1, (True, 0): 2, (True, 2): 3}[(model == _lib.QG, eq)]
    r = np.ascontiguousarray(np.exp(2j * np.pi * (np.arange(1.0, 33.0) / 32.0))).view(np.float64)
    kk, ll = np.ascontiguousarray(kk, np.float64), np.ascontiguousarray(ll, np.float64)
    out6 = (ctypes.c_void_p * 6)(*[pl.ptr for pl in planes])
    cap = max(4096, shape[0] * shape[1] // 8)
    cnt = ctypes.c_int(0)
    li, ki = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
    eng.chk(eng.L.nq_any_etdrk4(eng.h, dev_eq, ctypes.byref(p), _lib._dptr(kk), _lib._dptr(ll), _lib._dptr(r), shape[0], shape[1], out6,
                                _etdrk4.DELTA, cap, ctypes.byref(cnt), li.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
                                ki.ctypes.data_as(ctypes.POINTER(ctypes.c_int))), "nq_any_etdrk4")
    n = cnt.value
    if n > cap:
        raise RuntimeError("any-size ETDRK4 planes: %d entries near the contour, room for %d" % (n, cap))
    if n > 0:
        li, ki = li[:n].astype(np.int64), ki[:n].astype(np.int64)
        order = np.lexsort((ki, li))
        li, ki = li[order], ki[order]
        vals = np.ascontiguousarray(_etdrk4.contour_tables(_etdrk4.linear_operator(model, eq, kk[ki], ll[li], prm) * dt, dt), np.complex128)
        l32, k32 = np.ascontiguousarray(li, np.int32), np.ascontiguousarray(ki, np.int32)
        eng.chk(eng.L.nq_any_etdrk4_patch(eng.h, out6, shape[1], n, l32.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
                                          k32.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), _lib._dptr(vals.view(np.float64))),
                "nq_any_etdrk4_patch")
    return dict(zip(("E", "Eh", "Q", "f0", "fab", "fc"), planes))


class _Facade(object):
    """What the host classes and Saving.py still ask a context for."""

    def __init__(self, eng, budgets):
        self.eng, self.budgets_enabled = eng, bool(budgets)

    def sync(self):
        self.eng.sync()

    def device_bytes(self):
        return self.eng.device_bytes()

    def close(self):
        self.eng.close()


def _etd_stage(E, y, N, Q, F):
    """(E y + N Q) filtr"""
    return E.muladd(y, N * Q) * F


def _etd_final(c, y0, N0, Na, Nb, Nc, F):
    """(E y0 + N0 f0 + 2 (Na + Nb) fab + Nc fc) filtr   (ref: niwqg/Kernel.py:381-387)"""
    acc = c["E"].muladd(y0, N0 * c["f0"])
    acc = ((Na + Nb) * 2.0).muladd(c["fab"], acc)
    acc = Nc.muladd(c["fc"], acc)
    return acc * F


# =====================================================================================================================
class KernelFamily(object):
    """Mix-in over niwqg_amd.Kernel.Kernel (CoupledModel / UnCoupledModel / YBJModel) for grids without a fused plan: every
    method that touches the device is restated on planes, in the reference's own order of operations."""
    _tick_snapshot = None            # (the literal sequence leaves the reference's own leftovers: nothing to keep at a tick)


    _any_size = True
    _REAL = ("q", "p", "q_psi", "u", "v", "qw", "pw", "pv", "phi2")
    _CPLX = ("qh", "ph", "phi", "phih", "phix", "phiy", "qwh", "gphi2h", "lapphi")
    _uv_stage4 = False          # u, v are literally the reference's here: no reconstruction

    # ---- construction ---------------------------------------------------------------------------------------------
    def _create_context(self, phys, budgets, device, slab, nchunks):
        if slab:
            raise NotImplementedError("nx = %d has no fused plan: the any-size path runs on one GPU only" % self.nx)
        # (exact_qh / dealias need nothing extra here: this path always carries the reference's full-plane qh)
        nx = self.nx
        eng = self._eng = Engine(device)
        self._build_planes()
        d = self.__dict__
        K = self._K = {}
        for name in ("ik", "il", "wv2", "wv", "wv4", "wv2i"):
            K[name] = eng.plane(d[name])
        K["mil"], K["mwv2"], K["mwv2i"] = -K["il"], -K["wv2"], -K["wv2i"]
        K["F"] = eng.plane(self.filtr)
        prm = dict(phys)
        self._coef_q = _coefficient_planes(eng, self.model_id, 0, nx, self.kk, self.ll, self.dt, prm, (nx, nx))
        self._coef_w = _coefficient_planes(eng, self.model_id, 1, nx, self.kk, self.ll, self.dt, prm, (nx, nx))
        self._d = {}
        for name in ("qh", "ph", "phi", "phih"):                       # ref: niwqg/CoupledModel.py:33-55 (zeros)
            self._d[name] = eng.zeros((nx, nx))
        for name in ("q", "p"):
            self._d[name] = eng.zeros((nx, nx), real=True)
        self._d["phix"], self._d["phiy"] = eng.zeros((nx, nx)), eng.zeros((nx, nx))
        self._budgets = bool(budgets) and self.model_id != _lib.YBJ
        return _Facade(eng, self._budgets)

    # ---- attribute access: device planes come back as numpy arrays ---------------------------------------------------
    _COEF_NAMES = dict(expch=("q", "E"), expch_h=("q", "Eh"), Qh=("q", "Q"), f0=("q", "f0"), fab=("q", "fab"), fc=("q", "fc"),
                       expchw=("w", "E"), expch_hw=("w", "Eh"), Qhw=("w", "Q"), f0w=("w", "f0"), fabw=("w", "fab"), fcw=("w", "fc"))

    def __getattr__(self, name):
        d = self.__dict__
        if name in KernelFamily._REAL or name in KernelFamily._CPLX:
            if name in d.get("_user", {}):
                return d["_user"][name]
            pl = d.get("_d", {}).get(name)
            if pl is not None:
                return pl.get()
            if name == "q_psi" and "q" in d.get("_d", {}):
                return d["_d"]["q"].get()
            if name == "lapphi":
                return self.ifft(-self.wv2 * self.phih)
            raise AttributeError(name)
        if name in KernelFamily._COEF_NAMES and "_coef_q" in d:
            eq, which = KernelFamily._COEF_NAMES[name]
            return (self._coef_q if eq == "q" else self._coef_w)[which].get()
        if name in ("expch2", "expch2w") and "_coef_q" in d:
            v = (self._coef_q if name == "expch2" else self._coef_w)["E"].get()
            return v * v
        return super(KernelFamily, self).__getattr__(name)

    def _field(self, name):
        return KernelFamily.__getattr__(self, name)

    def _dirty(self):
        self._user.clear()
        self._cache.clear()

    # ---- transforms (ref: niwqg/Kernel.py:553-566) ------------------------------------------------------------------
    def _fft(self, a, inverse=False):
        e = self._eng
        out = Plane(e, a.shape)
        e.chk(e.L.nq_any_fft(e.h, out.ptr, a.ptr, a.shape[0], a.shape[1], 1, int(inverse)), "nq_any_fft")
        e.chk(e.L.nq_any_fft(e.h, out.ptr, out.ptr, a.shape[0], a.shape[1], 0, int(inverse)), "nq_any_fft")
        return out

    def _ifft(self, a):
        return self._fft(a, True)

    def fft(self, x):
        """numpy.fft.fft2 of a host array through the device (ref: niwqg/Kernel.py:565)"""
        return self._fft(self._eng.plane(np.asarray(x), real=False)).get()

    def ifft(self, x):
        return self._ifft(self._eng.plane(np.asarray(x), real=False)).get()

    # ---- model closures ---------------------------------------------------------------------------------------------
    def _grad_phi(self):
        d, K = self._d, self._K
        d["phix"], d["phiy"] = self._ifft(K["ik"] * d["phih"]), self._ifft(K["il"] * d["phih"])

    def _jacobian_phic_phi(self):
        """ref: niwqg/CoupledModel.py:59-73"""
        d = self._d
        self._grad_phi()
        px, py = d["phix"], d["phiy"]
        jh = self._fft(((px.conj() * py - py.conj() * px) * 1j).real)
        jh.set_item(0, 0, 0.0)
        return jh

    def jacobian_phic_phi(self):
        return self._jacobian_phic_phi().get()

    def _invert_d(self):
        d, K = self._d, self._K
        if self.model_id == _lib.COUPLED:                            # ref: niwqg/CoupledModel.py:75-97
            d["phi2"] = d["phi"].abs2()
            d["gphi2h"] = K["mwv2"] * self._fft(d["phi2"])
            d["qwh"] = ((d["gphi2h"] * 0.5 + self._jacobian_phic_phi()) * 0.5 / self.f) * K["F"]
            d["pw"] = self._ifft(K["wv2i"] * d["qwh"]).real
            d["pv"] = self._ifft(K["mwv2i"] * d["qh"]).real
            d["p"] = d["pv"] + d["pw"]
            d["ph"] = self._fft(d["p"])
            d["q"] = self._ifft(d["qh"]).real
        elif self.model_id == _lib.YBJ:                              # ref: niwqg/YBJModel.py:141-146 (p, q left alone)
            d["ph"] = K["mwv2i"] * d["qh"]
        else:                                                        # ref: niwqg/UnCoupledModel.py:54-64 (phix, phiy NOT refreshed)
            d["p"] = self._ifft(K["mwv2i"] * d["qh"]).real
            d["ph"] = self._fft(d["p"])
            d["q"] = self._ifft(d["qh"]).real

    def _invert(self):
        self._invert_d()
        self._dirty()

    def _rel_vorticity_d(self):
        d = self._d
        if self.model_id == _lib.COUPLED:                            # ref: niwqg/CoupledModel.py:145-152
            d["qw"] = self._ifft(d["qwh"]).real
            d["q_psi"] = d["q"] - d["qw"]
        else:                                                        # ref: niwqg/Kernel.py:492-501
            d["q_psi"] = d["q"]

    def _calc_rel_vorticity(self):
        self._rel_vorticity_d()

    def _uv_d(self):
        d, K = self._d, self._K
        d["u"], d["v"] = self._ifft(K["mil"] * d["ph"]).real, self._ifft(K["ik"] * d["ph"]).real

    def _jacobian_psi_q(self):
        """ref: niwqg/Kernel.py:471-486"""
        d, K = self._d, self._K
        self._uv_d()
        q = self._ifft(d["qh"]).real
        jh = K["ik"] * self._fft(d["u"] * q) + K["il"] * self._fft(d["v"] * q)
        jh.set_item(0, 0, 0.0)
        return jh

    def jacobian_psi_q(self):
        out = self._jacobian_psi_q().get()
        self._user.pop("u", None)
        self._user.pop("v", None)
        return out

    def _jacobian_psi_phi(self):
        """ref: niwqg/Kernel.py:457-469; niwqg/YBJModel.py:123-133 keeps [0,0]"""
        d = self._d
        jh = self._fft(d["u"] * d["phix"] + d["v"] * d["phiy"])
        if self.model_id != _lib.YBJ:
            jh.set_item(0, 0, 0.0)
        return jh

    def jacobian_psi_phi(self):
        return self._jacobian_psi_phi().get()

    # ---- initial state (ref: niwqg/Kernel.py:520-551) -------------------------------------------------------------------
    def set_q(self, q):
        q = np.asarray(q, np.float64)
        _lib.Context._shape(q, (self.nx, self.nx), "set_q")
        d = self._d
        d["q"] = self._eng.plane(q, real=True)
        d["qh"] = self._fft(d["q"])
        self._invert_d()
        self._rel_vorticity_d()
        self._uv_d()
        self._dirty()
        self._user["q"] = q
        self.Ke = self.ke = self._calc_ke_qg()

    def set_phi(self, phi):
        phi = np.asarray(phi, np.complex128)
        _lib.Context._shape(phi, (self.nx, self.nx), "set_phi")
        d = self._d
        d["phi"] = self._eng.plane(phi, real=False)
        d["phih"] = self._fft(d["phi"])
        self._user.pop("phi", None)
        self._user["phi"] = phi
        self.Pw = self._calc_pe_niw()
        self.Kw = self._calc_ke_niw()

    # ---- scalar integrals ---------------------------------------------------------------------------------------------
    def _spec_var(self, ah):
        """ref: niwqg/Kernel.py:654-658"""
        a = ah.copy()
        a.set_item(0, 0, 0.0)
        return a.sumabs2() / float(self.M) ** 2

    def _calc_ke_qg(self):          # ref: niwqg/Kernel.py:600-602
        return 0.5 * self._spec_var(self._K["wv"] * self._d["ph"])

    def _calc_ke_niw(self):         # ref: niwqg/Kernel.py:604-606
        return 0.5 * self._d["phi"].abs2().mean()

    def _grad2_mean(self):
        d = self._d
        return (d["phix"].abs2() + d["phiy"].abs2()).mean()

    def _calc_pe_niw(self):         # ref: niwqg/Kernel.py:608-611 (side effect: phix, phiy)
        self._grad_phi()
        return 0.25 * self._grad2_mean() / self.kappa2

    def refresh_grad_phi(self):
        self._grad_phi()

    def _calc_grad_phi(self):       # ref: niwqg/YBJModel.py:135-139
        self._grad_phi()

    def _calc_cfl(self):            # ref: niwqg/Kernel.py:660-662
        d = self._d
        return max(d["u"].absmax(), d["v"].absmax(), d["phi"].absmax()) * self.dt / self.dx

    def _status_cfl(self):
        return self._calc_cfl()

    def _lapphi(self):
        return self._ifft(self._K["mwv2"] * self._d["phih"])

    def _calc_ep_phi(self):         # ref: niwqg/Kernel.py:629-633
        d = self._d
        return (-self.nu4w * d["lapphi"].abs2().mean() - self.nuw * self._grad2_mean() - self.muw * d["phi"].abs2().mean())

    def _calc_ep_psi(self):         # ref: niwqg/Kernel.py:635-640
        d, K = self._d, self._K
        lap2psi = self._ifft(K["wv4"] * d["ph"]).real
        lapq = self._ifft(K["mwv2"] * d["qh"]).real
        if self.model_id == _lib.YBJ:        # p is allocated and never filled there (niwqg/YBJModel.py:43, :141-146)
            return self.nu4 * d["q"].dot(lap2psi) / self.M
        return (self.nu4 * d["q"].dot(lap2psi) - self.nu * d["p"].dot(lapq) + self.mu * d["p"].dot(d["q"])) / self.M

    def _calc_chi_q(self):          # ref: niwqg/Kernel.py:642-644
        return -self.nu4 * self._spec_var(self._K["wv2"] * self._d["qh"])

    def _calc_chi_phi(self):        # ref: niwqg/Kernel.py:646-652
        d, K = self._d, self._K
        lphix = self._ifft(K["ik"] * K["mwv2"] * d["phih"])
        lphiy = self._ifft(K["il"] * K["mwv2"] * d["phih"])
        return (-0.5 * self.nu4w * (lphix.abs2() + lphiy.abs2()).mean() / self.kappa2
                - 0.5 * self.nuw * d["lapphi"].abs2().mean() / self.kappa2
                - 0.5 * self.muw * self._grad2_mean() / self.kappa2)

    def _calc_ens(self):            # ref: niwqg/Kernel.py:625-627
        return 0.5 * self._d["q"].sumabs2() / self.M

    def _calc_energy_conversion(self):
        """ref: niwqg/Kernel.py:664-701"""
        d, K = self._d, self._K
        self._uv_d()
        self._user.pop("u", None)
        self._user.pop("v", None)
        self._rel_vorticity_d()
        J = d["u"] * d["phix"] + d["v"] * d["phiy"]
        d["lapphi"] = self._lapphi()
        lap2phi = self._ifft(K["wv4"] * d["phih"])
        diss = lap2phi * (-self.nu4w) + d["lapphi"] * self.nuw - d["phi"] * self.muw
        J_diss = -((diss * J.conj()).imag)
        L_diss = (diss * d["phi"].conj()).real * 0.5 * d["q_psi"]
        divFw = (d["phi"].conj() * d["lapphi"]).imag * (0.5 * self.hslash)
        self.gamma1 = (d["q_psi"] * divFw * 0.5).mean() / self.f
        self.gamma2 = 0.5 * self.hslash * (d["lapphi"].conj() * J).real.mean() / self.f
        self.xi1 = J_diss.mean() / self.f
        self.xi2 = L_diss.mean() / self.f
        self.pi = (0.5 * d["phi"].mean() * (d["q_psi"] * d["phi"].conj()).mean()).imag

    def _calc_icke_niw(self):
        self.ke_niw = self._calc_ke_niw()
        self.cke_niw = 0.5 * (abs(self._d["phi"].mean()) ** 2)
        self.ike_niw = self.ke_niw - self.cke_niw

    def _calc_conc(self):           # ref: niwqg/Kernel.py:613-619
        d = self._d
        a2 = d["phi"].abs2()
        ups = a2 - a2.mean()
        with np.errstate(invalid="ignore", divide="ignore"):
            return np.float64((ups * d["q_psi"]).mean()) / np.float64(ups.std()) / np.float64(d["q_psi"].std())

    def _calc_skewness(self):       # ref: niwqg/Kernel.py:621-623
        qp = self._d["q_psi"]
        with np.errstate(invalid="ignore", divide="ignore"):
            return np.float64((qp ** 3).mean()) / (np.float64((qp ** 2).mean()) ** 1.5)

    def _calc_ke_qg_decomp(self):   # ref: niwqg/CoupledModel.py:99-113
        d, K = self._d, self._K
        phq = K["mwv2i"] * d["qh"]
        phw = K["wv2i"] * d["qwh"]
        self.ke_qg_q = 0.5 * self._spec_var(K["wv"] * phq)
        self.ke_qg_w = 0.5 * self._spec_var(K["wv"] * phw)
        uq, vq = self._ifft(K["mil"] * phq).real, self._ifft(K["ik"] * phq).real
        uw, vw = self._ifft(K["mil"] * phw).real, self._ifft(K["ik"] * phw).real
        self.ke_qg_qw = uq.dot(uw) / self.M + vq.dot(vw) / self.M

    def _calc_kernel_derived_fields(self):
        self._calc_energy_conversion()
        self._calc_icke_niw()

    # ---- one ETDRK4 step ------------------------------------------------------------------------------------------------
    def _budget_rates(self):
        self._calc_energy_conversion()
        k = -(self.gamma1 + self.gamma2) + (self.xi1 + self.xi2) + self._calc_ep_psi()
        p = self.gamma1 + self.gamma2 + self._calc_chi_phi()
        a = self._calc_ep_phi()
        return k, p, a

    def _nonlinear_w(self):
        d = self._d
        return (-self._jacobian_psi_phi()) - self._fft(d["phi"] * d["q_psi"]) * 0.5j

    def _to_physical(self):
        d = self._d
        d["phi"] = self._ifft(d["phih"])
        self._invert_d()
        self._rel_vorticity_d()

    def _step_ybj(self):
        """ref: niwqg/YBJModel.py:52-87"""
        d, cw, F = self._d, self._coef_w, self._K["F"]
        y0 = d["phih"].copy()
        self._grad_phi()
        N0 = self._nonlinear_w()
        d["phih"] = _etd_stage(cw["Eh"], y0, N0, cw["Q"], F)
        y1 = d["phih"].copy()
        self._grad_phi()
        Na = self._nonlinear_w()
        d["phih"] = _etd_stage(cw["Eh"], y0, Na, cw["Q"], F)
        self._grad_phi()
        Nb = self._nonlinear_w()
        d["phih"] = _etd_stage(cw["Eh"], y1, Nb * 2.0 - N0, cw["Q"], F)
        self._rel_vorticity_d()
        self._grad_phi()
        Nc = self._nonlinear_w()
        d["phih"] = _etd_final(cw, y0, N0, Na, Nb, Nc, F)
        d["phi"] = self._ifft(d["phih"])

    def _step_etdrk4(self):
        """ref: niwqg/Kernel.py:307-397"""
        if self.model_id == _lib.YBJ:
            self._step_ybj()
            self._dirty()
            return
        d, cq, cw, F = self._d, self._coef_q, self._coef_w, self._K["F"]
        bud = self._budgets
        rates = []
        if bud:
            rates.append(self._budget_rates())
        q0 = d["qh"].copy()
        N0 = -self._jacobian_psi_q()
        d["qh"] = _etd_stage(cq["Eh"], q0, N0, cq["Q"], F)
        q1 = d["qh"].copy()
        w0 = d["phih"].copy()
        N0w = self._nonlinear_w()
        d["phih"] = _etd_stage(cw["Eh"], w0, N0w, cw["Q"], F)
        w1 = d["phih"].copy()
        self._to_physical()
        if bud:
            rates.append(self._budget_rates())
        Na = -self._jacobian_psi_q()
        d["qh"] = _etd_stage(cq["Eh"], q0, Na, cq["Q"], F)
        Naw = self._nonlinear_w()
        d["phih"] = _etd_stage(cw["Eh"], w0, Naw, cw["Q"], F)
        self._to_physical()
        if bud:
            rates.append(self._budget_rates())
        Nb = -self._jacobian_psi_q()
        d["qh"] = _etd_stage(cq["Eh"], q1, Nb * 2.0 - N0, cq["Q"], F)
        Nbw = self._nonlinear_w()
        d["phih"] = _etd_stage(cw["Eh"], w1, Nbw * 2.0 - N0w, cw["Q"], F)
        self._to_physical()
        if bud:
            rates.append(self._budget_rates())
        Nc = -self._jacobian_psi_q()
        d["qh"] = _etd_final(cq, q0, N0, Na, Nb, Nc, F)
        Ncw = self._nonlinear_w()
        d["phih"] = _etd_final(cw, w0, N0w, Naw, Nbw, Ncw, F)
        if bud:
            (k1, p1, a1), (k2, p2, a2), (k3, p3, a3), (k4, p4, a4) = rates
            self.Ke += self.dt * (k1 + 2 * (k2 + k3) + k4) / 6.
            self.Pw += self.dt * (p1 + 2 * (p2 + p3) + p4) / 6.
            self.Kw += self.dt * (a1 + 2 * (a2 + a3) + a4) / 6.
        self._to_physical()
        self._dirty()

    def run(self):
        """ref: niwqg/Kernel.py:183-203 (one step per iteration: nothing to batch on this path)"""
        from .Saving import save_snapshots, save_diagnostics
        if self.save_to_disk:
            save_snapshots(self, fields=['t', 'q', 'phi'])
        while self.t < self.tmax:
            self._step_forward()
        if self.save_to_disk:
            save_diagnostics(self)


# =====================================================================================================================
class QGFamily(object):
    """Mix-in over niwqg_amd.QGModel.Model for grids without a fused plan (ref: niwqg/QGModel.py): spectral planes have the
    reference's (ny, nx/2+1) shape, ``fft`` / ``ifft`` numpy.fft.rfft2 / irfft2 semantics built from the engine's c2c transforms
    (forward: both axes on the full plane, first nx/2+1 columns kept; inverse: y transform on the half plane, Hermitian extension in
    x with the imaginary parts of columns 0 and nx/2 dropped -- what numpy.fft.irfft does -- then the x transform)."""
    _tick_snapshot = None            # (the literal sequence leaves the reference's own leftovers: nothing to keep at a tick)


    _any_size = True
    _uv_stage4 = False

    def _create_context(self, phys, budgets, device, slab, nchunks):
        if slab:
            raise NotImplementedError("nx = %d has no fused plan: the any-size path runs on one GPU only" % self.nx)
        nx, nk = self.nx, self.nx // 2 + 1
        eng = self._eng = Engine(device)
        for name in ("ik", "il", "wv2", "wv", "wv4", "wv2i"):
            getattr(self, name)                                         # builds the lazy host planes
        d = self.__dict__
        K = self._K = {}
        for name in ("ik", "il", "wv2", "wv", "wv4", "wv2i"):
            K[name] = eng.plane(d[name])
        K["mil"], K["mwv2"], K["mwv2i"] = -K["il"], -K["wv2"], -K["wv2i"]
        K["F"] = eng.plane(self.filtr)
        w = np.full((nx, nk), 2.0)                                      # spec_var's weights (ref: niwqg/QGModel.py:611-619)
        w[:, 0] = w[:, -1] = 1.0
        w[0, 0] = 0.0
        K["svw"] = eng.plane(w)
        prm = dict(phys)
        self._coef_q = _coefficient_planes(eng, _lib.QG, 0, nx, self.kk, self.ll, self.dt, prm, (nx, nk))
        self._coef_c = _coefficient_planes(eng, _lib.QG, 2, nx, self.kk, self.ll, self.dt, prm, (nx, nk)) if self.passive_scalar else None
        self._d = dict(q=eng.zeros((nx, nx), real=True), p=eng.zeros((nx, nx), real=True), qh=eng.zeros((nx, nk)), ph=eng.zeros((nx, nk)))
        self._budgets = bool(budgets)
        return _Facade(eng, self._budgets)

    _COEF_NAMES = dict(expch=("q", "E"), expch_h=("q", "Eh"), Qh=("q", "Q"), f0=("q", "f0"), fab=("q", "fab"), fc=("q", "fc"),
                       expchc=("c", "E"), expch_hc=("c", "Eh"), Qhc=("c", "Q"), f0c=("c", "f0"), fabc=("c", "fab"), fcc=("c", "fc"))

    def __getattr__(self, name):
        d = self.__dict__
        if name in ("q", "p", "u", "v", "c", "qh", "ph", "ch"):
            if name in d.get("_user", {}):
                return d["_user"][name]
            pl = d.get("_d", {}).get(name)
            if pl is not None:
                return pl.get()
            raise AttributeError(name)
        if name in QGFamily._COEF_NAMES and "_coef_q" in d:
            eq, which = QGFamily._COEF_NAMES[name]
            co = self._coef_q if eq == "q" else self._coef_c
            if co is not None:
                return co[which].get()
        if name in ("expch2", "expch2c") and "_coef_q" in d:
            co = self._coef_q if name == "expch2" else self._coef_c
            if co is not None:
                v = co["E"].get()
                return v * v
        if name == "lapc" and d.get("passive_scalar") and "ch" in d.get("_d", {}):
            return self._irfft(self._K["mwv2"] * self._d["ch"]).get()
        return super(QGFamily, self).__getattr__(name)

    def _dirty(self):
        self._user.clear()
        self._cache.clear()

    # ---- transforms (ref: niwqg/QGModel.py:551-552) -------------------------------------------------------------------
    def _rfft(self, a):
        e, nx, nk = self._eng, self.nx, self.nx // 2 + 1
        full = Plane(e, (nx, nx))
        e.chk(e.L.nq_any_fft(e.h, full.ptr, a.ptr, nx, nx, 1, 0), "nq_any_fft")
        e.chk(e.L.nq_any_fft(e.h, full.ptr, full.ptr, nx, nx, 0, 0), "nq_any_fft")
        out = Plane(e, (nx, nk))
        e.chk(e.L.nq_any_take_cols(e.h, out.ptr, full.ptr, nx, nx, nk), "nq_any_take_cols")
        return out

    def _irfft(self, ah):
        e, nx, nk = self._eng, self.nx, self.nx // 2 + 1
        half = Plane(e, (nx, nk))
        e.chk(e.L.nq_any_fft(e.h, half.ptr, ah.ptr, nx, nk, 0, 1), "nq_any_fft")
        full = Plane(e, (nx, nx))
        # after the y transform the extension is row by row (rows = 1 in the mirror rule): full[y, nx-k] = conj(half[y, k]),
        # imaginary parts of columns 0 and nx/2 dropped
        e.chk(e.L.nq_any_expand_half(e.h, full.ptr, half.ptr, nx, nx, 2), "nq_any_expand_half")
        e.chk(e.L.nq_any_fft(e.h, full.ptr, full.ptr, nx, nx, 1, 1), "nq_any_fft")
        return full.real

    def fft(self, x):
        return self._rfft(self._eng.plane(np.asarray(x, np.float64), real=True)).get()

    def ifft(self, x):
        x = np.asarray(x, np.complex128)
        _lib.Context._shape(x, (self.nx, self.nx // 2 + 1), "ifft")
        return self._irfft(self._eng.plane(x)).get()

    # ---- the model (ref: niwqg/QGModel.py:469-534) ----------------------------------------------------------------------
    def _spec_var(self, ah):
        return float(ah._reduce(RD_WSUMABS2, self._K["svw"])[0]) / float(self.M) ** 2

    def _uv_d(self):
        d, K = self._d, self._K
        d["u"], d["v"] = self._irfft(K["mil"] * d["ph"]), self._irfft(K["ik"] * d["ph"])

    def _jacobian_psi_q(self):
        d, K = self._d, self._K
        self._uv_d()
        q = self._irfft(d["qh"])
        return K["ik"] * self._rfft(d["u"] * q) + K["il"] * self._rfft(d["v"] * q)

    def jacobian_psi_q(self):
        out = self._jacobian_psi_q().get()
        self._user.pop("u", None)
        self._user.pop("v", None)
        return out

    def _jacobian_psi_c(self):
        d, K = self._d, self._K
        d["c"] = self._irfft(d["ch"])
        return K["ik"] * self._rfft(d["u"] * d["c"]) + K["il"] * self._rfft(d["v"] * d["c"])

    def jacobian_psi_c(self):
        return self._jacobian_psi_c().get()

    def _invert_d(self):
        d, K = self._d, self._K
        d["ph"] = K["mwv2i"] * d["qh"]
        d["p"] = self._irfft(d["ph"])

    def _invert(self):
        self._invert_d()
        self._dirty()

    def set_q(self, q):
        q = np.asarray(q, np.float64)
        _lib.Context._shape(q, (self.nx, self.nx), "set_q")
        d = self._d
        d["q"] = self._eng.plane(q, real=True)
        d["qh"] = self._rfft(d["q"])
        self._invert_d()
        keep = {k: v for k, v in self._user.items() if k in ("c",)}
        self._dirty()
        self._user.update(keep)
        self._user["q"] = q
        self.Ke = self._calc_ke_qg()

    def set_c(self, c):
        if not self.passive_scalar:
            raise RuntimeError("set_c: the model was built with passive_scalar=False")
        c = np.asarray(c, np.float64)
        _lib.Context._shape(c, (self.nx, self.nx), "set_c")
        d = self._d
        d["c"] = self._eng.plane(c, real=True)
        d["ch"] = self._rfft(d["c"])
        self._user["c"] = c
        self.cvar = self._spec_var(d["ch"])

    def _calc_ke_qg(self):          # ref: niwqg/QGModel.py:580-582
        return 0.5 * self._spec_var(self._K["wv"] * self._d["ph"])

    def _calc_ens(self):
        return 0.5 * self._d["q"].sumabs2() / self.M

    def _calc_ep_psi(self):         # ref: niwqg/QGModel.py:588-593 (q is the start-of-step one inside a step)
        d, K = self._d, self._K
        lap2psi = self._irfft(K["wv4"] * d["ph"])
        lapq = self._irfft(K["mwv2"] * d["qh"])
        return (self.nu4 * d["q"].dot(lap2psi) - self.nu * d["p"].dot(lapq) + self.mu * d["p"].dot(d["q"])) / self.M

    def _calc_chi_q(self):          # ref: niwqg/QGModel.py:606-609
        return -self.nu4 * self._spec_var(self._K["wv2"] * self._d["qh"])

    def _calc_cfl(self):            # ref: niwqg/QGModel.py:621-629
        self._uv_d()
        self._user.pop("u", None)
        self._user.pop("v", None)
        return max(self._d["u"].absmax(), self._d["v"].absmax()) * self.dt / self.dx

    def _calc_derived_fields(self):     # ref: niwqg/QGModel.py:724-737
        if self.passive_scalar:
            d, K = self._d, self._K
            self.C2 = self._spec_var(d["ch"])
            self.gradC2 = self._spec_var(K["wv"] * d["ch"])
            d["lapc"] = self._irfft(K["mwv2"] * d["ch"])
            self.Gamma_c = 2 * d["lapc"].dot(self._irfft(self._jacobian_psi_c())) / self.M
            self.__dict__.pop("lapc", None)
        else:
            self.C2, self.gradC2, self.cvar, self.Gamma_c = 0., 0., 0., 0.
            self.lapc = np.array([0.])

    def _calc_ep_c(self):           # ref: niwqg/QGModel.py:595-598
        if not self.passive_scalar:
            return -2 * self.nu4c * 0. - 2 * self.nu * self.gradC2 - 2 * self.muc * self.C2
        return -2 * self.nu4c * self._d["lapc"].sumabs2() / self.M - 2 * self.nu * self.gradC2 - 2 * self.muc * self.C2

    def _calc_chi_c(self):          # ref: niwqg/QGModel.py:600-604
        if not self.passive_scalar:
            return 0.0
        d = self._d
        lap2c = self._irfft(self._K["wv4"] * d["ch"])
        return (2 * self.nu4c * lap2c.dot(d["lapc"]) / self.M - 2 * self.nu * d["lapc"].sumabs2() / self.M
                - 2 * self.muc * self.gradC2)

    def _step_etdrk4(self):
        """ref: niwqg/QGModel.py:328-407"""
        d, c, F = self._d, self._coef_q, self._K["F"]
        ps, cc = self.passive_scalar, self._coef_c
        q0 = d["qh"].copy()
        N0 = -self._jacobian_psi_q()
        d["qh"] = _etd_stage(c["Eh"], q0, N0, c["Q"], F)
        q1 = d["qh"].copy()
        if ps:
            c0 = d["ch"].copy()
            M0 = -self._jacobian_psi_c()
            d["ch"] = _etd_stage(cc["Eh"], c0, M0, cc["Q"], F)
            c1h = d["ch"].copy()
            self._calc_derived_fields()
            e1 = self._calc_ep_c()
        self._invert_d()
        k1 = self._calc_ep_psi()
        Na = -self._jacobian_psi_q()
        d["qh"] = _etd_stage(c["Eh"], q0, Na, c["Q"], F)
        if ps:
            Ma = -self._jacobian_psi_c()
            d["ch"] = _etd_stage(cc["Eh"], c0, Ma, cc["Q"], F)
            self._calc_derived_fields()
            e2 = self._calc_ep_c()
        self._invert_d()
        k2 = self._calc_ep_psi()
        Nb = -self._jacobian_psi_q()
        d["qh"] = _etd_stage(c["Eh"], q1, Nb * 2.0 - N0, c["Q"], F)
        if ps:
            Mb = -self._jacobian_psi_c()
            d["ch"] = _etd_stage(cc["Eh"], c1h, Mb * 2.0 - M0, cc["Q"], F)
            self._calc_derived_fields()
            e3 = self._calc_ep_c()
        self._invert_d()
        k3 = self._calc_ep_psi()
        Nc = -self._jacobian_psi_q()
        d["qh"] = _etd_final(c, q0, N0, Na, Nb, Nc, F)
        if ps:
            Mc = -self._jacobian_psi_c()
            d["ch"] = _etd_final(cc, c0, M0, Ma, Mb, Mc, F)
            self._calc_derived_fields()
            e4 = self._calc_ep_c()
            self.cvar += self.dt * (e1 + 2 * (e2 + e3) + e4) / 6.
        self._invert_d()
        d["q"] = self._irfft(d["qh"])
        if ps:
            d["c"] = self._irfft(d["ch"])
        k4 = self._calc_ep_psi()
        self.Ke += self.dt * (k1 + 2 * (k2 + k3) + k4) / 6.
        self._dirty()

    def run(self):
        """ref: niwqg/QGModel.py:184-207"""
        from .Saving import save_snapshots, save_diagnostics
        if self.save_to_disk:
            save_snapshots(self, fields=self._snapshot_fields())
        while self.t < self.tmax:
            self._step_forward()
        if self.save_to_disk:
            save_diagnostics(self)


_specialised = {}


def specialise(cls, mixin):
    """the class ``cls`` with the any-size mix-in in front of it (cached; same name and module, so that logs, pickles of the class
    name and ``isinstance`` checks against ``cls`` behave)"""
    key = (cls, mixin)
    if key not in _specialised:
        _specialised[key] = type(cls.__name__, (mixin, cls), {"__module__": cls.__module__, "__doc__": cls.__doc__})
    return _specialised[key]
