"""ctypes binding of libniwqg_amd.so (C ABI: include/niwqg_amd.h).

The library is built in-tree by ``build()`` (hipcc, gfx950) and loaded from this directory.  There is
no CPU fallback: if the library is missing or no GPU is present, constructing a model raises.
"""
import ctypes
import os
import subprocess

import numpy as np

from . import _etdrk4

try:                      # torch bundles its own libamdhip64.so.7; load it first so that our library
    import torch          # binds to the same HIP runtime instance (one runtime per process)
except Exception:         # pragma: no cover - torch is optional for single-GPU use
    torch = None

HERE = os.path.dirname(os.path.abspath(__file__))
# NIWQG_AMD_LIB: another build of the same sources (A/B experiments with compile-time knobs, tools/); default: the in-tree library
LIB_PATH = os.environ.get("NIWQG_AMD_LIB") or os.path.join(HERE, "libniwqg_amd.so")
SRC = os.path.join(HERE, "csrc", "nq_lib.hip")
HEADERS = [os.path.join(HERE, "csrc", h) for h in ("nq_fft.hpp", "nq_generic.hpp", "nq_step.hpp", "nq_anysize.hpp")] + [
    os.path.join(os.path.dirname(HERE), "include", "niwqg_amd.h")]

COUPLED, UNCOUPLED, QG, YBJ = 0, 1, 2, 3
(F_Q, F_QH, F_P, F_PH, F_PHI, F_PHIH, F_U, F_V, F_QPSI, F_QW, F_QWH, F_PHIX, F_PHIY, F_QH_MINUS, F_C, F_CH, F_QH_STAGE4,
 F_PHIH_STAGE4, F_QH_MINUS_STAGE4, F_QH_TICK, F_PHIH_TICK, F_QH_MINUS_TICK, F_QWH_TICK) = range(23)
(S_KE, S_PW, S_KW, S_KE_QG, S_KE_NIW, S_PE_NIW, S_CFL, S_MAX_PHI) = range(8)

EXPORTS = ["nq_create", "nq_destroy", "nq_last_error", "nq_set_q", "nq_set_c", "nq_set_phi", "nq_invert", "nq_refresh_grad_phi",
           "nq_step", "nq_profile_stride", "nq_request_stage4_max", "nq_get_stage4_max", "nq_tick_snapshot", "nq_sync", "nq_get_field", "nq_get_qh_passenger", "nq_get_scalar", "nq_fft2", "nq_ifft2", "nq_rfft2",
           "nq_irfft2", "nq_jacobian_psi_q", "nq_jacobian_psi_c", "nq_jacobian_psi_phi", "nq_jacobian_phic_phi", "nq_products_uq_vq", "nq_refraction", "nq_field_doubles", "nq_get_coeff", "nq_coeff_near_contour", "nq_coeff_patch", "nq_diagnostics",
           "nq_stream_copy_gbs", "nq_timer_start", "nq_timer_stop", "nq_event_record", "nq_event_elapsed", "nq_profile_enable", "nq_profile_read", "nq_profile_read_all", "nq_group_elems", "nq_create_slab",
           "nq_slab_info", "nq_group_buffers", "nq_upload_spectral", "nq_download_spectral", "nq_phase",
           "nq_reduce_buffer", "nq_reduce_read", "nq_reduce_write", "nq_device_bytes", "nq_stream",
           "nq_comm_probe", "nq_comm_unique_id", "nq_comm_init", "nq_slab_attach_peers", "nq_slab_set_callbacks", "nq_slab_set_null_link", "nq_slab_config", "nq_slab_set_stage_buffers", "nq_slab_spectral", "nq_slab_spectral_read",
           "nq_slab_step", "nq_slab_put_rows", "nq_slab_commit", "nq_slab_get_rows", "nq_slab_diagnostics",
           "nq_slab_local_max", "nq_slab_counters", "nq_slab_allreduce_ms", "nq_snapshot_begin", "nq_snapshot_end",
           "nq_any_create", "nq_any_destroy", "nq_any_last_error", "nq_any_sync", "nq_any_device_bytes", "nq_any_alloc", "nq_any_free",
           "nq_any_upload", "nq_any_download", "nq_any_fft", "nq_any_ew", "nq_any_reduce", "nq_any_expand_half", "nq_any_take_cols",
           "nq_any_set_elem", "nq_any_etdrk4", "nq_any_etdrk4_patch"]

FUSED_SIZES = (64, 128, 256, 512, 1024, 2048, 4096, 8192)       # grids the fused ETDRK4 kernels have a plan for (csrc: NQ_FOR_SIZES)


def has_fused_plan(nx):
    return nx in FUSED_SIZES


EXCHANGE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int)
ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int)


class Params(ctypes.Structure):
    _fields_ = [("model", ctypes.c_int), ("nx", ctypes.c_int), ("budgets", ctypes.c_int),
                ("dual_q", ctypes.c_int), ("dt", ctypes.c_double), ("U", ctypes.c_double),
                ("f", ctypes.c_double), ("kappa2", ctypes.c_double), ("nu", ctypes.c_double),
                ("nu4", ctypes.c_double), ("mu", ctypes.c_double), ("nuw", ctypes.c_double),
                ("nu4w", ctypes.c_double), ("muw", ctypes.c_double), ("beta", ctypes.c_double),
                ("passive_scalar", ctypes.c_int), ("nu4c", ctypes.c_double), ("nuc", ctypes.c_double),
                ("muc", ctypes.c_double)]


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(p) > t for p in [SRC] + HEADERS)


def build(force=False, verbose=False):
    """Compile the HIP library for gfx950 (cross-compiles without a GPU)."""
    if not force and not needs_build():
        return LIB_PATH
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value",
           SRC, "-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB_PATH


_lib = None


def lib():
    """Load (once) and type the shared library."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("niwqg_amd: %s is missing - run niwqg_amd._lib.build() (hipcc, gfx950); "
                           "there is no CPU fallback" % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    dp = ctypes.POINTER(ctypes.c_double)
    vp = ctypes.c_void_p
    L.nq_create.argtypes = [ctypes.POINTER(Params), dp, dp, dp, dp, ctypes.c_int, ctypes.POINTER(vp)]
    L.nq_last_error.argtypes = [vp]
    L.nq_last_error.restype = ctypes.c_char_p
    for name in ("nq_destroy", "nq_invert", "nq_refresh_grad_phi", "nq_sync", "nq_timer_start"):
        getattr(L, name).argtypes = [vp]
    for name in ("nq_set_q", "nq_set_c", "nq_set_phi", "nq_jacobian_psi_q", "nq_jacobian_psi_c", "nq_jacobian_psi_phi", "nq_jacobian_phic_phi",
                 "nq_products_uq_vq", "nq_refraction", "nq_diagnostics"):
        getattr(L, name).argtypes = [vp, dp]
    for name in ("nq_fft2", "nq_ifft2", "nq_rfft2", "nq_irfft2"):
        getattr(L, name).argtypes = [vp, dp, dp]
    L.nq_step.argtypes = [vp, ctypes.c_int]
    L.nq_request_stage4_max.argtypes = [vp]
    L.nq_profile_stride.argtypes = [vp, ctypes.c_int]
    L.nq_tick_snapshot.argtypes = [vp]
    L.nq_get_stage4_max.argtypes = [vp, dp]
    L.nq_get_field.argtypes = [vp, ctypes.c_int, dp]
    L.nq_field_doubles.argtypes = [vp, ctypes.c_int]
    L.nq_field_doubles.restype = ctypes.c_longlong
    L.nq_get_qh_passenger.argtypes = [vp, dp]
    L.nq_get_scalar.argtypes = [vp, ctypes.c_int, dp]
    L.nq_get_coeff.argtypes = [vp, ctypes.c_int, ctypes.c_int, dp]
    L.nq_coeff_near_contour.argtypes = [vp, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    L.nq_coeff_patch.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, dp]
    L.nq_stream_copy_gbs.argtypes = [vp, ctypes.c_longlong, ctypes.c_int, dp]
    L.nq_timer_stop.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
    L.nq_event_record.argtypes = [vp, ctypes.c_int]
    L.nq_event_elapsed.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
    L.nq_profile_enable.argtypes = [vp, ctypes.c_int]
    L.nq_profile_read.argtypes = [vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_float)]
    L.nq_profile_read_all.argtypes = [vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_float)]
    L.nq_group_elems.argtypes = [ctypes.POINTER(Params), ctypes.c_int, ctypes.c_int]
    L.nq_group_elems.restype = ctypes.c_longlong
    L.nq_create_slab.argtypes = [ctypes.POINTER(Params), dp, dp, dp, dp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                 ctypes.POINTER(vp), vp, ctypes.POINTER(vp)]
    L.nq_slab_info.argtypes = [vp, ctypes.POINTER(ctypes.c_int)]
    L.nq_group_buffers.argtypes = [vp, ctypes.c_int, ctypes.POINTER(vp), ctypes.POINTER(vp),
                                   ctypes.POINTER(ctypes.c_longlong)]
    L.nq_upload_spectral.argtypes = [vp, ctypes.c_int, dp]
    L.nq_download_spectral.argtypes = [vp, ctypes.c_int, dp]
    L.nq_phase.argtypes = [vp, ctypes.c_int, ctypes.c_int]
    L.nq_reduce_buffer.argtypes = [vp, ctypes.c_int, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_int)]
    L.nq_reduce_read.argtypes = [vp, ctypes.c_int, dp]
    L.nq_reduce_write.argtypes = [vp, ctypes.c_int, dp]
    L.nq_comm_probe.argtypes = []
    L.nq_comm_unique_id.argtypes = [vp]
    L.nq_comm_init.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int]
    L.nq_slab_attach_peers.argtypes = [ctypes.POINTER(vp), ctypes.c_int]
    L.nq_slab_set_callbacks.argtypes = [vp, EXCHANGE_FN, ALLREDUCE_FN, vp]
    L.nq_slab_set_null_link.argtypes = [vp]
    L.nq_slab_config.argtypes = [vp, ctypes.c_int]
    L.nq_slab_set_stage_buffers.argtypes = [vp, vp, vp]
    L.nq_slab_spectral.argtypes = [vp, ctypes.c_int]
    L.nq_slab_spectral_read.argtypes = [vp, ctypes.c_int, dp]
    L.nq_slab_step.argtypes = [vp, ctypes.c_int]
    L.nq_slab_put_rows.argtypes = [vp, ctypes.c_int, dp]
    L.nq_slab_commit.argtypes = [vp, ctypes.c_int]
    L.nq_slab_get_rows.argtypes = [vp, ctypes.c_int, dp]
    L.nq_slab_diagnostics.argtypes = [vp, dp]
    L.nq_slab_local_max.argtypes = [vp, dp]
    L.nq_slab_counters.argtypes = [vp, dp, ctypes.c_int]
    L.nq_slab_allreduce_ms.argtypes = [vp, dp]
    L.nq_snapshot_begin.argtypes = [vp, ctypes.c_int]
    L.nq_snapshot_end.argtypes = [vp, dp, dp]
    # the any-size engine (include/niwqg_amd.h: nq_any_*)
    ip = ctypes.POINTER(ctypes.c_int)
    L.nq_any_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
    L.nq_any_destroy.argtypes = [vp]
    L.nq_any_last_error.argtypes = [vp]
    L.nq_any_last_error.restype = ctypes.c_char_p
    L.nq_any_sync.argtypes = [vp]
    L.nq_any_device_bytes.argtypes = [vp]
    L.nq_any_device_bytes.restype = ctypes.c_longlong
    L.nq_any_alloc.argtypes = [vp, ctypes.c_longlong, ctypes.POINTER(vp)]
    L.nq_any_free.argtypes = [vp, vp, ctypes.c_longlong]
    L.nq_any_upload.argtypes = [vp, vp, dp, ctypes.c_longlong]
    L.nq_any_download.argtypes = [vp, vp, dp, ctypes.c_longlong]
    L.nq_any_fft.argtypes = [vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.nq_any_ew.argtypes = [vp, ctypes.c_int, vp, vp, vp, vp, ctypes.c_longlong, dp]
    L.nq_any_reduce.argtypes = [vp, ctypes.c_int, vp, vp, ctypes.c_longlong, dp]
    L.nq_any_expand_half.argtypes = [vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.nq_any_take_cols.argtypes = [vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.nq_any_set_elem.argtypes = [vp, vp, ctypes.c_longlong, ctypes.c_double, ctypes.c_double]
    L.nq_any_etdrk4.argtypes = [vp, ctypes.c_int, ctypes.POINTER(Params), dp, dp, dp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(vp),
                                ctypes.c_double, ctypes.c_int, ip, ip, ip]
    L.nq_any_etdrk4_patch.argtypes = [vp, ctypes.POINTER(vp), ctypes.c_int, ctypes.c_int, ip, ip, dp]
    L.nq_device_bytes.argtypes = [vp]
    L.nq_device_bytes.restype = ctypes.c_longlong
    L.nq_stream.argtypes = [vp]
    L.nq_stream.restype = vp
    _lib = L
    return L


def _dptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def coeff_near_contour(L, h, eq, delta):
    """(l, k) index arrays (k global) of the entries of equation eq within delta of the ETDRK4 contour (nq_coeff_near_contour)"""
    cap = 1 << 16
    while True:
        li, ki = np.empty(cap, np.int32), np.empty(cap, np.int32)
        n = L.nq_coeff_near_contour(h, eq, float(delta), cap, li.ctypes.data, ki.ctypes.data)
        if n < 0:
            raise RuntimeError("nq_coeff_near_contour failed (%d): %s" % (n, L.nq_last_error(h).decode()))
        if n <= cap:
            return li[:n].astype(np.int64), ki[:n].astype(np.int64)
        cap = n


def coeff_patch(L, h, eq, li, ki, vals):
    li, ki = np.ascontiguousarray(li, np.int32), np.ascontiguousarray(ki, np.int32)
    vals = np.ascontiguousarray(vals, np.complex128)
    if vals.shape != (len(li), 4) or len(ki) != len(li):
        raise ValueError("coeff_patch: %d entries, values of shape %s" % (len(li), vals.shape))
    rc = L.nq_coeff_patch(h, eq, len(li), li.ctypes.data, ki.ctypes.data, _dptr(vals.view(np.float64)))
    if rc != 0:
        raise RuntimeError("nq_coeff_patch failed (%d): %s" % (rc, L.nq_last_error(h).decode()))


class Context:
    """Thin object wrapper over nq_ctx; all arrays in and out are numpy."""

    def __init__(self, model, nx, kk, ll, filtr, dt, U=0.0, f=1e-4, kappa2=1.0, nu=0.0, nu4=0.0, mu=0.0,
                 nuw=0.0, nu4w=0.0, muw=0.0, beta=0.0, budgets=True, device=0, dual_q=False, passive_scalar=False,
                 nu4c=0.0, nuc=0.0, muc=0.0):
        self.L = lib()
        self.model, self.nx = model, int(nx)
        self.nk = nx if model != QG else nx // 2 + 1
        self.dual_q = bool(dual_q) and model != QG
        p = Params(model=model, nx=nx, budgets=int(bool(budgets)), dual_q=int(self.dual_q), dt=dt, U=U, f=f, kappa2=kappa2,
                   nu=nu, nu4=nu4, mu=mu, nuw=nuw, nu4w=nu4w, muw=muw, beta=beta,
                   passive_scalar=int(bool(passive_scalar) and model == QG), nu4c=nu4c, nuc=nuc, muc=muc)
        kk = np.ascontiguousarray(kk, dtype=np.float64)
        ll = np.ascontiguousarray(ll, dtype=np.float64)
        filtr = np.ascontiguousarray(filtr, dtype=np.float64)
        if kk.shape != (self.nk,) or ll.shape != (nx,) or filtr.shape != (nx, self.nk):
            raise ValueError("kk %s, ll %s, filtr %s do not fit nx = %d (nk = %d)" % (kk.shape, ll.shape, filtr.shape, nx, self.nk))
        # roots of unity of the ETDRK4 contour mean, built exactly like the reference (Kernel.py:424-426)
        r = np.exp(2j * np.pi * (np.arange(1.0, 33.0) / 32.0))
        r = np.ascontiguousarray(r).view(np.float64)
        h = ctypes.c_void_p()
        rc = self.L.nq_create(ctypes.byref(p), _dptr(kk), _dptr(ll), _dptr(filtr), _dptr(r), device, ctypes.byref(h))
        if rc != 0:
            raise RuntimeError("nq_create failed (%d): %s" % (rc, self.L.nq_last_error(None).decode()))
        self.h = h
        self.budgets_enabled = bool(budgets)
        # the entries of Qh, f0, fab, fc next to the contour, recomputed as the reference computes them (_etdrk4.py)
        prm = dict(U=U, f=f, kappa2=kappa2, nu=nu, nu4=nu4, mu=mu, nuw=nuw, nu4w=nu4w, muw=muw, beta=beta, nu4c=nu4c, nuc=nuc, muc=muc)
        eqs = [0] + ([1] if model != QG else []) + ([2] if p.passive_scalar else [])
        self.contour_patched = _etdrk4.patch_near_contour(
            lambda eq, delta: coeff_near_contour(self.L, self.h, eq, delta), lambda eq, li, ki, v: coeff_patch(self.L, self.h, eq, li, ki, v),
            model, self.nx, kk, ll, filtr, dt, prm, eqs)

    def take_budget_increments(self):
        """Ke, Pw, Kw increments accumulated on the device since the last call (Kernel.py:390-392)."""
        return tuple(self.scalar(s) for s in (S_KE, S_PW, S_KW))

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed (%d): %s" % (what, rc, self.L.nq_last_error(self.h).decode()))

    def close(self):
        if getattr(self, "h", None):
            self.L.nq_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- state
    @staticmethod
    def _shape(a, shape, what):
        """the C ABI takes plain pointers: a host array of the wrong size must never reach it"""
        if a.shape != tuple(shape):
            raise ValueError("%s: array of shape %s, expected %s" % (what, a.shape, tuple(shape)))

    def set_q(self, q):
        q = np.ascontiguousarray(q, dtype=np.float64)
        self._shape(q, (self.nx, self.nx), "set_q")
        self._chk(self.L.nq_set_q(self.h, _dptr(q)), "nq_set_q")

    def set_c(self, c):
        c = np.ascontiguousarray(c, dtype=np.float64)
        self._shape(c, (self.nx, self.nx), "set_c")
        self._chk(self.L.nq_set_c(self.h, _dptr(c)), "nq_set_c")

    def set_phi(self, phi):
        phi = np.ascontiguousarray(phi, dtype=np.complex128)
        self._shape(phi, (self.nx, self.nx), "set_phi")
        self._chk(self.L.nq_set_phi(self.h, _dptr(phi.view(np.float64))), "nq_set_phi")

    def invert(self):
        self._chk(self.L.nq_invert(self.h), "nq_invert")

    def refresh_grad_phi(self):
        self._chk(self.L.nq_refresh_grad_phi(self.h), "nq_refresh_grad_phi")

    def step(self, n=1):
        self._chk(self.L.nq_step(self.h, int(n)), "nq_step")

    def sync(self):
        self._chk(self.L.nq_sync(self.h), "nq_sync")

    def profile_stride(self, stride):
        """bracket only every stride-th launch of the enabled kernel class(es) (include/niwqg_amd.h: nq_profile_stride)"""
        self._chk(self.L.nq_profile_stride(self.h, int(stride)), "nq_profile_stride")

    def tick_snapshot(self):
        """keep qh, phih, qwh as the diagnostics tick sees them (include/niwqg_amd.h: nq_tick_snapshot)"""
        self._chk(self.L.nq_tick_snapshot(self.h), "nq_tick_snapshot")

    def request_stage4_max(self):
        """the last step of the next step() call also records max |u|, max |v| of its fourth stage (include/niwqg_amd.h)"""
        self._chk(self.L.nq_request_stage4_max(self.h), "nq_request_stage4_max")

    def status_cfl_max(self):
        """max(|u|, |v|) of the fourth stage of the last step (as requested before it) and |phi| of the new state: what the
        reference's status line takes its CFL from after a step without a tick (ref niwqg/Kernel.py:594, :660-662, :364-368)"""
        uv = np.zeros(2)
        self._chk(self.L.nq_get_stage4_max(self.h, _dptr(uv)), "nq_get_stage4_max")
        return max(uv[0], uv[1], self.scalar(S_MAX_PHI))

    # --- reads
    _REAL = (F_Q, F_P, F_U, F_V, F_QPSI, F_QW, F_C)
    _HALF = (F_QH, F_PH, F_QWH, F_QH_MINUS, F_CH, F_QH_STAGE4, F_QH_MINUS_STAGE4, F_QH_TICK, F_QH_MINUS_TICK, F_QWH_TICK)

    def field(self, fid):
        n, h = self.nx, self.nx // 2 + 1
        nd = self.L.nq_field_doubles(self.h, fid)           # the library states the size; shapes as in the header
        if nd < 0:
            raise RuntimeError("nq_get_field: unknown field id %d" % fid)
        if fid in self._REAL:
            out = np.empty((n, n), np.float64)
        elif fid in self._HALF:
            out = np.empty((n, h), np.complex128)
        else:
            out = np.empty((n, n), np.complex128)
        if out.view(np.float64).size != nd:
            raise RuntimeError("field %d: the library writes %d doubles, the binding allocated %d" % (fid, nd, out.view(np.float64).size))
        self._chk(self.L.nq_get_field(self.h, fid, _dptr(out.view(np.float64))), "nq_get_field(%d)" % fid)
        return out

    def stream_copy_gbs(self, nbytes=512 << 20, reps=5):
        """GB/s (read + written) of a plain 1r + 1w copy kernel on this device (nq_stream_copy_gbs)"""
        v = ctypes.c_double()
        self._chk(self.L.nq_stream_copy_gbs(self.h, int(nbytes), int(reps), ctypes.byref(v)), "nq_stream_copy_gbs")
        return v.value

    def qh_passenger(self):
        """anti-Hermitian part of the reference's qh on row ny/2, k = 0..nx/2 (include/niwqg_amd.h: nq_get_qh_passenger)"""
        out = np.zeros(self.nx // 2 + 1, np.complex128)
        self._chk(self.L.nq_get_qh_passenger(self.h, _dptr(out.view(np.float64))), "nq_get_qh_passenger")
        return out

    def scalar(self, sid):
        v = ctypes.c_double()
        self._chk(self.L.nq_get_scalar(self.h, sid, ctypes.byref(v)), "nq_get_scalar(%d)" % sid)
        return v.value

    def diagnostic_sums(self):
        """The 32 raw sums of one diagnostics tick (include/niwqg_amd.h: nq_diagnostics); nothing but these 256
        bytes leaves the GPU."""
        out = np.zeros(32)
        self._chk(self.L.nq_diagnostics(self.h, _dptr(out)), "nq_diagnostics")
        return out

    def coeff(self, eq, which):
        n = self.nx
        w = n if eq == 1 else n // 2 + 1
        out = np.empty((n, w), np.complex128)
        self._chk(self.L.nq_get_coeff(self.h, eq, which, _dptr(out.view(np.float64))), "nq_get_coeff")
        return out

    # --- snapshots (niwqg_amd/Saving.py)
    def snapshot_begin(self, with_phi=True):
        self._chk(self.L.nq_snapshot_begin(self.h, int(bool(with_phi))), "nq_snapshot_begin")

    def snapshot_end(self, with_phi=True):
        n = self.nx
        q = np.empty((n, n), np.float64)
        phi = np.empty((n, n), np.complex128) if with_phi else None
        self._chk(self.L.nq_snapshot_end(self.h, _dptr(q), _dptr(phi.view(np.float64)) if with_phi else None), "nq_snapshot_end")
        return q, phi

    # --- FFT seam
    def _xf(self, fn, a, in_dtype, out_shape, out_dtype, in_shape=None):
        a = np.ascontiguousarray(a, dtype=in_dtype)
        self._shape(a, in_shape or (self.nx, self.nx), fn.__name__)
        out = np.empty(out_shape, out_dtype)
        self._chk(fn(self.h, _dptr(a.view(np.float64)), _dptr(out.view(np.float64))), fn.__name__)
        return out

    def fft2(self, a):
        return self._xf(self.L.nq_fft2, a, np.complex128, (self.nx, self.nx), np.complex128)

    def ifft2(self, a):
        return self._xf(self.L.nq_ifft2, a, np.complex128, (self.nx, self.nx), np.complex128)

    def rfft2(self, a):
        return self._xf(self.L.nq_rfft2, a, np.float64, (self.nx, self.nx // 2 + 1), np.complex128)

    def irfft2(self, a):
        return self._xf(self.L.nq_irfft2, a, np.complex128, (self.nx, self.nx), np.float64, in_shape=(self.nx, self.nx // 2 + 1))

    # --- Jacobians in the reference's layouts (assembled on the device)
    def jacobian_psi_q(self):
        """Kernel family: (ny, nx) with [0,0] = 0 (Kernel.py:471-486); QGModel: (ny, nx/2+1) (QGModel.py:469-481)"""
        out = np.empty((self.nx, self.nx if self.model != QG else self.nx // 2 + 1), np.complex128)
        self._chk(self.L.nq_jacobian_psi_q(self.h, _dptr(out.view(np.float64))), "nq_jacobian_psi_q")
        return out

    def jacobian_psi_c(self):
        """QGModel's passive scalar: ik*fft(u c) + il*fft(v c), (ny, nx/2+1) (QGModel.py:483-495), through the row kernel"""
        out = np.empty((self.nx, self.nx // 2 + 1), np.complex128)
        self._chk(self.L.nq_jacobian_psi_c(self.h, _dptr(out.view(np.float64))), "nq_jacobian_psi_c")
        return out

    def products_uq_vq(self):
        """fft(u q), fft(v q) on k = 0..nx/2"""
        n, h = self.nx, self.nx // 2 + 1
        out = np.empty((2, n, h), np.complex128)
        self._chk(self.L.nq_products_uq_vq(self.h, _dptr(out.view(np.float64))), "nq_products_uq_vq")
        return out[0], out[1]

    def jacobian_psi_phi(self):
        out = np.empty((self.nx, self.nx), np.complex128)
        self._chk(self.L.nq_jacobian_psi_phi(self.h, _dptr(out.view(np.float64))), "nq_jacobian_psi_phi")
        return out

    def refraction(self):
        """fft(phi * q_psi) from the row kernel (Kernel.py:332 without the -0.5j)"""
        out = np.empty((self.nx, self.nx), np.complex128)
        self._chk(self.L.nq_refraction(self.h, _dptr(out.view(np.float64))), "nq_refraction")
        return out

    def jacobian_phic_phi(self):
        out = np.empty((self.nx, self.nx), np.complex128)
        self._chk(self.L.nq_jacobian_phic_phi(self.h, _dptr(out.view(np.float64))), "nq_jacobian_phic_phi")
        return out

    # --- timing
    def timer_start(self):
        self._chk(self.L.nq_timer_start(self.h), "nq_timer_start")

    def timer_stop(self):
        ms = ctypes.c_float()
        self._chk(self.L.nq_timer_stop(self.h, ctypes.byref(ms)), "nq_timer_stop")
        return ms.value

    def event_record(self, slot):
        self._chk(self.L.nq_event_record(self.h, int(slot)), "nq_event_record")

    def event_elapsed(self, a, b):
        ms = ctypes.c_float()
        self._chk(self.L.nq_event_elapsed(self.h, int(a), int(b), ctypes.byref(ms)), "nq_event_elapsed")
        return ms.value

    KERNEL_CLASSES = {"x_products": 0, "x_wavepv": 1, "s_q": 2, "s_phi": 3, "s_invert": 4, "y_A": 5}

    def profile_enable(self, kernel_class):
        self._chk(self.L.nq_profile_enable(self.h, int(kernel_class)), "nq_profile_enable")

    def profile_read(self):
        n, ms = ctypes.c_int(), ctypes.c_float()
        self._chk(self.L.nq_profile_read(self.h, ctypes.byref(n), ctypes.byref(ms)), "nq_profile_read")
        return n.value, ms.value

    def profile_read_all(self):
        """{class name: (launches, total ms)} after profile_enable(-2)"""
        n, ms = (ctypes.c_int * 6)(), (ctypes.c_float * 6)()
        self._chk(self.L.nq_profile_read_all(self.h, n, ms), "nq_profile_read_all")
        return {name: (n[k], ms[k]) for name, k in self.KERNEL_CLASSES.items()}

    def device_bytes(self):
        return int(self.L.nq_device_bytes(self.h))
