"""1-D slab decomposition of ONE simulation over several GPUs (one process per GPU, RCCL over xGMI).

Physical / mixed-space rows are split over ranks on the "x side" (row kernels), spectral / mixed-space
columns on the "y side" (spectral kernels); the arrays that cross together form four exchange groups, each
moved by ONE ``all_to_all_single`` per transition: 4 per ETDRK4 stage for CoupledModel, 16 per step (SURVEY
section 8e).  The device library lays its buffers out so that the x-side buffer *is* the send/receive
buffer (blocked rows) and the y-side buffer *is* the column slab: no pack or unpack passes.

Two transports share all of the logic:
  * ``TorchTransport``  -- real ranks, ``torch.distributed`` (backend "nccl" = RCCL on ROCm);
  * ``VirtualTransport`` -- P ranks inside one process on one GPU, blocks moved with tensor copies; this is how
    the decomposition is tested on a single-GPU box (tests/test_gpu_slab.py).
"""
import ctypes

import numpy as np

from . import _lib

(PH_PRODUCTS, PH_UPDATE, PH_WAVEPV, PH_INVERT, PH_EMIT_PHI, PH_INVERT_NOW, PH_BUDGET_SUMS,
 PH_BUDGET_FINISH) = range(8)


class SlabRank(object):
    """One rank's device context plus its torch-owned exchange buffers."""

    def __init__(self, model, nx, kk, ll, filtr, dt, nranks, rank, device, stream=None, budgets=True, **phys):
        import torch
        self.torch = torch
        self.L = _lib.lib()
        self.model, self.nx, self.nranks, self.rank = model, int(nx), int(nranks), int(rank)
        self.dev = torch.device("cuda", device)
        p = _lib.Params(model=model, nx=nx, budgets=int(bool(budgets)), dual_q=0, dt=dt,
                        U=phys.get("U", 0.0), f=phys.get("f", 1e-4), kappa2=phys.get("kappa2", 1.0),
                        nu=phys.get("nu", 0.0), nu4=phys.get("nu4", 0.0), mu=phys.get("mu", 0.0),
                        nuw=phys.get("nuw", 0.0), nu4w=phys.get("nu4w", 0.0), muw=phys.get("muw", 0.0),
                        beta=phys.get("beta", 0.0))
        self.budgets = bool(budgets)
        self.gx, self.gy = [], []            # x-side / y-side tensors of the four groups (None when empty)
        ext = (ctypes.c_void_p * 9)()
        for g in range(4):
            n = self.L.nq_group_elems(ctypes.byref(p), nranks, g)
            if n < 0:
                raise RuntimeError("nq_group_elems: nx=%d not divisible over %d ranks" % (nx, nranks))
            if n == 0:
                self.gx.append(None)
                self.gy.append(None)
                continue
            tx = torch.zeros(n, dtype=torch.complex128, device=self.dev)
            ty = torch.zeros(n, dtype=torch.complex128, device=self.dev)
            self.gx.append(tx)
            self.gy.append(ty)
            ext[2 * g], ext[2 * g + 1] = tx.data_ptr(), ty.data_ptr()
        self.sums = torch.zeros(64, dtype=torch.float64, device=self.dev)     # see nq_reduce_buffer
        ext[8] = self.sums.data_ptr()
        if stream is None:
            stream = torch.cuda.current_stream(self.dev).cuda_stream
        kk = np.ascontiguousarray(kk, np.float64)
        ll = np.ascontiguousarray(ll, np.float64)
        filtr = np.ascontiguousarray(filtr, np.float64)
        r = np.ascontiguousarray(np.exp(2j * np.pi * (np.arange(1.0, 33.0) / 32.0))).view(np.float64)
        h = ctypes.c_void_p()
        rc = self.L.nq_create_slab(ctypes.byref(p), _lib._dptr(kk), _lib._dptr(ll), _lib._dptr(filtr), _lib._dptr(r),
                                   device, nranks, rank, ext, ctypes.c_void_p(stream), ctypes.byref(h))
        if rc != 0:
            raise RuntimeError("nq_create_slab failed (%d): %s" % (rc, self.L.nq_last_error(None).decode()))
        self.h = h
        info = (ctypes.c_int * 8)()
        self.L.nq_slab_info(self.h, info)
        (_, _, self.nloc, self.wf, self.kf0, self.wh, self.kh0, self.ph) = list(info)

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed (%d): %s" % (what, rc, self.L.nq_last_error(self.h).decode()))

    def phase(self, ph, stage=0):
        self._chk(self.L.nq_phase(self.h, ph, stage), "nq_phase(%d,%d)" % (ph, stage))

    def upload(self, which, arr):
        arr = np.ascontiguousarray(arr, np.complex128)
        self._chk(self.L.nq_upload_spectral(self.h, which, _lib._dptr(arr.view(np.float64))), "nq_upload_spectral")

    def download(self, which):
        w = self.wh if which == 0 else self.wf
        out = np.empty((self.nx, w), np.complex128)
        self._chk(self.L.nq_download_spectral(self.h, which, _lib._dptr(out.view(np.float64))), "nq_download_spectral")
        return out

    def refresh_grad_phi(self):
        self._chk(self.L.nq_refresh_grad_phi(self.h), "nq_refresh_grad_phi")

    def budget_increments(self):
        out = []
        for sid in (_lib.S_KE, _lib.S_PW, _lib.S_KW):
            v = ctypes.c_double()
            self._chk(self.L.nq_get_scalar(self.h, sid, ctypes.byref(v)), "nq_get_scalar")
            out.append(v.value)
        return out

    def sync(self):
        self._chk(self.L.nq_sync(self.h), "nq_sync")

    def close(self):
        if getattr(self, "h", None):
            self.L.nq_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TorchTransport(object):
    """Real ranks: one SlabRank per process; collectives through torch.distributed (RCCL).

    ``stage_via_host=True`` bounces every buffer through host memory so that a CPU-only backend (gloo) can carry
    the collectives: that is how two processes sharing ONE GPU rehearse the multi-process path in the tests."""

    def __init__(self, dist, stage_via_host=False):
        self.dist = dist
        self.host = bool(stage_via_host)

    def exchange(self, ranks, g, to_y):
        r = ranks[0]
        if r.gx[g] is None:
            return
        send, recv = (r.gx[g], r.gy[g]) if to_y else (r.gy[g], r.gx[g])
        if send.is_complex():                     # RCCL has no complex type: move (re, im) pairs as float64 rows
            send, recv = r.torch.view_as_real(send), r.torch.view_as_real(recv)
        if self.host:
            hs = send.cpu()
            hr = r.torch.empty_like(hs)
            self.dist.all_to_all_single(hr, hs)
            recv.copy_(hr)
            return
        self.dist.all_to_all_single(recv, send)

    def allreduce(self, ranks, lo, hi):
        part = ranks[0].sums[lo:hi]
        if self.host:
            h = part.cpu()
            self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM)
            part.copy_(h)
            return
        self.dist.all_reduce(part, op=self.dist.ReduceOp.SUM)


class VirtualTransport(object):
    """All ranks in this process (same GPU): block (s -> d) copies stand in for the all-to-all."""

    def exchange(self, ranks, g, to_y):
        P = len(ranks)
        if ranks[0].gx[g] is None:
            return
        for d in range(P):
            dst = (ranks[d].gy[g] if to_y else ranks[d].gx[g]).view(P, -1)
            for s in range(P):
                src = (ranks[s].gx[g] if to_y else ranks[s].gy[g]).view(P, -1)
                dst[s].copy_(src[d])

    def allreduce(self, ranks, lo, hi):
        tot = sum(r.sums[lo:hi] for r in ranks)
        for r in ranks:
            r.sums[lo:hi].copy_(tot)


class SlabSimulation(object):
    """Drives the phases and exchanges of one slab-decomposed simulation.

    ``ranks`` holds this process's SlabRank objects: one for real runs, all of them for virtual runs.
    """

    def __init__(self, ranks, transport):
        self.ranks, self.tr = ranks, transport
        self.model = ranks[0].model
        self.coupled = self.model == _lib.COUPLED
        self.waves = self.model != _lib.QG
        self.budgets = ranks[0].budgets

    def _all(self, ph, stage=0):
        for r in self.ranks:
            r.phase(ph, stage)

    # --- initial state (same order semantics as Kernel.set_q / set_phi, quirk Q2) -------------------------
    def set_q_spectrum(self, qh_half):
        """qh_half: full (ny, nx/2+1) half spectrum of q on the host (numpy.fft.rfft2(q))."""
        for r in self.ranks:
            r.upload(0, qh_half[:, r.kh0:r.kh0 + r.wh])
        if self.coupled:
            self._all(PH_WAVEPV)
            self.tr.exchange(self.ranks, 2, True)
        self._all(PH_INVERT_NOW)
        self.tr.exchange(self.ranks, 3, False)
        if self.budgets and self.waves:
            self.tr.allreduce(self.ranks, 48, 51)

    def set_phi_spectrum(self, phih):
        """phih: full (ny, nx) spectrum of phi on the host (numpy.fft.fft2(phi))."""
        for r in self.ranks:
            r.upload(1, phih[:, r.kf0:r.kf0 + r.wf])
        self._all(PH_EMIT_PHI)
        self.tr.exchange(self.ranks, 1, False)
        if self.budgets:
            self.tr.allreduce(self.ranks, 44, 48)
        self.refresh_grad_phi()

    def refresh_grad_phi(self):
        for r in self.ranks:
            r.refresh_grad_phi()

    def set_q(self, q):
        """physical q (ny, nx) on the host (every rank holds the same array)"""
        self.set_q_spectrum(np.fft.rfft2(q))

    def set_phi(self, phi):
        self.set_phi_spectrum(np.fft.fft2(phi))

    def describe(self):
        return "one all_to_all_single per transition (16 per Coupled step), phases dispatched from Python"

    def reset_counters(self):
        pass

    def counters(self, nsteps):
        """volume this rank hands to the all-to-alls per step (off-rank part), for the xGMI arithmetic in DESIGN.md 9"""
        P = self.ranks[0].nranks
        per_stage = sum(t.numel() * 16 for t in self.ranks[0].gx if t is not None)
        return {"exchange_GB_sent_per_rank_per_step": 4 * per_stage * (P - 1) / max(P, 1) / 1e9}

    # --- time stepping --------------------------------------------------------------------------------------
    def step(self, nsteps=1):
        tr, ranks = self.tr, self.ranks
        for _ in range(nsteps):
            for s in range(4):
                self._all(PH_PRODUCTS, s)
                tr.exchange(ranks, 0, True)
                self._all(PH_UPDATE, s)
                if self.waves:
                    tr.exchange(ranks, 1, False)
                if self.coupled:
                    self._all(PH_WAVEPV)
                    tr.exchange(ranks, 2, True)
                    self._all(PH_INVERT, s)
                tr.exchange(ranks, 3, False)
            if self.budgets:
                self._all(PH_BUDGET_SUMS)
                tr.allreduce(ranks, 0, 44)
                self._all(PH_BUDGET_FINISH)

    def sync(self):
        for r in self.ranks:
            r.sync()

    # --- gathering (tests / output): only meaningful when this process holds every rank --------------------
    def gather_qh(self):
        return np.concatenate([r.download(0) for r in self.ranks], axis=1)

    def gather_phih(self):
        return np.concatenate([r.download(1) for r in self.ranks], axis=1)


def make_ranks(model, nx, kk, ll, filtr, dt, nranks, device=0, only_rank=None, budgets=True, **phys):
    """All ranks on one device (virtual) or just `only_rank` (real run, one process per GPU)."""
    which = range(nranks) if only_rank is None else [only_rank]
    return [SlabRank(model, nx, kk, ll, filtr, dt, nranks, r, device, budgets=budgets, **phys) for r in which]
