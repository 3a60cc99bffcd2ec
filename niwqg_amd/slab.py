"""1-D slab decomposition of ONE simulation over several GPUs (one process per GPU, RCCL over xGMI).

Physical / mixed-space rows are split over ranks on the "x side" (row kernels), spectral / mixed-space columns on the
"y side" (spectral kernels); the arrays that cross together form four exchange groups.  The whole
step -- phases AND exchanges -- runs inside the library (``nq_slab_step``): one host call per ``step(n)``, the exchanges on
a second HIP stream, cut into row chunks so that row kernels run under the transfers.  How the blocks cross is the
context's *link*:

  * ``"rccl"``      real ranks: grouped ncclSend/ncclRecv issued by the library itself; ``torch.distributed`` is used
                    once, to hand rank 0's 128-byte unique id to the other ranks;
  * ``"peers"``     all ranks live in this process on one GPU (the one-GPU test double): the same choreography of
                    streams and events with device-to-device copies for the wire;
  * ``"callback"``  the library calls back into Python at every exchange and ``torch.distributed`` moves the buffers
                    (``all_to_all_single`` / ``all_reduce``), optionally staged through host memory so that a CPU-only
                    backend (gloo) can carry them: how two processes sharing ONE GPU rehearse the multi-process path.

``set_q`` / ``set_phi`` take physical fields and transform them on the devices: every rank uploads only its own rows.
"""
import ctypes

import numpy as np

from . import _etdrk4, _lib

(PH_PRODUCTS, PH_UPDATE, PH_WAVEPV, PH_INVERT, PH_EMIT_PHI, PH_INVERT_NOW, PH_BUDGET_SUMS,
 PH_BUDGET_FINISH) = range(8)

_REAL_ROWS = (_lib.F_Q, _lib.F_P, _lib.F_U, _lib.F_V, _lib.F_QW, _lib.F_C)
_CPLX_ROWS = (_lib.F_PHI, _lib.F_PHIX, _lib.F_PHIY)


class SlabRank(object):
    """One rank's device context.  The exchange buffers belong to the library unless ``torch_buffers`` asks for torch
    tensors (the callback link moves them with torch.distributed)."""

    def __init__(self, model, nx, kk, ll, filtr, dt, nranks, rank, device, budgets=True, torch_buffers=False, **phys):
        self.L = _lib.lib()
        self.model, self.nx, self.nranks, self.rank, self.device = model, int(nx), int(nranks), int(rank), int(device)
        p = _lib.Params(model=model, nx=nx, budgets=int(bool(budgets)), dual_q=int(bool(phys.get("dual_q", False))), dt=dt,
                        U=phys.get("U", 0.0), f=phys.get("f", 1e-4), kappa2=phys.get("kappa2", 1.0),
                        nu=phys.get("nu", 0.0), nu4=phys.get("nu4", 0.0), mu=phys.get("mu", 0.0),
                        nuw=phys.get("nuw", 0.0), nu4w=phys.get("nu4w", 0.0), muw=phys.get("muw", 0.0),
                        beta=phys.get("beta", 0.0),
                        passive_scalar=int(bool(phys.get("passive_scalar", False)) and model == _lib.QG),
                        nu4c=phys.get("nu4c", 0.0), nuc=phys.get("nuc", 0.0), muc=phys.get("muc", 0.0))
        self.budgets = bool(budgets)
        self.gx, self.gy, self.sums = [None] * 5, [None] * 5, None
        ext = None
        if torch_buffers:
            import torch
            self.torch = torch
            dev = torch.device("cuda", device)
            ext = (ctypes.c_void_p * 9)()
            for g in range(4):
                n = self.L.nq_group_elems(ctypes.byref(p), nranks, g)
                if n < 0:
                    raise RuntimeError("nq_group_elems: nx=%d not divisible over %d ranks" % (nx, nranks))
                if n == 0:
                    continue
                self.gx[g] = torch.zeros(n, dtype=torch.complex128, device=dev)
                self.gy[g] = torch.zeros(n, dtype=torch.complex128, device=dev)
                ext[2 * g], ext[2 * g + 1] = self.gx[g].data_ptr(), self.gy[g].data_ptr()
            self.sums = torch.zeros(64, dtype=torch.float64, device=dev)     # see nq_reduce_buffer
            ext[8] = self.sums.data_ptr()
            torch.cuda.synchronize(dev)          # the zero fills ran on torch's stream; the library has its own
        kk = np.ascontiguousarray(kk, np.float64)
        ll = np.ascontiguousarray(ll, np.float64)
        filtr = np.ascontiguousarray(filtr, np.float64)
        r = np.ascontiguousarray(np.exp(2j * np.pi * (np.arange(1.0, 33.0) / 32.0))).view(np.float64)
        h = ctypes.c_void_p()
        # stream NULL: the context makes its own compute stream; everything that has to be ordered against it goes
        # through the library (exchange stream + events) or happens with that stream drained (callbacks)
        rc = self.L.nq_create_slab(ctypes.byref(p), _lib._dptr(kk), _lib._dptr(ll), _lib._dptr(filtr), _lib._dptr(r),
                                   device, nranks, rank, ext, None, ctypes.byref(h))
        if rc != 0:
            raise RuntimeError("nq_create_slab failed (%d): %s" % (rc, self.L.nq_last_error(None).decode()))
        self.h = h
        info = (ctypes.c_int * 8)()
        self.L.nq_slab_info(self.h, info)
        (_, _, self.nloc, self.wf, self.kf0, self.wh, self.kh0, self.ph) = list(info)
        # this rank's columns of the contour-adjacent ETDRK4 entries, recomputed as the reference computes them (_etdrk4.py)
        prm = {k: float(getattr(p, k)) for k in ("U", "f", "kappa2", "nu", "nu4", "mu", "nuw", "nu4w", "muw", "beta", "nu4c", "nuc", "muc")}
        eqs = [0] + ([1] if model != _lib.QG else []) + ([2] if p.passive_scalar else [])
        self.contour_patched = _etdrk4.patch_near_contour(
            lambda eq, delta: _lib.coeff_near_contour(self.L, self.h, eq, delta),
            lambda eq, li, ki, v: _lib.coeff_patch(self.L, self.h, eq, li, ki, v), model, self.nx, kk, ll, filtr, dt, prm, eqs)
        if torch_buffers and model == _lib.YBJ and nranks > 1:          # the stage-result group of YBJModel's step
            n = self.L.nq_group_elems(ctypes.byref(p), nranks, 4)
            self.gx[4] = torch.zeros(n, dtype=torch.complex128, device=dev)
            self.gy[4] = torch.zeros(n, dtype=torch.complex128, device=dev)
            torch.cuda.synchronize(dev)
            self._chk(self.L.nq_slab_set_stage_buffers(self.h, ctypes.c_void_p(self.gx[4].data_ptr()),
                                                       ctypes.c_void_p(self.gy[4].data_ptr())), "nq_slab_set_stage_buffers")

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed (%d): %s" % (what, rc, self.L.nq_last_error(self.h).decode()))

    def phase(self, ph, stage=0):
        self._chk(self.L.nq_phase(self.h, ph, stage), "nq_phase(%d,%d)" % (ph, stage))

    def upload(self, which, arr):
        arr = np.ascontiguousarray(arr, np.complex128)
        self._chk(self.L.nq_upload_spectral(self.h, which, _lib._dptr(arr.view(np.float64))), "nq_upload_spectral")

    def download(self, which):
        w = self.wf if which in (1, 7, 10) else self.wh     # 0: qh, 2: ph, 3: qwh ... (half-spectrum slabs); 1, 7, 10: phih
        out = np.empty((self.nx, w), np.complex128)
        self._chk(self.L.nq_download_spectral(self.h, which, _lib._dptr(out.view(np.float64))), "nq_download_spectral")
        return out

    def qh_passenger(self):
        """this rank's columns of the anti-Hermitian passenger row of qh (include/niwqg_amd.h: nq_get_qh_passenger)"""
        out = np.zeros(max(self.wh, 0), np.complex128)
        if self.wh > 0:
            self._chk(self.L.nq_get_qh_passenger(self.h, _lib._dptr(out.view(np.float64))), "nq_get_qh_passenger")
        return out

    def put_rows(self, which, rows):
        rows = np.ascontiguousarray(rows, np.complex128 if which == 1 else np.float64)     # 0: q, 1: phi, 2: c
        if rows.shape != (self.nloc, self.nx):
            raise ValueError("put_rows: rows of shape %s, this rank holds %s" % (rows.shape, (self.nloc, self.nx)))
        self._chk(self.L.nq_slab_put_rows(self.h, which, _lib._dptr(rows.view(np.float64))), "nq_slab_put_rows")

    def get_rows(self, fid):
        out = np.empty((self.nloc, self.nx), np.float64 if fid in _REAL_ROWS else np.complex128)
        self._chk(self.L.nq_slab_get_rows(self.h, fid, _lib._dptr(out.view(np.float64))), "nq_slab_get_rows(%d)" % fid)
        return out

    def stage4_max(self):
        out = np.zeros(2)
        self._chk(self.L.nq_get_stage4_max(self.h, _lib._dptr(out)), "nq_get_stage4_max")
        return out

    def local_max(self):
        out = np.zeros(3)
        self._chk(self.L.nq_slab_local_max(self.h, _lib._dptr(out)), "nq_slab_local_max")
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


def all_to_all_blocks(dist, torch, send, recv, stage_via_host):
    """one exchange group: P equal blocks of ``send`` go to the P ranks, block s of ``recv`` comes from rank s"""
    if send.is_complex():                     # RCCL / gloo have no complex type: move (re, im) pairs as float64 rows
        send, recv = torch.view_as_real(send), torch.view_as_real(recv)
    if stage_via_host:
        hs = send.cpu()
        hr = torch.empty_like(hs)
        dist.all_to_all_single(hr, hs)
        recv.copy_(hr)
    else:
        dist.all_to_all_single(recv, send)


def reference_all_to_all(sends):
    """what P ranks hold after an all-to-all of their ``sends`` (list of 1-D arrays of P equal blocks): pure numpy"""
    P = len(sends)
    blocks = [np.asarray(s).reshape(P, -1) for s in sends]
    return [np.concatenate([blocks[s][d] for s in range(P)]) for d in range(P)]


def agree_rccl_id(L, lead, dist):
    """The communicator id of the RCCL link, or None when ANY rank cannot use librccl -- decided by collectives that every
    rank takes part in whatever happened locally, so that no rank is ever left alone in a broadcast or in
    ncclCommInitRank: (1) every rank probes the library (load only) and the outcomes are MIN-reduced; (2) rank 0 asks
    for the id and ALWAYS broadcasts 129 bytes, the last one saying whether the id is real."""
    import torch
    on_dev = dist.get_backend() == "nccl"
    dev = torch.device("cuda", lead.device) if on_dev else torch.device("cpu")
    ok = torch.tensor([1.0 if L.nq_comm_probe() == 0 else 0.0], device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if float(ok[0]) != 1.0:
        return None
    msg = torch.zeros(129, dtype=torch.uint8)
    if lead.rank == 0:
        buf = (ctypes.c_ubyte * 128)()
        if L.nq_comm_unique_id(buf) == 0:
            msg = torch.tensor(list(buf) + [1], dtype=torch.uint8)
    msg = msg.to(dev)
    dist.broadcast(msg, src=0)
    msg = [int(v) for v in msg.cpu()]
    return msg[:128] if msg[128] == 1 else None


class SlabSimulation(object):
    """One slab-decomposed simulation: this process's ranks (one real rank, or all of them as peers) and their link."""

    def __init__(self, ranks, link="peers", dist=None, nchunks=2, stage_via_host=False, uid=None):
        self.ranks, self.link, self.dist = ranks, link, dist
        self.L = ranks[0].L
        self.model = ranks[0].model
        self.coupled = self.model == _lib.COUPLED
        self.waves = self.model != _lib.QG
        self.budgets = ranks[0].budgets
        self.nranks, self.nx = ranks[0].nranks, ranks[0].nx
        self.lead = ranks[0]
        if link == "peers":
            if len(ranks) != self.nranks:
                raise ValueError("peers link: this process must hold all %d ranks" % self.nranks)
            arr = (ctypes.c_void_p * self.nranks)(*[r.h for r in ranks])
            self.lead._chk(self.L.nq_slab_attach_peers(arr, self.nranks), "nq_slab_attach_peers")
        elif link == "rccl":
            if len(ranks) != 1 or dist is None:
                raise ValueError("rccl link: one rank per process and a torch.distributed group")
            if uid is None:
                uid = agree_rccl_id(self.L, self.lead, dist)
            if uid is None:
                raise RuntimeError("rccl link: librccl could not be set up on every rank (agreed over the process group)")
            buf = (ctypes.c_ubyte * 128)(*uid)
            self.lead._chk(self.L.nq_comm_init(self.lead.h, buf, self.nranks, self.lead.rank), "nq_comm_init")
        elif link == "callback":
            import torch
            if len(ranks) != 1 or dist is None or self.lead.gx[0] is None:
                raise ValueError("callback link: one rank per process, a torch.distributed group and torch buffers (torch_buffers=True)")
            r, host = self.lead, bool(stage_via_host)
            dev = torch.device("cuda", r.device)

            def exchange(user, g, to_y):
                try:
                    send, recv = (r.gx[g], r.gy[g]) if to_y else (r.gy[g], r.gx[g])
                    all_to_all_blocks(dist, torch, send, recv, host)
                    torch.cuda.synchronize(dev)          # complete before the library's stream goes on
                    return 0
                except Exception as e:                   # never let an exception cross the C boundary
                    self._cb_error = e
                    return 1

            def allreduce(user, which):
                try:
                    n = {0: 44, 1: 4, 2: 3, 3: 1, 4: 16, 5: 16}[which]
                    buf = np.zeros(n)
                    r._chk(self.L.nq_reduce_read(r.h, which, _lib._dptr(buf)), "nq_reduce_read")
                    t = torch.from_numpy(buf)
                    if not host:
                        t = t.to(dev)
                    dist.all_reduce(t, op=dist.ReduceOp.SUM)
                    buf = np.ascontiguousarray(t.cpu().numpy())
                    r._chk(self.L.nq_reduce_write(r.h, which, _lib._dptr(buf)), "nq_reduce_write")
                    return 0
                except Exception as e:
                    self._cb_error = e
                    return 1

            self._cb_error = None
            self._xcb = _lib.EXCHANGE_FN(exchange)       # keep the trampolines alive as long as the simulation
            self._rcb = _lib.ALLREDUCE_FN(allreduce)
            r._chk(self.L.nq_slab_set_callbacks(r.h, self._xcb, self._rcb, None), "nq_slab_set_callbacks")
        elif link == "null":
            # measurement aid (bench.py --rank-of P): ONE rank of the decomposition, alone; nothing crosses but its own block
            if len(ranks) != 1:
                raise ValueError("null link: exactly one rank")
            self.lead._chk(self.L.nq_slab_set_null_link(self.lead.h), "nq_slab_set_null_link")
        else:
            raise ValueError("link %r" % (link,))
        for r in ranks:
            r._chk(self.L.nq_slab_config(r.h, int(nchunks)), "nq_slab_config")

    def _lead_chk(self, rc, what):
        if rc != 0 and getattr(self, "_cb_error", None) is not None:
            e, self._cb_error = self._cb_error, None
            raise RuntimeError("%s: the exchange callback failed: %r" % (what, e))
        self.lead._chk(rc, what)

    def describe(self):
        how = {"rccl": "grouped ncclSend/ncclRecv issued inside the library",
               "peers": "peer ranks in one process (device / peer copies%s)" % (
                   "" if len(set(r.device for r in self.ranks)) == 1 else " across %d devices" % len(set(r.device for r in self.ranks))),
               "callback": "torch.distributed all_to_all_single from a library callback",
               "null": "NO exchange (one rank of the decomposition measured alone)"}[self.link]
        return "in-library step, %s, %d row chunks per exchange" % (how, int(self.counters()["nchunks"]))

    # --- initial state (same order semantics as Kernel.set_q / set_phi, quirk Q2) -------------------------
    def _set(self, which, field):
        for r in self.ranks:
            r.put_rows(which, field[r.rank * r.nloc:(r.rank + 1) * r.nloc])
        self._lead_chk(self.L.nq_slab_commit(self.lead.h, which), "nq_slab_commit")

    def set_q(self, q):
        """physical q (ny, nx); every rank uploads only its own rows (the argument may also be a view of them: anything
        indexable by the global row range of the local ranks)"""
        self._set(0, q)

    def set_phi(self, phi):
        self._set(1, phi)

    def set_c(self, c):
        """QGModel's passive scalar (QGModel.set_c): rows in, then the inversion re-emits every mixed-space row"""
        for r in self.ranks:
            r.put_rows(2, c[r.rank * r.nloc:(r.rank + 1) * r.nloc])
        self._lead_chk(self.L.nq_slab_commit(self.lead.h, 3), "nq_slab_commit(set_c)")

    def refresh_grad_phi(self):
        for r in self.ranks:
            r.refresh_grad_phi()

    # --- time stepping --------------------------------------------------------------------------------------
    def step(self, nsteps=1):
        self._lead_chk(self.L.nq_slab_step(self.lead.h, int(nsteps)), "nq_slab_step")

    def sync(self):
        for r in self.ranks:
            r.sync()

    def diagnostics(self):
        out = np.zeros(32)
        self._lead_chk(self.L.nq_slab_diagnostics(self.lead.h, _lib._dptr(out)), "nq_slab_diagnostics")
        return out

    def max_over_ranks(self, values):
        """element-wise max over all ranks of the simulation of a small vector this process computed for its ranks"""
        v = np.max(np.asarray(values, float).reshape(len(self.ranks), -1), axis=0)
        if self.link not in ("peers", "null"):
            import torch
            self.sync()        # never two communicators active on the device (nq_sync drains the exchange stream too)
            t = torch.from_numpy(v.copy())
            if self.dist.get_backend() == "nccl":
                t = t.to(torch.device("cuda", self.lead.device))
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            v = t.cpu().numpy()
        return v

    def cfl_max(self):
        """max over the whole grid of |u|, |v|, |phi| (Kernel._calc_cfl without dt/dx)"""
        return float(np.max(self.max_over_ranks([r.local_max() for r in self.ranks])))

    def request_stage4_max(self):
        self.lead._chk(self.L.nq_request_stage4_max(self.lead.h), "nq_request_stage4_max")

    def tick_snapshot(self):
        for r in self.ranks:
            r._chk(self.L.nq_tick_snapshot(r.h), "nq_tick_snapshot")

    def status_cfl_max(self):
        """Context.status_cfl_max over the ranks: the fourth stage's max |u|, |v| with the new state's max |phi|"""
        return float(np.max(self.max_over_ranks([list(r.stage4_max()) + [r.local_max()[2]] for r in self.ranks])))

    def counters(self, reset=0):
        red = np.zeros(2)
        self.lead._chk(self.L.nq_slab_allreduce_ms(self.lead.h, _lib._dptr(red)), "nq_slab_allreduce_ms")
        out = np.zeros(6)
        self.lead._chk(self.L.nq_slab_counters(self.lead.h, _lib._dptr(out), int(reset)), "nq_slab_counters")
        return dict(host_calls=out[0], steps=out[1], exchange_chunks=out[2], bytes_sent=out[3], exchange_ms=out[4], nchunks=out[5],
                    allreduce_ms=red[0], allreduces=red[1])

    # --- gathering -----------------------------------------------------------------------------------------
    def _gather(self, parts, axis):
        """parts: this process's pieces in rank order -> the whole array on every rank"""
        if self.link in ("peers", "null"):
            return np.concatenate(parts, axis=axis)
        import torch
        self.sync()            # the library's streams (compute AND exchange) are drained before torch's communicator runs
        mine = np.ascontiguousarray(parts[0])
        bufs = [None] * self.nranks
        self.dist.all_gather_object(bufs, mine)
        return np.concatenate(bufs, axis=axis)

    def gather_rows(self, fid):
        return self._gather([r.get_rows(fid) for r in self.ranks], 0)

    def invert(self):
        """Kernel._invert on the current state"""
        self._lead_chk(self.L.nq_slab_commit(self.lead.h, 2), "nq_slab_commit(invert)")

    def gather_spectral(self, which):
        """0: qh, 1: phih, 2: ph, 3: qwh, 4: second copy of qh, 5: ch -- the column slabs of all ranks side by side"""
        return self._gather([r.download(which) for r in self.ranks], 1)

    def gather_qh_passenger(self):
        return self._gather([r.qh_passenger() for r in self.ranks], 0)

    def gather_qh(self):
        return self._gather([r.download(0) for r in self.ranks], 1)

    def gather_phih(self):
        return self._gather([r.download(1) for r in self.ranks], 1)


def make_ranks(model, nx, kk, ll, filtr, dt, nranks, device=0, only_rank=None, budgets=True, torch_buffers=False, **phys):
    """All ranks in this process (peers: on one device, or -- `device` a list -- rank r on device[r]) or just `only_rank` (real run,
    one process per GPU)."""
    which = range(nranks) if only_rank is None else [only_rank]
    dev = (lambda r: device[r]) if isinstance(device, (list, tuple)) else (lambda r: device)
    return [SlabRank(model, nx, kk, ll, filtr, dt, nranks, r, dev(r), budgets=budgets, torch_buffers=torch_buffers, **phys)
            for r in which]


def connect(ranks, dist, nchunks=2):
    """The multi-process simulation of this rank: RCCL issued by the library when it can be set up on EVERY rank, else
    (or with NIWQG_AMD_SLAB_LINK=callback, or on a gloo group) torch.distributed moves the buffers from library callbacks.
    All ranks end up on the same link: the outcome of the RCCL set-up is agreed on through the process group."""
    import os
    import torch
    gloo = dist.get_backend() == "gloo"
    if gloo:
        return SlabSimulation(ranks, "callback", dist=dist, nchunks=nchunks, stage_via_host=True)
    sim, err = None, None
    if os.environ.get("NIWQG_AMD_SLAB_LINK", "rccl") == "rccl":
        # agreement BEFORE any collective set-up call (agree_rccl_id), then ncclCommInitRank on every rank or on none
        uid = agree_rccl_id(ranks[0].L, ranks[0], dist)
        if uid is not None:
            try:
                sim = SlabSimulation(ranks, "rccl", dist=dist, nchunks=nchunks, uid=uid)
            except RuntimeError as e:
                err = e
            flag = torch.tensor([1.0 if sim is not None else 0.0], device=torch.device("cuda", ranks[0].device))
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if float(flag[0]) == 1.0:
                return sim
        else:
            err = ranks[0].L.nq_last_error(None).decode(errors="replace")
        import sys
        sys.stderr.write("niwqg_amd.slab rank %d: RCCL link not available on every rank (%s): falling back to "
                         "torch.distributed callbacks\n" % (ranks[0].rank, err or "another rank failed"))
    return SlabSimulation(ranks, "callback", dist=dist, nchunks=nchunks, stage_via_host=False)


class SlabContext(object):
    """What the model classes (niwqg_amd.Kernel / QGModel) need from a device context, on top of a slab-decomposed
    simulation: the same methods as ``_lib.Context``, fields gathered over the ranks on demand.  ``peers`` = P puts all P
    ranks into this process on one GPU (tests); otherwise one rank per process, launched with torch.distributed.run."""

    def __init__(self, model, nx, kk, ll, filtr, dt, peers=None, nchunks=2, device=None, budgets=True, **phys):
        self.model, self.nx = model, int(nx)
        self._kk, self._ll = np.array(kk, np.float64), np.array(ll, np.float64)
        self.budgets_enabled = bool(budgets)
        self.kappa2 = phys.get("kappa2", 1.0)
        if peers:
            ranks = make_ranks(model, nx, kk, ll, filtr, dt, int(peers), device=device or 0, budgets=budgets, **phys)
            self.sim = SlabSimulation(ranks, "peers", nchunks=nchunks)
            self.group = None
        else:
            from .distributed import Group
            g = self.group = Group()
            if g.dist is None:
                raise RuntimeError("slab model: no process group -- launch with torch.distributed.run (WORLD_SIZE > 1) or pass slab=P")
            import torch
            dev = g.local_rank % max(torch.cuda.device_count(), 1) if device is None else device
            ranks = make_ranks(model, nx, kk, ll, filtr, dt, g.world, device=dev, only_rank=g.rank, budgets=budgets,
                               torch_buffers=True, **phys)          # torch tensors: usable by either link
            self.sim = connect(ranks, g.dist, nchunks)
        self._ds = None

    # --- state
    def _touch(self):
        self._ds = None

    def set_q(self, q):
        q = np.asarray(q, np.float64)
        _lib.Context._shape(q, (self.nx, self.nx), "set_q")
        self.sim.set_q(q)
        self._touch()

    def set_phi(self, phi):
        phi = np.asarray(phi, np.complex128)
        _lib.Context._shape(phi, (self.nx, self.nx), "set_phi")
        self.sim.set_phi(phi)
        self._touch()

    def set_c(self, c):
        c = np.asarray(c, np.float64)
        _lib.Context._shape(c, (self.nx, self.nx), "set_c")
        self.sim.set_c(c)
        self._touch()

    def invert(self):
        self.sim.invert()
        self._touch()

    def refresh_grad_phi(self):
        self.sim.refresh_grad_phi()

    def step(self, n=1):
        self.sim.step(n)
        self._touch()

    def sync(self):
        self.sim.sync()

    def request_stage4_max(self):
        self.sim.request_stage4_max()

    def tick_snapshot(self):
        self.sim.tick_snapshot()

    def status_cfl_max(self):
        return self.sim.status_cfl_max()

    def take_budget_increments(self):
        incs = [r.budget_increments() for r in self.sim.ranks]       # identical on every rank; reading resets each
        return tuple(incs[0])

    # --- reads
    def field(self, fid):
        L = _lib
        if fid in _REAL_ROWS or fid in _CPLX_ROWS:
            return self.sim.gather_rows(fid)
        if fid == L.F_QPSI:
            q = self.sim.gather_rows(L.F_Q)
            return q - self.sim.gather_rows(L.F_QW) if self.model == L.COUPLED else q
        which = {L.F_QH: 0, L.F_PHIH: 1, L.F_PH: 2, L.F_QWH: 3, L.F_QH_MINUS: 4, L.F_CH: 5, L.F_QH_STAGE4: 6, L.F_PHIH_STAGE4: 7,
                 L.F_QH_MINUS_STAGE4: 8, L.F_QH_TICK: 9, L.F_PHIH_TICK: 10, L.F_QH_MINUS_TICK: 11, L.F_QWH_TICK: 12}.get(fid)
        if which is None:
            raise RuntimeError("field %d is not available on a slab-decomposed model" % fid)
        return self.sim.gather_spectral(which)

    def qh_passenger(self):
        return self.sim.gather_qh_passenger()

    def diagnostic_sums(self):
        if self._ds is None:
            self._ds = self.sim.diagnostics()
        return self._ds

    def scalar(self, sid):
        L = _lib
        M2 = float(self.nx) ** 4
        if sid in (L.S_KE, L.S_PW, L.S_KW):
            out = None
            for r in self.sim.ranks:                                    # read (and reset) on every local rank
                v = ctypes.c_double()
                r._chk(r.L.nq_get_scalar(r.h, sid, ctypes.byref(v)), "nq_get_scalar(%d)" % sid)
                out = v.value if out is None else out
            return out
        if sid == L.S_CFL:
            return self.sim.cfl_max()
        ds = self.diagnostic_sums()
        if sid == L.S_KE_QG:
            return 0.5 * ds[11] / M2
        if sid == L.S_KE_NIW:
            return 0.5 * ds[0] / M2
        if sid == L.S_PE_NIW:
            return 0.25 * ds[1] / M2 / self.kappa2
        raise RuntimeError("scalar %d is not available on a slab-decomposed model" % sid)

    def _single_rank_only(self, *a, **k):
        raise NotImplementedError("this call needs the whole plane on one device: not available on a slab-decomposed model "
                                  "(use a single-GPU model of the same parameters)")

    coeff = _single_rank_only

    # --- the whole-plane calls of the class API: global arrays in and out on every rank, transforms and row kernels on the
    # slabs (nq_slab_spectral).  Inverse transforms go through the forward path, ifft(X) = conj(fft(conj X)) / N^2: only
    # the x -> y exchange groups are free between steps.
    def _spectral(self, what, half):
        sim = self.sim
        sim._lead_chk(sim.L.nq_slab_spectral(sim.lead.h, int(what)), "nq_slab_spectral(%d)" % what)
        parts = []
        for r in sim.ranks:
            out = np.empty((self.nx, r.wh if half else r.wf), np.complex128)
            r._chk(r.L.nq_slab_spectral_read(r.h, int(half), _lib._dptr(out.view(np.float64))), "nq_slab_spectral_read")
            parts.append(out)
        return sim._gather(parts, 1)

    def rfft2(self, a):
        a = np.asarray(a, np.float64)
        _lib.Context._shape(a, (self.nx, self.nx), "rfft2")
        for r in self.sim.ranks:
            r.put_rows(0, a[r.rank * r.nloc:(r.rank + 1) * r.nloc])
        return self._spectral(5, True)

    def fft2(self, a):
        a = np.asarray(a, np.complex128)
        _lib.Context._shape(a, (self.nx, self.nx), "fft2")
        if self.model == _lib.QG:                      # no complex carrier in QGModel's exchange groups: two real transforms
            from .Kernel import hermitian_full
            return hermitian_full(self.rfft2(a.real)) + 1j * hermitian_full(self.rfft2(a.imag))
        for r in self.sim.ranks:
            r.put_rows(1, a[r.rank * r.nloc:(r.rank + 1) * r.nloc])
        return self._spectral(6, False)

    def ifft2(self, a):
        return np.conj(self.fft2(np.conj(np.asarray(a, np.complex128)))) / float(self.nx) ** 2

    def irfft2(self, a):
        """numpy.fft.irfft2 semantics: the self-mirrored columns count with their Hermitian part (in l) only"""
        from .Kernel import hermitian_full, project_self_mirrored_columns
        a = np.asarray(a, np.complex128)
        _lib.Context._shape(a, (self.nx, self.nx // 2 + 1), "irfft2")
        return np.ascontiguousarray(self.ifft2(hermitian_full(project_self_mirrored_columns(a))).real)

    def products_uq_vq(self):
        """fft(u q), fft(v q) on k = 0..nx/2"""
        return self._spectral(0, True), self._spectral(1, True)

    def jacobian_psi_q(self):
        """Kernel family: (ny, nx) with [0,0] = 0 (Kernel.py:471-486); QGModel: (ny, nx/2+1) (QGModel.py:469-481)"""
        from .Kernel import hermitian_full
        a, b = self.products_uq_vq()
        ll = self._ll[:, None]
        if self.model == _lib.QG:
            return 1j * self._kk[None, :] * a + 1j * ll * b
        out = 1j * self._kk[None, :] * hermitian_full(a) + 1j * ll * hermitian_full(b)
        out[0, 0] = 0.0
        return out

    def jacobian_psi_phi(self):
        out = self._spectral(2, False)
        if self.model != _lib.YBJ:                     # YBJModel's own version keeps [0,0] (YBJModel.py:123-133)
            out[0, 0] = 0.0
        return out

    def refraction(self):
        """fft(phi * q_psi) from the row kernel (Kernel.py:332 without the -0.5j)"""
        return -1j * self._spectral(3, False)

    def jacobian_phic_phi(self):
        from .Kernel import hermitian_full
        out = hermitian_full(self._spectral(4, True))
        out[0, 0] = 0.0
        return out

    def close(self):
        for r in self.sim.ranks:
            r.close()
