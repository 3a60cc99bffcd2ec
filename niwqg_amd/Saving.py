"""Output of set-up, snapshots and diagnostics with the reference's on-disk layout (ref: niwqg/Saving.py:6-101):

    <path>/setup.h5                     grid/{nx, x, y, wv, k, l}
    <path>/snapshots/%015.0f.h5         t, q, phi            (file name = model time; QGModel: t, q[, c])
    <path>/diagnostics.h5               one dataset per registered diagnostic

Files are written through ``h5py`` when it can be imported.  It is absent from this image, so the writer is a seam:
``set_writer(factory)`` installs any ``factory(filename) -> object with create_dataset(name, data=...) and close()``
(tests use a recording stub; the NpzWriter below keeps the same dataset names in a numpy archive).  Without h5py and
without an installed writer ``save_to_disk=True`` raises at construction -- never silently drops output.

Snapshots do not stall the time stepping: the device forms q and phi in physical space into buffers of their own, a
second stream copies them to pinned host memory (nq_snapshot_begin), and the file is written when the host next looks
(nq_snapshot_end), usually while the following batch of steps is already running on the GPU.

A model that is slab-decomposed over several processes gathers its fields collectively (every rank takes part) and rank 0
alone touches the files.
"""
import os

import numpy as np

_writer_factory = None


def set_writer(factory):
    """Install (or with None: remove) the file writer used instead of h5py.File(fno, 'w')."""
    global _writer_factory
    _writer_factory = factory


class NpzWriter(object):
    """Same dataset names, numpy archive on disk (the file keeps the reference's name, '.h5' included)."""

    def __init__(self, fno):
        self.fno, self.data = fno, {}

    def create_dataset(self, name, data=None, dtype=None):
        self.data[name] = np.asarray(data) if dtype is None else np.asarray(data, dtype=dtype)

    def close(self):
        with open(self.fno, "wb") as f:
            np.savez(f, **{k.replace("/", "__"): v for k, v in self.data.items()})


def writer_available():
    if _writer_factory is not None:
        return True
    try:
        import h5py  # noqa: F401
        return True
    except Exception:
        return False


def _open(fno):
    if _writer_factory is not None:
        return _writer_factory(fno)
    import h5py
    return h5py.File(fno, 'w')


def _writes(self):
    """False on the ranks > 0 of a model decomposed over several processes"""
    group = getattr(getattr(self, "_ctx", None), "group", None)
    return group is None or group.rank == 0


def initialize_save_snapshots(self, path):
    """ref: niwqg/Saving.py:6-22"""
    self.fno = path
    self._pending_snapshots = []
    if self.save_to_disk and not writer_available():
        raise NotImplementedError("save_to_disk=True: h5py is not importable and no writer is installed "
                                  "(niwqg_amd.Saving.set_writer)")
    if (not os.path.isdir(self.fno)) and self.save_to_disk and _writes(self):
        os.makedirs(self.fno)
        os.makedirs(self.fno + "/snapshots/")


def file_exist(fno, overwrite=True):
    """ref: niwqg/Saving.py:24-36"""
    if os.path.exists(fno):
        if overwrite:
            os.remove(fno)
        else:
            raise IOError("File exists: {0}".format(fno))


def save_setup(self):
    """ref: niwqg/Saving.py:38-57"""
    if self.save_to_disk and _writes(self):
        fno = self.fno + '/setup.h5'
        file_exist(fno, overwrite=self.overwrite)
        h5file = _open(fno)
        h5file.create_dataset("grid/nx", data=(self.nx), dtype=int)
        h5file.create_dataset("grid/x", data=(self.x))
        h5file.create_dataset("grid/y", data=(self.y))
        h5file.create_dataset("grid/wv", data=self.wv)
        h5file.create_dataset("grid/k", data=self.kk)
        h5file.create_dataset("grid/l", data=self.ll)
        h5file.close()


def save_snapshots(self, fields=['t', 'q', 'p']):
    """ref: niwqg/Saving.py:59-86.  Inside run() (``self._defer_snapshots``) q and phi leave the device asynchronously and the
    file is written by flush_snapshots while the next batch of steps runs; called from anywhere else (_step_forward in a
    user loop, run_with_snapshots) the file exists when this returns, as in the reference."""
    if (not (self.tc % self.tsnaps)) and self.save_to_disk:
        fno = self.fno + '/snapshots/{:015.0f}'.format(self.t) + '.h5'
        flush_snapshots(self)                       # one snapshot in flight at a time (one set of pinned buffers)
        want = [f for f in fields if f in ("q", "phi")]
        ctx = self._ctx
        asynchronous = bool(want) and hasattr(ctx, "snapshot_begin")
        if asynchronous:
            ctx.snapshot_begin("phi" in want)
        other = {f: (self.t if f == 't' else np.array(getattr(self, f))) for f in fields if not (asynchronous and f in want)}
        self._pending_snapshots.append((fno, list(fields), other, asynchronous))
        if not getattr(self, "_defer_snapshots", False):
            flush_snapshots(self)


def flush_snapshots(self):
    """Write the snapshots whose device-to-host copies were started by save_snapshots."""
    for fno, fields, other, asynchronous in self._pending_snapshots:
        if asynchronous:
            q, phi = self._ctx.snapshot_end("phi" in fields)
            other = dict(other, q=q)
            if phi is not None:
                other["phi"] = phi
        if not _writes(self):
            continue
        file_exist(fno)
        h5file = _open(fno)
        for field in fields:
            h5file.create_dataset(field, data=other[field])
        h5file.close()
    self._pending_snapshots = []


def flush_pending_quietly(self):
    """flush_snapshots for a ``finally`` clause: when an exception (say, the CFL assertion) is already propagating, a second
    failure in here -- the copy's completion, the file write -- is logged and dropped so that the first one is what the caller sees;
    on the normal path (nothing pending any more) it does nothing."""
    import sys
    if not getattr(self, "_pending_snapshots", None):
        return
    if sys.exc_info()[0] is None:
        return flush_snapshots(self)
    try:
        flush_snapshots(self)
    except Exception as e:                           # pragma: no cover - needs a failing writer during a failing run
        self._pending_snapshots = []
        try:
            self.logger.error("snapshot flush failed while another error was propagating: %r", e)
        except Exception:
            pass


def save_diagnostics(self):
    """ref: niwqg/Saving.py:88-101"""
    if not _writes(self):
        return
    fno = self.fno + '/diagnostics.h5'
    file_exist(fno, overwrite=self.overwrite)
    h5file = _open(fno)
    for key in self.diagnostics.keys():
        h5file.create_dataset(key, data=(self.diagnostics[key]['value']))
    h5file.close()
