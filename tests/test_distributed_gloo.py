"""world_size-2 gloo run (CPU) of the multi-process plumbing used by bench.py --gpus N."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import sys, json
    sys.path.insert(0, %r)
    from niwqg_amd.distributed import Group, shard_members, aggregate_throughput
    g = Group(backend="gloo")
    assert g.world == 2
    mine = shard_members(7, g.rank, g.world)
    counts = g.sum([len(mine), sum(mine)])
    assert counts == [7.0, 21.0], counts
    g.barrier()
    # rank 1 is slower: whole-job rate = (10 + 10 steps) / 2.0 s
    rate, slowest = aggregate_throughput(g, 10, 1.0 + g.rank)
    assert abs(rate - 10.0) < 1e-12 and slowest == 2.0, (rate, slowest)
    assert g.max(float(g.rank)) == 1.0
    if g.rank == 0:
        print(json.dumps({"ok": True, "members_rank0": mine}))
    g.close()
""" % ROOT)


def test_two_rank_gloo_plumbing(tmp_path):
    from conftest import free_port
    port = free_port()
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert '"ok": true' in out.stdout
    assert '"members_rank0": [0, 1, 2, 3]' in out.stdout


def test_member_sharding_covers_everything_once():
    from niwqg_amd.distributed import shard_members
    for n, w in ((64, 8), (7, 2), (3, 8), (64, 6)):
        got = sum((shard_members(n, r, w) for r in range(w)), [])
        assert got == list(range(n))
        sizes = [len(shard_members(n, r, w)) for r in range(w)]
        assert max(sizes) - min(sizes) <= 1


SLAB_WORKER = textwrap.dedent("""
    import sys
    sys.path.insert(0, %r)
    import numpy as np
    import torch
    from niwqg_amd.distributed import Group
    from niwqg_amd import slab

    g = Group(backend="gloo")
    P, n = g.world, 12

    def buf(rank, cplx):
        a = torch.arange(n, dtype=torch.float64) + 1000.0 * rank
        return torch.complex(a, -a - 1.0) if cplx else a

    for cplx in (False, True):                    # float64 group and complex128 group (moved as (re, im) rows)
        for host in (False, True):                # direct and staged through a host copy
            send, recv = buf(g.rank, cplx), torch.zeros(n, dtype=torch.complex128 if cplx else torch.float64)
            slab.all_to_all_blocks(g.dist, torch, send, recv, host)
            want = slab.reference_all_to_all([buf(r, cplx).numpy() for r in range(P)])[g.rank]
            assert np.array_equal(recv.numpy(), want), (recv, want)
    # the block algebra of the library's exchange: block d of rank s lands as block s of rank d
    sends = [np.arange(P * 3) + 100 * r for r in range(P)]
    got = slab.reference_all_to_all(sends)
    for d in range(P):
        for s_ in range(P):
            assert np.array_equal(got[d].reshape(P, -1)[s_], sends[s_].reshape(P, -1)[d])
    if g.rank == 0:
        print("slab transports agree")
    g.close()
""" % ROOT)


def test_torch_transport_equals_virtual_transport(tmp_path):
    """What the callback link hands to torch.distributed (all_to_all_single on the blocked group buffers, complex groups
    as (re, im) rows, optionally through a host copy) moves exactly the blocks of the reference permutation, which is
    also what the library's own links implement (block d of rank s -> block s of rank d): 2 gloo ranks on CPU tensors."""
    from conftest import free_port
    port = free_port()
    script = tmp_path / "slab_worker.py"
    script.write_text(SLAB_WORKER)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "slab transports agree" in out.stdout


PROBE = textwrap.dedent("""
    import sys, ctypes
    sys.path.insert(0, %r)
    from niwqg_amd import _lib
    L = _lib.lib()
    rc = L.nq_comm_probe()
    msg = L.nq_last_error(None).decode()
    buf = (ctypes.c_ubyte * 128)()
    rc2 = L.nq_comm_unique_id(buf)
    msg2 = L.nq_last_error(None).decode()
    print("RC", rc, rc2)
    print("MSG", msg, "|", msg2)
""" % ROOT)


def test_missing_librccl_is_an_error_code_not_a_crash(tmp_path):
    """NIWQG_AMD_RCCL_LIB pointing nowhere: nq_comm_probe and nq_comm_unique_id return -6 with dlopen's message (the error
    text used to be built from two dlerror() calls, the second of which returns NULL: undefined behaviour inside an
    extern "C" entry point, and the RuntimeError that slab.connect falls back on was never raised)."""
    script = tmp_path / "probe.py"
    script.write_text(PROBE)
    env = dict(os.environ, NIWQG_AMD_RCCL_LIB="/nonexistent/librccl.so")
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "RC -6 -6" in out.stdout, out.stdout
    line = [x for x in out.stdout.splitlines() if x.startswith("MSG")][0]
    assert "NIWQG_AMD_RCCL_LIB" in line and "/nonexistent/librccl.so" in line, line


AGREE_WORKER = textwrap.dedent("""
    import os, sys, types
    sys.path.insert(0, %r)
    rank = int(os.environ["RANK"])
    if rank == 0:
        os.environ["NIWQG_AMD_RCCL_LIB"] = "/nonexistent/librccl.so"      # the load fails on rank 0 only
    else:
        os.environ["NIWQG_AMD_RCCL_LIB"] = os.environ["MOCK_RCCL"]         # and succeeds on rank 1
    from niwqg_amd import _lib, slab
    from niwqg_amd.distributed import Group
    g = Group(backend="gloo")
    lead = types.SimpleNamespace(rank=g.rank, device=0)
    uid = slab.agree_rccl_id(_lib.lib(), lead, g.dist)
    assert uid is None, uid                  # agreed by both ranks: nobody goes on to ncclCommInitRank alone
    g.barrier()                              # and the process group is still in step
    if g.rank == 0:
        print("agreed: no rccl link")
    g.close()
""" % ROOT)


def test_rccl_setup_failure_on_one_rank_is_agreed_before_any_collective_setup(tmp_path):
    """slab.agree_rccl_id on two gloo ranks with the library load forced to fail on rank 0: both ranks learn it from the
    MIN-reduced probe, rank 0 never skips a broadcast that rank 1 waits in, and both return None (-> callback link)."""
    from conftest import free_port
    mock = os.path.join(ROOT, "tests", "mock_rccl", "libmock_rccl.so")
    if not os.path.exists(mock):
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", mock,
                        os.path.join(ROOT, "tests", "mock_rccl", "mock_rccl.cpp"), "-lpthread"], check=False)
    if not os.path.exists(mock):
        import pytest
        pytest.skip("tests/mock_rccl/libmock_rccl.so is not built (__graft_entry__.build())")
    port = free_port()
    script = tmp_path / "agree_worker.py"
    script.write_text(AGREE_WORKER)
    env = dict(os.environ, MOCK_RCCL=mock)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "agreed: no rccl link" in out.stdout
