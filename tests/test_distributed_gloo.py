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
    import torch
    from niwqg_amd.distributed import Group
    from niwqg_amd import slab

    class FakeRank(object):          # only the buffers: what the transports touch
        def __init__(self, P, rank, n):
            self.torch = torch
            c = torch.complex(torch.arange(n, dtype=torch.float64) + 10.0 * rank, -torch.arange(n, dtype=torch.float64))
            self.gx = [torch.arange(n, dtype=torch.float64) + 1000.0 * rank, torch.zeros(n, dtype=torch.complex128), None, None]
            self.gy = [torch.zeros(n, dtype=torch.float64), c, None, None]
            self.sums = torch.full((64,), float(rank + 1), dtype=torch.float64)

    g = Group(backend="gloo")
    P, n = g.world, 12
    mine = FakeRank(P, g.rank, n)
    tr = slab.TorchTransport(g.dist)
    tr.exchange([mine], 0, True)                  # x -> y through all_to_all_single
    tr.allreduce([mine], 0, 44)
    # the same thing with every rank in one process
    allr = [FakeRank(P, r, n) for r in range(P)]
    vt = slab.VirtualTransport()
    vt.exchange(allr, 0, True)
    vt.allreduce(allr, 0, 44)
    assert torch.equal(mine.gy[0], allr[g.rank].gy[0]), (mine.gy[0], allr[g.rank].gy[0])
    assert torch.equal(mine.sums, allr[g.rank].sums)
    assert float(mine.sums[0]) == P * (P + 1) / 2 and float(mine.sums[50]) == g.rank + 1
    tr.exchange([mine], 1, False)                 # complex128 group, y -> x
    vt.exchange(allr, 1, False)
    assert torch.equal(mine.gx[1], allr[g.rank].gx[1]) and mine.gx[1].abs().sum() > 0
    tr.exchange([mine], 2, False)                 # empty group: no-op
    if g.rank == 0:
        print("slab transports agree")
    g.close()
""" % ROOT)


def test_torch_transport_equals_virtual_transport(tmp_path):
    """The real transport (all_to_all_single / all_reduce) and the single-process stand-in used by the GPU
    tests move the same blocks: 2 gloo ranks on CPU tensors."""
    from conftest import free_port
    port = free_port()
    script = tmp_path / "slab_worker.py"
    script.write_text(SLAB_WORKER)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "slab transports agree" in out.stdout
