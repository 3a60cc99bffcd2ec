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
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29517", str(script)],
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
