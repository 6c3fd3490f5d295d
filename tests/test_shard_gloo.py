"""N > 1 path on CPU: two gloo ranks shard a synthetic cohort, 'call' their samples and gather the records."""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
import numpy as np
import torch.distributed as dist
sys.path.insert(0, os.environ["SP_ROOT"])
import __graft_entry__ as ge
pkg = ge.load_package()
from pb_starphase_amd import shard
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
n_samples, genes = 7, 3
cost = [(s * 37) % 11 + 1 for s in range(n_samples)]
mine = shard.partition(n_samples, world, rank, cost)
calls = np.zeros(len(mine) * genes, shard.CALL_DTYPE)
for i, s in enumerate(mine):
    for g in range(genes):
        calls[i * genes + g] = (s, g, 100 * s + g, 100 * s + g + 1)      # stand-in for the per-gene GPU calls
table = shard.gather_calls(calls)
# equal record counts on every rank: the variant without the count exchange gives the same table
even = np.zeros(genes, shard.CALL_DTYPE)
for g in range(genes):
    even[g] = (rank, g, 7 * rank + g, 7 * rank + g + 1)
t1, t2 = shard.gather_calls(even), shard.gather_calls(even, same_count=True)
assert t1.tolist() == t2.tolist() and len(t2) == world * genes
print(json.dumps({"rank": rank, "mine": mine, "table": table.tolist()}))
dist.destroy_process_group()
'''


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_partition_properties():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    ge.load_package()
    from pb_starphase_amd import shard
    for n in (0, 1, 5, 8, 256):
        for world in (1, 2, 4, 8):
            parts = [shard.partition(n, world, r) for r in range(world)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(map(len, parts)) - min(map(len, parts)) <= 1
    cost = [5, 1, 1, 1, 1, 1, 4, 4]
    parts = [shard.partition(8, 2, r, cost) for r in range(2)]
    assert sorted(sum(parts, [])) == list(range(8))
    loads = [sum(cost[u] for u in p) for p in parts]
    assert abs(loads[0] - loads[1]) <= 1
    assert len(shard.gather_calls(np.zeros(0, shard.CALL_DTYPE))) == 0


def test_two_rank_gloo_gather(tmp_path):
    import json
    port = free_port()
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SP_ROOT=ROOT)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        out, err = p.communicate(timeout=180)
        assert p.returncode == 0, err[-2000:]
        outs.append(json.loads(out.strip().splitlines()[-1]))
    owned = sorted(outs[0]["mine"] + outs[1]["mine"])
    assert owned == list(range(7)) and not set(outs[0]["mine"]) & set(outs[1]["mine"])
    expected = [[s, g, 100 * s + g, 100 * s + g + 1] for s in range(7) for g in range(3)]
    for o in outs:
        assert [list(r) for r in o["table"]] == expected            # every rank holds the full, sorted table


def test_solo_group_and_the_stream_tickets():
    """a rank on its own inside a larger job gathers only its own records (bench.py's `one_gpu_same_cohort` block: no collective while the other ranks wait at a barrier);
    the lanes of one locus take every sample of the stream exactly once (bench.py's Tickets)"""
    sys.path.insert(0, ROOT)
    import threading
    import __graft_entry__ as ge
    ge.load_package()
    from pb_starphase_amd import shard
    import bench
    calls = np.zeros(5, shard.CALL_DTYPE)
    calls["sample"] = [4, 1, 3, 1, 0]; calls["gene"] = [0, 1, 0, 0, 2]
    out = shard.gather_calls(calls, group=shard.SoloGroup())
    assert out["sample"].tolist() == [0, 1, 1, 3, 4] and out["gene"].tolist() == [2, 0, 1, 0, 0]
    assert shard.SoloGroup().gather(np.arange(3, dtype=np.int64)).tolist() == [[0, 1, 2]]
    t = bench.Tickets(1000)
    taken = [[] for _ in range(4)]

    def lane(k):
        while True:
            i = t.take()
            if i is None:
                return
            taken[k].append(i)
    threads = [threading.Thread(target=lane, args=(k,)) for k in range(4)]
    for x in threads:
        x.start()
    for x in threads:
        x.join()
    assert sorted(sum(taken, [])) == list(range(1000)) and t.take() is None
