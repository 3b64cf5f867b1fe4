"""world_size-2 (and 3) `gloo` runs of the row-sharded solve: one process per shard, the real
torch.distributed all-gather (DistExchange) between the phases, numpy shard engines standing in for the
GPUs.  Every rank must end with its slice of the whole-table oracle result and the same global stats."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, seed, kw, out_dir):
    for p in (ROOT, os.path.join(ROOT, "rio-rs_amd"), os.path.join(ROOT, "oracle"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import sharded
    from shard_engine_cpu import CpuShardEngine
    from test_sharded_protocol import random_case
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cur, load, aff, cap, alive = random_case(seed, **kw)
        b = sharded.shard_bounds(len(cur), world)
        lo, hi = b[rank], b[rank + 1]
        eng = CpuShardEngine(cur[lo:hi], load[lo:hi], aff[lo:hi], cap, alive)
        sol = sharded.ShardedSolver([eng], sharded.DistExchange(), spill_rounds=2)
        st1 = sol.tick()
        # second tick on the committed state with some nodes failing: the clean_server stream (config 5)
        alive2 = alive.copy()
        alive2[::3] = 0
        eng.alive = alive2.astype(bool)
        st2 = sol.tick()
        np.savez(os.path.join(out_dir, "r%d.npz" % rank), a1=eng.cur, a2=eng.assign, used=eng.used,
                 st1=np.array([st1[k] for k in sorted(st1)], np.uint64), st2=np.array([st2[k] for k in sorted(st2)], np.uint64))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,seed,kw", [
    (2, 21, dict(n=6000, m=24, cap_scale=0.9, dead_frac=0.1)),
    (2, 22, dict(n=4000, m=16, cap_scale=1.4, warm=0.0)),
    (3, 23, dict(n=5000, m=8, cap_scale=0.7, zero_load=0.15)),
])
def test_gloo_sharded_tick_equals_oracle(oracle, tmp_path, world, seed, kw):
    from test_sharded_protocol import random_case
    import sharded
    port = _free_port()
    mp.spawn(_worker, args=(world, port, seed, kw, str(tmp_path)), nprocs=world, join=True)
    cur, load, aff, cap, alive = random_case(seed, **kw)
    want1, used1, ost1 = oracle.tick(cur, load, aff, cap, alive, 2)
    alive2 = alive.copy()
    alive2[::3] = 0
    want2, used2, ost2 = oracle.tick(want1, load, aff, cap, alive2, 2)
    b = sharded.shard_bounds(len(cur), world)
    got2 = []
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "r%d.npz" % r))
        got2.append(z["a2"])
        assert np.array_equal(z["used"], used2)
        assert [int(v) for v in z["st1"]] == [ost1[k] for k in sorted(ost1)]
        assert [int(v) for v in z["st2"]] == [ost2[k] for k in sorted(ost2)]
    assert np.array_equal(np.concatenate(got2), want2)
    assert ost2["evicted"] > 0
