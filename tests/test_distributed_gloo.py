"""CPU, gloo, world_size 2 and 3: the multi-GPU path's only exchange (all-gather of the observation
shard, ring / direct / auto-tuned) and the env-index sharding, exactly as bench.py drives them
under RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, per_rank, method, out_q):
    sys.path.insert(0, ROOT)
    from gym_miniworld_amd.distributed import ObsGatherer, shard_range
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    total = per_rank * world
    first, count = shard_range(total, rank, world)
    g = ObsGatherer((count, 4, 5, 3), torch.uint8, "cpu", world, rank=rank, method=method)
    obs = torch.empty((count, 4, 5, 3), dtype=torch.uint8)
    ok = True
    for step in range(6):
        # a fake "render": the pixel value encodes (global env index, step); the buffer is reused every
        # step exactly like the library-owned obs buffer
        for i in range(count):
            obs[i] = ((first + i) * 7 + step * 13) % 251
        g.push(obs)
        obs.fill_(255)   # the library overwrites obs right away: the gather must have snapshotted it
        full = g.latest()
        for e in range(total):
            ok &= bool((full[e] == (e * 7 + step * 13) % 251).all())
    g.drain()
    seeds = (1 + first + np.arange(count)).tolist()   # a function of the global env index only
    out_q.put((rank, ok, seeds, g.method))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,method", [(2, "ring"), (2, "direct"), (3, "direct"), (2, "auto")])
def test_obs_all_gather_and_sharding(world, method):
    per_rank = 3
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, per_rank, method, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in got)
    assert sum((s for _, _, s, _ in got), []) == list(range(1, per_rank * world + 1))
    assert len({m for _, _, _, m in got}) == 1 and got[0][3] in ("ring", "direct")   # every rank took the same decision
