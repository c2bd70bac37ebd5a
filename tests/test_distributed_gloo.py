"""CPU, world_size 2, gloo: the multi-GPU path's only collective (all-gather of the observation
shard) and the env-index sharding, exactly as bench.py drives them under RCCL."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, out_q):
    sys.path.insert(0, ROOT)
    from gym_miniworld_amd.distributed import ObsGatherer, shard_range
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, count = shard_range(total, rank, world)
    g = ObsGatherer((count, 4, 5, 3), torch.uint8, "cpu", world)
    results = []
    obs = torch.empty((count, 4, 5, 3), dtype=torch.uint8)
    for step in range(5):
        # a fake "render": pixel value encodes (global env index, step); the buffer is reused
        # every step like the library-owned obs buffer is
        for i in range(count):
            obs[i] = (first + i) * 7 % 251 + step
        g.push(obs)
        if step >= 1:
            results.append(g.latest().clone())
    g.drain()
    last = g.latest().clone()
    ok = True
    for step, full in ((4, last),):
        for e in range(total):
            ok &= bool((full[e] == e * 7 % 251 + step).all())
    # seeds are a function of the global env index only -> independent of the sharding
    seeds = (1 + first + np.arange(count)).tolist()
    out_q.put((rank, ok, seeds, [r.shape[0] for r in results]))
    dist.barrier()
    dist.destroy_process_group()


def test_obs_all_gather_and_sharding_world2():
    world, total = 2, 6
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in got)
    assert got[0][2] + got[1][2] == list(range(1, total + 1))
    assert all(n == total for _, _, _, shapes in got for n in shapes)
