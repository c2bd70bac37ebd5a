"""CPU, gloo, world_size 2 and 3: the multi-GPU path's only exchange - one round per step carrying the observation
shard and the aux pack (reward, done, ep_steps, feature, goal_pos), ring / direct / auto-tuned, consumed one round
behind as a pipelined learner does - the action scatter on the way back, and the env-index sharding, exactly as
bench.py drives them under RCCL."""
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
    from gym_miniworld_amd.distributed import ShardExchange, shard_range, unpack_aux
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    total = per_rank * world
    first, count = shard_range(total, rank, world)
    g = ShardExchange((count, 4, 5, 3), torch.uint8, "cpu", world, rank=rank, method=method)
    obs = torch.empty((count, 4, 5, 3), dtype=torch.uint8)
    aux = torch.empty((count, 8), dtype=torch.float64)
    ok = True

    def expect_obs(e, step):
        return (e * 7 + step * 13) % 251

    def expect_aux(e, step):   # reward, done, ep_steps, feature[2], goal_pos[3]
        return [0.25 * e - step, float((e + step) % 2), float(step + 3 * e), float(e % 2), float(step % 2), 10.0, 0.0, 6.0 - 12.0 * (e % 2)]

    def check(full, faux, step):
        good = True
        u = unpack_aux(faux)
        for e in range(total):
            good &= bool((full[e] == expect_obs(e, step)).all())
            good &= faux[e].tolist() == expect_aux(e, step)
            good &= bool(u["done"][e]) == bool((e + step) % 2) and int(u["ep_steps"][e]) == step + 3 * e
        return good

    learner_only = method == "learner" and rank != 0
    verified = True
    for step in range(6):
        # the learner (rank 0) decides actions for every env; each rank gets its slice back
        all_actions = torch.tensor([(e * 5 + step) % 3 for e in range(total)], dtype=torch.int64).unsqueeze(1) if rank == 0 else None
        mine = g.scatter_actions(all_actions, src=0)
        ok &= mine.dtype == torch.int32 and mine.tolist() == [((first + i) * 5 + step) % 3 for i in range(count)]
        # a fake "step": every value encodes (global env index, step); the buffers are reused every step exactly
        # like the library-owned outputs
        for i in range(count):
            obs[i] = expect_obs(first + i, step)
            aux[i] = torch.tensor(expect_aux(first + i, step), dtype=torch.float64)
        g.push(obs, aux=aux)
        obs.fill_(255); aux.fill_(-1)   # the library overwrites its outputs right away: push must have snapshotted them
        if learner_only:                # gather-to-learner: the other ranks keep their own shard only
            mo, ma = g.latest()
            ok &= mo.shape[0] == count and bool((mo[0] == expect_obs(first, step)).all()) and ma[0].tolist() == expect_aux(first, step)
        else:
            if step > 0:                     # a pipelined learner consumes the round before the newest one ...
                po, pa = g.previous()
                ok &= check(po, pa, step - 1)
            full, faux = g.latest()          # ... and the newest one is right too
            ok &= check(full, faux, step)
        if step in (1, 4):                   # the self-check bench.py runs in its warm-up: every rank's shard arrived
            good, seen = g.verify()
            verified &= good and seen == world
    ok &= verified
    ok &= g.measure(2) >= 0.0
    if method != "learner":   # a corrupted shard must be noticed
        g.push(obs, aux=aux)
        g.latest()      # waits for the round
        if rank == 0:   # one byte of the LAST rank's shard flipped in rank 0's copy of the batch
            g.gathered[g.cur ^ 1][(world - 1) * count] ^= 1
        good, _ = g.verify()
        ok &= not good
    g.drain()
    seeds = (1 + first + np.arange(count)).tolist()   # a function of the global env index only
    out_q.put((rank, ok, seeds, g.method))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,method", [(2, "ring"), (2, "direct"), (3, "direct"), (3, "ring"), (2, "auto"), (2, "learner"), (3, "learner")])
def test_step_exchange_action_scatter_and_sharding(world, method):
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
    assert len({m for _, _, _, m in got}) == 1 and got[0][3] in ("ring", "direct", "learner")   # every rank took the same decision
