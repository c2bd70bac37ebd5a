#!/usr/bin/env python3
"""Randomised parity soak (not part of the test suite): many worlds x random-walk poses per task, HIP path
vs the CPU oracle.  Prints, per task, the number of frames, the largest observation difference, how many
pixels exceed +-1 LSB and the largest depth difference.  usage: parity_soak.py [frames_per_task]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from gym_miniworld_amd.batch import BatchedMiniWorld, ENV_SPECS  # noqa: E402
from oracle import oracle as O  # noqa: E402

TASKS = [("MiniWorld-Hallway-v0", "Hallway"), ("MiniWorld-OneRoom-v0", "OneRoom"), ("MiniWorld-FourRooms-v0", "FourRooms"),
         ("MiniWorld-Maze-v0", "Maze"), ("MiniWorld-MazeS3-v0", "Maze"), ("MiniWorld-TMaze-v0", "TMaze"),
         ("MiniWorld-TMazeTwoBoxDynamicFeatures100K-v0", "TMazeTwoBox"), ("MiniWorld-SimToRealGoTo-v0", "SimToRealGoTo"),
         ("MiniWorld-SimToRealPush-v0", "SimToRealPush"), ("MiniWorld-PutNext-v0", "PutNext"), ("MiniWorld-YMaze-v0", "YMaze")]
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
n = 64
for env_id, task in TASKS:
    spec = ENV_SPECS[env_id]
    params = spec[3]().to_table() if spec[3] else None
    worst, over, dworst, count, t0 = 0, 0, 0.0, 0, time.time()
    for dr in ((1,) if task.startswith("SimToReal") else (0, 1)):
        b = BatchedMiniWorld(env_id, num_envs=n, seed=1000 + dr, domain_rand=bool(dr), want_depth=True)
        envs = [O.OracleEnv(task, seed=1000 + dr + i, domain_rand=bool(dr), task_args=spec[1] or None, params=params,
                            max_episode_steps=spec[2]) for i in range(n)]
        b.reset()
        for e in envs:
            e.reset(render=False)
        rng = np.random.default_rng(dr)
        rounds = max(1, frames // (n * (1 if task.startswith("SimToReal") else 2)))
        for r in range(rounds):
            # a short random walk on the oracle side (its poses are then injected, so both render the same f64 pose)
            for _ in range(6):
                for e in envs:
                    # (turn / move_back only in SimToRealPush: a forward move could push a box on the oracle side alone)
                    _, _, d, _ = e.step(int(rng.choice([0, 1, 3])) if task == "SimToRealPush" else int(rng.integers(0, 3)))
            st = [e.state() for e in envs]
            pos = np.array([[s.agent_pos[0], s.agent_pos[2]] for s in st])
            dirs = rng.uniform(-np.pi, np.pi, size=n)
            b.set_agent(0, pos_xz=pos, dir=dirs)
            obs = b.render().cpu().numpy()
            dep = b.depth.cpu().numpy()[..., 0]
            for i, e in enumerate(envs):
                e.set_agent(pos[i, 0], pos[i, 1], dirs[i])
                ref, refd = e.render_obs(depth=True)
                d = np.abs(obs[i].astype(int) - ref.astype(int))
                worst = max(worst, int(d.max())); over += int((d.max(axis=2) > 1).sum())
                dworst = max(dworst, float(np.abs(dep[i] - refd).max())); count += 1
        b.close()
    print("%-46s frames %5d  max |d obs| %d  pixels > 1 LSB %d  max |d depth| %.2e  (%.0f s)"
          % (env_id, count, worst, over, dworst, time.time() - t0), flush=True)
