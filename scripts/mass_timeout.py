#!/usr/bin/env python3
"""The step at which every env times out together (fixed-length episodes: Maze's 1536-step limit under a random policy)
regenerates the whole batch on the side chain.  Times that step beside an ordinary one.  usage: mass_timeout.py [env_id] [n]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import make_actions  # noqa: E402
from gym_miniworld_amd.batch import BatchedMiniWorld  # noqa: E402

env_id = sys.argv[1] if len(sys.argv) > 1 else "MiniWorld-Maze-v0"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
env = BatchedMiniWorld(env_id, num_envs=n, seed=1, max_episode_steps=30)
acts = make_actions(100, 0, n, env.device)
env.reset()
times = []
for t in range(95):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    env.step(acts[t])
    torch.cuda.synchronize()
    times.append(((time.perf_counter() - t0) * 1e3, int(env.done.sum())))
ordinary = [x for x, dn in times[5:] if dn < n // 100]
mass = [(x, dn) for x, dn in times if dn > n // 2]
print(env_id, "ordinary step %.3f ms (synchronous);  mass-timeout steps: %s" % (sum(ordinary) / len(ordinary), ["%.2f ms / %d envs" % m for m in mass]))
