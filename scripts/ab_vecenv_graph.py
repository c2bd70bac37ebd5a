#!/usr/bin/env python3
"""VecEnv boundary with and without hipGraph replay of the step.  usage: ab_vecenv_graph.py [env_id] [n]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import make_actions  # noqa: E402
from gym_miniworld_amd.vec_env import MiniWorldVecEnv  # noqa: E402

env_id = sys.argv[1] if len(sys.argv) > 1 else "MiniWorld-Maze-v0"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
dev = torch.device("cuda", 0)
K, Wm = 300, 50
for rep in range(2):
    for graph in (True, False):
        v = MiniWorldVecEnv(env_id, n, seed=1, device=0, to_float=False, feature_info=True, graph=graph)
        acts = make_actions(K + Wm, 0, n, dev).to(torch.int64).unsqueeze(2)
        v.reset()
        for t in range(Wm):
            v.step(acts[t])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(Wm, Wm + K):
            v.step(acts[t])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("graph", graph, "%.3f M env-steps/s  %.4f ms/step" % (n * K / dt / 1e6, dt / K * 1e3), flush=True)
        v.close()
