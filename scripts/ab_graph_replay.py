#!/usr/bin/env python3
"""Does replaying mwb_step as one hipGraph shorten the dependent-launch gaps of the C-ABI loop?  Same env, same actions,
eager launches vs graph replay (+ the copy of the step's actions into the captured buffer).  usage: ab_graph_replay.py [workload]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import WORKLOADS, make_actions  # noqa: E402
from gym_miniworld_amd.batch import BatchedMiniWorld  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "maze8192"
env_id, n, depth, dr, _ = WORKLOADS[wl]
dev = torch.device("cuda", 0)
K, Wm = 300, 50
for mode in ("eager", "graph", "eager", "graph"):
    env = BatchedMiniWorld(env_id, num_envs=n, seed=1, domain_rand=dr, want_depth=depth, device=0)
    acts = make_actions(K + Wm, 0, n, dev)
    env.reset()
    a_static = acts[0].clone()
    g = None
    for t in range(Wm):
        env.step(acts[t])
    if mode == "graph":
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            env.step(a_static)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(Wm, Wm + K):
        if g is None:
            env.step(acts[t])
        else:
            a_static.copy_(acts[t], non_blocking=True)
            g.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(mode, "%.3f M env-steps/s  %.4f ms/step" % (n * K / dt / 1e6, dt / K * 1e3), flush=True)
    env.close()
