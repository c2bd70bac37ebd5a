#!/usr/bin/env python3
"""Per-step bulk-render time from a cold start (why short bench runs read lower): prints the render kernel's ms
for each of the first steps after reset.  usage: warmup_series.py [spin_ms_before_reset]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import make_actions  # noqa: E402
from gym_miniworld_amd.batch import BatchedMiniWorld  # noqa: E402

spin = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
dev = torch.device("cuda", 0)
env = BatchedMiniWorld("MiniWorld-Maze-v0", num_envs=8192, seed=1, device=0)
acts = make_actions(120, 0, 8192, dev)
if spin > 0:
    a = torch.randn(4096, 4096, device=dev)
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < spin:
        (a @ a).sum().item()
env.reset()
torch.cuda.synchronize()
env.timing_enable(True)
out = []
for t in range(120):
    env.step(acts[t])
    out.append(env.timing_read()["render"])
print("spin %.0f ms:" % spin, " ".join("%.3f" % v for v in out))
