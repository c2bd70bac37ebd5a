#!/usr/bin/env python3
"""Where the VecEnv boundary's host time goes: per-step wall time of step_async (enqueue), the wait, and the post-processing
of step_wait, beside the device time of the same steps.  usage: vecenv_host_profile.py [env_id] [n]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import make_actions  # noqa: E402
from gym_miniworld_amd.vec_env import MiniWorldVecEnv  # noqa: E402

env_id = sys.argv[1] if len(sys.argv) > 1 else "MiniWorld-OneRoom-v0"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
v = MiniWorldVecEnv(env_id, n, seed=1, device=0, to_float=False, feature_info=True)
acts = make_actions(400, 0, n, torch.device("cuda", 0)).to(torch.int64).unsqueeze(2)
v.reset()
for t in range(50):
    v.step(acts[t])
T = {"async": 0.0, "wait": 0.0, "post": 0.0}
state = {"w": 0.0}
torch.cuda.synchronize()
t_all = time.perf_counter()
for t in range(50, 350):
    t0 = time.perf_counter()
    v.step_async(acts[t])
    t1 = time.perf_counter()
    v.step_wait()
    t2 = time.perf_counter()
    T["async"] += t1 - t0
    T["post"] += t2 - t1
torch.cuda.synchronize()
tot = (time.perf_counter() - t_all) / 300
print("%s x %d: %.1f us per step = enqueue %.1f + wait for the device and post-processing %.1f" %
      (env_id, n, tot * 1e6, T["async"] / 300 * 1e6, T["post"] / 300 * 1e6))
