#!/usr/bin/env python3
"""How much do the per-step HIP timing events (mwb_timing_enable) cost?  Same loop as bench.py, on / off."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from gym_miniworld_amd.batch import BatchedMiniWorld  # noqa: E402

for wl in sys.argv[1:] or ["maze8192", "oneroom4096"]:
    env_id, n, depth, dr, _ = bench.WORKLOADS[wl]
    env = BatchedMiniWorld(env_id, num_envs=n, seed=1, domain_rand=dr, want_depth=depth)
    K, W = 300, 100
    acts = bench.make_actions(K + W, 0, n, env.device)
    env.reset()
    for t in range(W):
        env.step(acts[t])
    for mode in (0, 1, 0, 1):
        torch.cuda.synchronize()
        env.timing_enable(bool(mode))
        t0 = time.perf_counter()
        for t in range(W, W + K):
            env.step(acts[t])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if mode:
            env.timing_read()
        env.timing_enable(False)
        print(wl, "timing events %s: %.4f ms/step" % ("on " if mode else "off", dt / K * 1e3), flush=True)
    env.close()
