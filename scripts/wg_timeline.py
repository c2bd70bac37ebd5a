#!/usr/bin/env python3
"""Occupancy timeline of the bulk render launch from per-workgroup s_memrealtime stamps (MWB_DEBUG bit 4)."""
import ctypes
import os
import sys

import numpy as np

os.environ["MWB_DEBUG"] = str(int(os.environ.get("MWB_DEBUG", "0")) | 16)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from gym_miniworld_amd.batch import BatchedMiniWorld  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "maze8192"
env_id, n, depth, dr, _ = bench.WORKLOADS[wl]
env = BatchedMiniWorld(env_id, num_envs=n, seed=1, domain_rand=dr, want_depth=depth)
acts = bench.make_actions(60, 0, n, env.device)
env.reset()
for t in range(60):
    env.step(acts[t])
torch.cuda.synchronize()
buf = np.zeros((n + 4096) * 2, np.uint64)
k = env.L.mwb_debug_wg_times(env.h, buf.ctypes.data_as(ctypes.c_void_p), n + 4096)
ts = buf[:2 * k].reshape(k, 2).astype(np.int64)
ts = ts[ts[:, 1] > 0]
t0 = ts[:, 0].min()
start, end = (ts[:, 0] - t0) / 100.0, (ts[:, 1] - t0) / 100.0   # microseconds
dur = end - start
print("%s: %d workgroups, launch span %.1f us, workgroup duration mean %.1f us  p5 %.1f  p50 %.1f  p95 %.1f  max %.1f"
      % (wl, len(ts), end.max(), dur.mean(), *np.percentile(dur, [5, 50, 95]), dur.max()))
grid = np.linspace(0, end.max(), 41)
act = [(np.sum((start <= g) & (end > g))) for g in grid]
print("active workgroups over time (40 bins):", " ".join("%d" % a for a in act))
busy = np.sum(dur)
print("sum of workgroup time / (1280 slots x span) = %.3f" % (busy / (1280 * end.max())))
