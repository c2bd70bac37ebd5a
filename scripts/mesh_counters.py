"""Counts behind the entity tasks' render time (MWB_EXP=4: walk_meshes counters).  usage: python scripts/mesh_counters.py <env id> [envs]"""
import sys, os, ctypes
sys.path.insert(0, '.')
os.environ["MWB_EXP"] = "4"
import torch
from gym_miniworld_amd.batch import BatchedMiniWorld
wl = sys.argv[1] if len(sys.argv) > 1 else "MiniWorld-PickupObjs-v0"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
b = BatchedMiniWorld(wl, num_envs=n, seed=1)
b.reset()
g = torch.Generator().manual_seed(0)
for _ in range(30):
    b.step(torch.randint(0, 3, (n,), generator=g, dtype=torch.int32))
out = (ctypes.c_ulonglong * 8)()
b.L.mwb_debug_counters(b.h, out, 1)
b.render()
b.L.mwb_debug_counters(b.h, out, 1)
c = list(out)[:6]
print(wl, "per frame: sample rays %.0f  walks %.0f  node visits %.0f  tri tests %.0f  wave loop iters %.1f  wave calls %.1f" % tuple(x / n for x in c))
print("   visits per walk %.1f; iterations per wave call %.0f; lane utilisation of the walk loop %.3f" % (c[2] / max(c[1], 1), c[4] / max(c[5], 1), (c[2] + c[1]) / max(c[4] * 64, 1)))
