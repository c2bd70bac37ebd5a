"""A/B timing of the render kernel's phases in one process via MWB_DEBUG (read at mwb_create):
1 = every pixel through the 8-sample path, 2 = skip the 8-sample path, 4 = skip interior shading,
6 = corner passes + prologue only; 70 = 6 without the per-pixel corner passes (prologue, item pre-tests, copy-out);
134 = prologue + copy-out only; bits 8+ = units of 128 B of LDS padding (occupancy experiments)."""
import sys, os, time
sys.path.insert(0, '.')
import torch
from gym_miniworld_amd.batch import BatchedMiniWorld
wl = sys.argv[1] if len(sys.argv) > 1 else "MiniWorld-Maze-v0"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
envs = {}
for dbg in ("0", "2", "4", "6", "70", "134"):
    os.environ["MWB_DEBUG"] = dbg
    envs[dbg] = BatchedMiniWorld(wl, num_envs=n, seed=1)
    envs[dbg].reset()
os.environ.pop("MWB_DEBUG")
g = torch.Generator().manual_seed(0)
acts = [torch.randint(0, 3, (n,), generator=g, dtype=torch.int32).cuda() for _ in range(60)]
for rnd in range(3):
    for dbg, b in envs.items():
        b.timing_enable(True)
        for a in acts: b.step(a)
        t = b.timing_read()
        print(rnd, "dbg", dbg, {k: round(v, 4) for k, v in t.items() if k != 'n'})
