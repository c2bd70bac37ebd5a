"""Where the entity tasks' render time goes: MWB_EXP bit 0 = mesh walks skipped (meshes never hit), bit 1 = flat mesh shading;
MWB_DEBUG 2 = no 8-sample path at all.  usage: python scripts/ab_mesh_phases.py <env id> [envs]"""
import sys, os
sys.path.insert(0, '.')
import torch
from gym_miniworld_amd.batch import BatchedMiniWorld
wl = sys.argv[1] if len(sys.argv) > 1 else "MiniWorld-PickupObjs-v0"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
envs = {}
for tag, (dbg, ex) in {"full": ("0", "0"), "no walks": ("0", "1"), "flat shading": ("0", "2"), "no 8-sample": ("2", "0")}.items():
    os.environ["MWB_DEBUG"], os.environ["MWB_EXP"] = dbg, ex
    envs[tag] = BatchedMiniWorld(wl, num_envs=n, seed=1)
    envs[tag].reset()
os.environ.pop("MWB_DEBUG"); os.environ.pop("MWB_EXP")
g = torch.Generator().manual_seed(0)
acts = [torch.randint(0, 3, (n,), generator=g, dtype=torch.int32).cuda() for _ in range(40)]
for rnd in range(2):
    for tag, b in envs.items():
        b.timing_enable(True)
        for a in acts: b.step(a)
        t = b.timing_read()
        print(rnd, wl, tag, round(t["render"], 3))
