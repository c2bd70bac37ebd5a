"""A/B whole-step throughput: reset overlap on/off (MWB_NO_OVERLAP read at mwb_create)."""
import sys, os, time
sys.path.insert(0, '.')
import torch
from gym_miniworld_amd.batch import BatchedMiniWorld
wl, n = "MiniWorld-Maze-v0", 8192
envs = {}
for tag, no in (("overlap", "0"), ("serial", "1")):
    os.environ["MWB_NO_OVERLAP"] = no
    envs[tag] = BatchedMiniWorld(wl, num_envs=n, seed=1)
    envs[tag].reset()
g = torch.Generator().manual_seed(0)
acts = [torch.randint(0, 3, (n,), generator=g, dtype=torch.int32).cuda() for _ in range(200)]
for rnd in range(3):
    for tag, b in envs.items():
        torch.cuda.synchronize(); t = time.perf_counter()
        for a in acts: b.step(a)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        b.timing_enable(True)
        for a in acts[:50]: b.step(a)
        print(rnd, tag, "steps/s %.0f  ms/step %.4f" % (n * len(acts) / dt, dt / len(acts) * 1e3), {k: round(v, 4) for k, v in b.timing_read().items()})
        b.timing_enable(False)
assert torch.equal(envs["overlap"].obs, envs["serial"].obs)
print("obs equal")
