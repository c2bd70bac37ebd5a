import sys, os
sys.path.insert(0, '.')
import torch
from gym_miniworld_amd.batch import BatchedMiniWorld
for wl in ("MiniWorld-Maze-v0", "MiniWorld-FourRooms-v0", "MiniWorld-OneRoom-v0"):
    n = 2048
    os.environ["MWB_DEBUG"] = str(16 | 2 | 4)   # paint classes, skip full path and interior shading
    b = BatchedMiniWorld(wl, num_envs=n, seed=1)
    b.reset()
    g = torch.Generator().manual_seed(0)
    tot = torch.zeros(5)
    for t in range(30):
        b.step(torch.randint(0, 3, (n,), generator=g, dtype=torch.int32))
        if t >= 10:
            b.obs.zero_(); b.render()
            o = b.obs
            fl = (o[..., 0] == 255).float().mean(); ce = (o[..., 1] == 255).float().mean(); wa = (o[..., 2] == 255).float().mean()
            # edge pixels of other kinds were painted (0,0,0) too, interior untouched (0 after zero_) -> count via second pass
            tot += torch.tensor([fl, ce, wa, 0, 1])
    print(wl, "edge-all-floor %.3f  edge-all-ceil %.3f  edge-all-wall(diff piece/room) %.3f" % tuple((tot[:3] / tot[4]).tolist()))
    b.close()
