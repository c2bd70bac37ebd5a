import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch, time
from gym_miniworld_amd.batch import BatchedMiniWorld
# uint8 + float32, sliding vs shifting equality incl. wraps and dones
for dt in ("uint8","float32"):
    a=BatchedMiniWorld("MiniWorld-OneRoomS6-v0", num_envs=64, seed=2, layout="CWH")
    b=BatchedMiniWorld("MiniWorld-OneRoomS6-v0", num_envs=64, seed=2, layout="CWH")
    a.reset(); b.reset()
    a.stack_enable(4, dt, sliding=True); b.stack_enable(4, dt, sliding=False)
    sa=a.stack_update(True); sb=b.stack_update(True)
    assert torch.equal(sa, sb)
    g=torch.Generator().manual_seed(0); dones=0
    for t in range(120):
        act=torch.randint(0,3,(64,),generator=g,dtype=torch.int32)
        a.step(act); b.step(act)
        sa=a.stack_update(); sb=b.stack_update()
        assert torch.equal(sa, sb), (dt,t)
        dones+=int(a.done.sum())
    print(dt,'ok dones',dones, sa.shape, sa.stride())
    a.close(); b.close()
# timing
for sliding in (True, False):
    e=BatchedMiniWorld("MiniWorld-Maze-v0", num_envs=8192, seed=1, layout="CWH")
    e.reset(); e.stack_enable(4,"float32",sliding=sliding); e.stack_update(True)
    torch.cuda.synchronize(); 
    ev0=torch.cuda.Event(enable_timing=True); ev1=torch.cuda.Event(enable_timing=True)
    ev0.record()
    for t in range(90): e.stack_update()
    ev1.record(); torch.cuda.synchronize()
    print('sliding',sliding,'stack pass %.4f ms avg' % (ev0.elapsed_time(ev1)/90))
    e.close()
