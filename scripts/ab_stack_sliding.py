"""Frame stack three ways: fused into the render kernels, sliding window with its own pass, shifting stack.  Equality of the
values (incl. window wraps and episode ends), then the time of a whole step.  usage: ab_stack_sliding.py"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, time
from gym_miniworld_amd.batch import BatchedMiniWorld
# uint8 + float32, sliding vs shifting equality incl. wraps and dones
for dt in ("uint8","float32"):
    a=BatchedMiniWorld("MiniWorld-OneRoomS6-v0", num_envs=64, seed=2, layout="CWH")
    b=BatchedMiniWorld("MiniWorld-OneRoomS6-v0", num_envs=64, seed=2, layout="CWH")
    a.reset(); b.reset()
    a.stack_enable(4, dt, fused=True); b.stack_enable(4, dt, sliding=False)
    a.reset(); b.reset()   # the fused stack is written by reset / step themselves
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
# timing: a whole step incl. the stack, three ways
from bench import make_actions
for name, kw in (("fused", dict(fused=True)), ("sliding", dict(sliding=True)), ("shifting", dict(sliding=False))):
    e=BatchedMiniWorld("MiniWorld-Maze-v0", num_envs=8192, seed=1, layout="CWH")
    e.stack_enable(4,"float32",**kw); e.reset(); e.stack_update(True)
    acts=make_actions(140, 0, 8192, e.device)
    for t in range(40): e.step(acts[t]); e.stack_update()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for t in range(40,140): e.step(acts[t]); e.stack_update()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/100
    print(name,'step + stack %.4f ms  %.2f M env-steps/s' % (dt*1e3, 8192/dt/1e6))
    e.close()
