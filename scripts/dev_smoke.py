import sys, time, numpy as np
sys.path.insert(0, '.')
import torch
from gym_miniworld_amd.batch import BatchedMiniWorld
from oracle import oracle as O
for env_id, task, args in [("MiniWorld-OneRoom-v0","OneRoom",None),("MiniWorld-Hallway-v0","Hallway",None),("MiniWorld-FourRooms-v0","FourRooms",None),("MiniWorld-Maze-v0","Maze",None)]:
  for dr in (0,1):
    n=8
    b = BatchedMiniWorld(env_id, num_envs=n, seed=5, domain_rand=dr, want_depth=True)
    obs = b.reset().cpu().numpy(); dep=b.depth.cpu().numpy()[...,0]
    st = b.get_state()
    md=0; mdd=0; bad=0
    for i in range(n):
        e = O.OracleEnv(task, seed=5+i, domain_rand=dr, task_args=args)
        e.reset(render=False); s=e.state()
        ok = (st['agent_pos'][i,0]==s.agent_pos[0] and st['agent_dir'][i]==s.agent_dir and st['box_pos'][i,2]==s.box_pos[2] and st['rng_pos'][i]==s.rng_pos and st['rng_keysum'][i]==s.rng_keysum and st['n_rooms'][i]==s.n_rooms and st['n_segs'][i]==s.n_segs)
        if not ok: bad+=1; print(' state mismatch', i, st['agent_pos'][i], list(s.agent_pos), st['rng_pos'][i], s.rng_pos, st['n_rooms'][i], s.n_rooms, st['n_segs'][i], s.n_segs)
        ref, refd = e.render_obs(depth=True)
        d = np.abs(obs[i].astype(int)-ref.astype(int)); md=max(md,d.max()); mdd=max(mdd, np.abs(dep[i]-refd).max())
        if d.max()>1: print('  env',i,'n>1:',(d.max(axis=2)>1).sum())
    print(env_id, 'dr',dr,'state bad',bad,'obs maxdiff',md,'depth maxdiff',mdd, 'mean', obs.mean())
    b.close()
# quick throughput
for env_id,n in [("MiniWorld-OneRoom-v0",4096),("MiniWorld-Maze-v0",8192)]:
    b = BatchedMiniWorld(env_id, num_envs=n, seed=1)
    b.reset(); torch.cuda.synchronize()
    a = torch.randint(0,3,(n,),dtype=torch.int32,device='cuda')
    for _ in range(5): b.step(a)
    torch.cuda.synchronize(); t=time.time()
    K=30
    for _ in range(K): b.step(a)
    torch.cuda.synchronize(); dt=time.time()-t
    print(env_id, n, 'steps/s %.0f'%(n*K/dt), 'ms/step %.3f'%(dt/K*1e3))
    b.timing_enable(True)
    for _ in range(10): b.step(a)
    print(b.timing_read())
    b.close()
