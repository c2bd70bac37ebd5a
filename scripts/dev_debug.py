import sys, os, numpy as np
sys.path.insert(0, '.')
import torch
from gym_miniworld_amd.batch import BatchedMiniWorld
from oracle import oracle as O
n=8
b = BatchedMiniWorld("MiniWorld-Maze-v0", num_envs=n, seed=5, domain_rand=0, want_depth=True)
obs = b.reset().cpu().numpy(); dep=b.depth.cpu().numpy()[...,0]
bad=[]
for i in range(n):
    e = O.OracleEnv("Maze", seed=5+i); e.reset(render=False)
    r, dd = e.render_obs(depth=True)
    d=np.abs(obs[i].astype(int)-r.astype(int)).max(axis=2); bad.append(int((d>1).sum()))
print('bad', bad)
