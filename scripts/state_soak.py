#!/usr/bin/env python3
"""Randomised state-parity soak (not part of the test suite): long random rollouts with auto-reset, many envs
per task; reward (f64), done, step count AND THE AGENT POSE (position and heading, bit for bit) compared with the
CPU oracle at EVERY step, the RNG position / key checksum, box poses and counters every 100 steps.
usage: state_soak.py [envs] [steps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from gym_miniworld_amd.batch import BatchedMiniWorld, ENV_SPECS  # noqa: E402
from oracle import oracle as O  # noqa: E402

TASKS = [("MiniWorld-Hallway-v0", "Hallway"), ("MiniWorld-OneRoom-v0", "OneRoom"), ("MiniWorld-FourRooms-v0", "FourRooms"),
         ("MiniWorld-MazeS3-v0", "Maze"), ("MiniWorld-Maze-v0", "Maze"), ("MiniWorld-TMaze-v0", "TMaze"),
         ("MiniWorld-TMazeDynamic-v0", "TMaze"), ("MiniWorld-TMazeTwoBoxDynamicFeatures100K-v0", "TMazeTwoBox"),
         ("MiniWorld-SimToRealGoTo-v0", "SimToRealGoTo"), ("MiniWorld-SimToRealPush-v0", "SimToRealPush"), ("MiniWorld-PutNext-v0", "PutNext"), ("MiniWorld-YMaze-v0", "YMaze"), ("MiniWorld-YMazeLeft-v0", "YMaze")]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
for env_id, task in TASKS:
    spec = ENV_SPECS[env_id]
    params = spec[3]().to_table() if spec[3] else None
    for dr in ((1,) if task.startswith("SimToReal") else (0, 1)):
        t0 = time.time()
        b = BatchedMiniWorld(env_id, num_envs=n, seed=7000, domain_rand=bool(dr))
        envs = [O.OracleEnv(task, seed=7000 + i, domain_rand=bool(dr), task_args=spec[1] or None, params=params,
                            max_episode_steps=spec[2], textures=True) for i in range(n)]
        b.reset()
        for e in envs:
            e.reset(render=False)
        rng = np.random.default_rng(1)
        na = b.n_actions
        episodes = rewards = mism = pose_mism = 0
        pose_ulp = 0.0
        for t in range(steps):
            a = rng.choice(na, size=n, p=[0.2, 0.2, 0.6] if na == 3 else [0.2, 0.2, 0.5, 0.1] if na == 4 else
                           [0.15, 0.15, 0.4, 0.05, 0.15, 0.06, 0.02, 0.02]).astype(np.int32)
            b.step(torch.from_numpy(a))
            rew, done, eps = b.reward64.cpu().numpy(), b.done.cpu().numpy(), b.ep_steps.cpu().numpy()
            for i, e in enumerate(envs):
                _, r, d, _ = e.step(int(a[i]))
                if r != rew[i] or d != bool(done[i]) or e.state().step_count != eps[i]:
                    mism += 1
                rewards += r != 0
                if d:
                    episodes += 1
                    e.reset(render=False)
            st = b.get_state()
            os_ = [e.state() for e in envs]
            op = np.array([list(s.agent_pos) for s in os_]); od = np.array([s.agent_dir for s in os_])
            bad = (st["agent_pos"] != op).any(axis=1) | (st["agent_dir"] != od)
            if bad.any():
                pose_mism += int(bad.sum())
                pose_ulp = max(pose_ulp, float(np.abs(st["agent_pos"] - op).max() / np.spacing(np.abs(op).max())))
            if t % 100 == 99:
                st = b.get_state()
                os_ = [e.state() for e in envs]
                for k, f in (("rng_pos", lambda s: s.rng_pos), ("rng_keysum", lambda s: s.rng_keysum), ("box_dir", lambda s: s.box_dir),
                             ("goal_idx", lambda s: s.goal_idx), ("task_step_count", lambda s: s.task_step_count)):
                    if not np.array_equal(st[k].astype(np.int64) if st[k].dtype.kind in "iu" else st[k], np.array([f(s) for s in os_])):
                        mism += 1
                if not np.array_equal(st["box_pos"], np.array([list(s.box_pos) for s in os_])):
                    mism += 1
                nb = b.n_boxes   # every box incl. the carry height, and who carries what (PutNext)
                if not np.array_equal(st["boxes_pos"], np.array([np.array(s.boxes_pos)[:nb] for s in os_])) or \
                        not np.array_equal(st["carrying"], np.array([s.carrying for s in os_])):
                    mism += 1
        b.close()
        print("%-46s dr%d  %d envs x %d steps  episodes %6d  nonzero rewards %6d  MISMATCHES %d  POSE MISMATCHES %d (max %.1f ulp)  (%.0f s)"
              % (env_id, dr, n, steps, episodes, rewards, mism, pose_mism, pose_ulp, time.time() - t0), flush=True)
