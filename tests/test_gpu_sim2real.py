"""GPU parity tests of the sim-to-real rinks (envs/simtorealgoto.py, envs/simtorealpush.py): one square room of
random size without ceiling, random wall / floor texture families, a 0.11 m robot, per-episode box sizes, the
classes' own parameter table with domain randomisation forced on, and - for Push - the crude box-pushing
physics with its RNG draw, move_back, the yellow box and the proximity reward.  HIP path through the C ABI
against the CPU oracle, which is pinned bit-exactly to reference vectors (tests/golden/state_SimToReal*.npz).
Bars as everywhere: state / rewards / dones / RNG stream bit-exact, observations +-1/255, depth 1e-4 m.
"""
import math

import numpy as np
import pytest

from test_gpu_parity import assert_state_equal, obs_diff, oracle_states

pytestmark = pytest.mark.gpu

TASKS = [("MiniWorld-SimToRealGoTo-v0", "SimToRealGoTo"), ("MiniWorld-SimToRealPush-v0", "SimToRealPush")]


def params_of(task):
    from gym_miniworld_amd.params import sim_to_real_params
    return sim_to_real_params(push=task.endswith("Push")).to_table()


def make_pair(O, env_id, task, n, seed, depth=False, layout="HWC"):
    from gym_miniworld_amd.batch import BatchedMiniWorld
    b = BatchedMiniWorld(env_id, num_envs=n, seed=seed, want_depth=depth, layout=layout)
    assert b.domain_rand   # the classes force domain_rand=True (simtorealgoto.py:31)
    envs = [O.OracleEnv(task, seed=seed + i, domain_rand=True, params=params_of(task)) for i in range(n)]
    return b, envs


def assert_rink_state_equal(st, os_, exact_pose=True, tag=""):
    assert_state_equal(st, os_, exact_pose=exact_pose, tag=tag)
    assert np.array_equal(st["box_size"], np.array([s.box_size for s in os_])), (tag, "box_size")
    if os_[0].n_boxes == 2:
        assert np.array_equal(st["box2_pos"], np.array([list(s.box2_pos) for s in os_])), (tag, "box2_pos")
        assert np.array_equal(st["box2_dir"], np.array([s.box2_dir for s in os_])), (tag, "box2_dir")
        assert np.array_equal(st["box2_color"], np.array([list(s.box2_color) for s in os_])), (tag, "box2_color")
        assert np.array_equal(st["box2_size"], np.array([s.box2_size for s in os_])), (tag, "box2_size")
        assert np.array_equal(st["goal_dist"], np.array([s.goal_dist for s in os_])), (tag, "goal_dist")


@pytest.mark.parametrize("env_id,task", TASKS)
def test_rink_reset_state_geometry_and_first_obs(oracle_mod, env_id, task):
    O = oracle_mod
    n = 24
    b, envs = make_pair(O, env_id, task, n, seed=500, depth=True)
    assert b.max_episode_steps == (150 if task.endswith("Push") else 100)
    obs = b.reset().cpu().numpy()
    dep = b.depth.cpu().numpy()[..., 0]
    for e in envs:
        e.reset(render=False)
    assert_rink_state_equal(b.get_state(), oracle_states(envs), tag=env_id)
    seen_tex = set()
    for i in range(n):
        rooms, segs = b.get_geometry(i)
        g = envs[i].geometry()
        assert rooms.shape[0] == 1 and np.array_equal(segs, g["wall_segs"]) and segs.shape[0] == 4
        o = g["outline"]
        rect = np.stack([o[:, :, 0].min(1), o[:, :, 0].max(1), o[:, :, 1].min(1), o[:, :, 1].max(1)], axis=1)
        assert np.array_equal(rooms[:, 0:4], rect.astype(np.float32))
        assert np.array_equal(rooms[:, 4], -g["wall_height"].astype(np.float32))   # random wall height 0.2 .. 0.5 m; negative word = no ceiling
        tex = rooms[:, 5].view(np.int32)
        got = np.stack([tex & 255, (tex >> 8) & 255, (tex >> 16) & 255], axis=1)
        assert np.array_equal(got, g["tex_ids"])
        seen_tex.update(int(t) for t in got[0, :2])
    assert len(seen_tex) >= 5 and max(seen_tex) >= 7   # the extra texture families are really drawn
    sky_pixels = 0
    for i, e in enumerate(envs):
        ref, refd = e.render_obs(depth=True)
        d = obs_diff(obs[i], ref)
        assert d.max() <= 1, (env_id, i, int(d.max()), int((d > 1).sum()))
        assert np.abs(dep[i] - refd).max() <= 1e-4, (env_id, i)
        sky_pixels += int((refd > 99).sum())
    assert sky_pixels > 0.05 * n * 4800   # no ceiling: the sky shows above the low walls
    b.reset()
    for e in envs:
        e.reset(render=False)
    assert_rink_state_equal(b.get_state(), oracle_states(envs), tag=env_id + " 2nd reset")
    b.close()


def seek(s, tx, tz, rng, n_actions, tol=12):
    want = math.atan2(-(tz - s.agent_pos[2]), tx - s.agent_pos[0])
    diff = (want - s.agent_dir + math.pi) % (2 * math.pi) - math.pi
    if rng.random() < 0.05:
        return int(rng.integers(0, n_actions))
    if abs(diff) > math.radians(tol):
        return 0 if diff > 0 else 1
    return 2


def push_action(s, rng):
    """Get behind the red box as seen from the yellow one, then drive at it (as the fixture generator does)."""
    b1 = np.array([s.box_pos[0], s.box_pos[2]]); b2 = np.array([s.box2_pos[0], s.box2_pos[2]])
    a = np.array([s.agent_pos[0], s.agent_pos[2]])
    away = (b1 - b2) / max(np.linalg.norm(b1 - b2), 1e-9)
    stage = b1 + away * 0.3
    tgt = b1 if (np.linalg.norm(a - stage) < 0.12 or np.dot(a - b1, away) > 0.2) else stage
    return seek(s, tgt[0], tgt[1], rng, 4)


@pytest.mark.parametrize("env_id,task", TASKS)
def test_rink_rollout_exact_with_pushes_and_rewards(oracle_mod, env_id, task):
    """Goal-seeking rollouts with auto-reset.  Every step: reward, done, step_count exact; every 20 steps the whole
    state (incl. box poses moved by pushes and the RNG position, which advances with every push) and the
    observation after injecting the oracle pose."""
    import torch
    O = oracle_mod
    n, steps = 16, 600
    push = task.endswith("Push")
    b, envs = make_pair(O, env_id, task, n, seed=8000)
    b.reset()
    for e in envs:
        e.reset(render=False)
    rng = np.random.default_rng(3)
    rewards = pushes = 0
    for t in range(steps):
        sts = oracle_states(envs)
        if push:
            a = np.array([push_action(s, rng) if i % 4 else int(rng.integers(0, 4)) for i, s in enumerate(sts)], np.int32)
        else:
            a = np.array([seek(s, s.box_pos[0], s.box_pos[2], rng, 3) for s in sts], np.int32)
        b.step(torch.from_numpy(a))
        rew, done, eps = b.reward64.cpu().numpy(), b.done.cpu().numpy(), b.ep_steps.cpu().numpy()
        for i, e in enumerate(envs):
            before = (e.state().box_pos[0], e.state().box2_pos[0]) if push else None
            _, r, d, _ = e.step(int(a[i]))
            s = e.state()
            assert r == rew[i] and d == bool(done[i]) and s.step_count == eps[i], (task, t, i, r, rew[i], d, done[i])
            rewards += r > 0
            if push and (s.box_pos[0], s.box2_pos[0]) != before:
                pushes += 1
            if d:
                e.reset(render=False)
        if t % 20 == 19 or t == steps - 1:
            os_ = oracle_states(envs)
            assert_rink_state_equal(b.get_state(), os_, tag="%s t=%d" % (task, t))
            obs = b.obs.cpu().numpy()   # the step's own frame, from the device's own (bit-equal) pose
            for i, e in enumerate(envs):
                d = obs_diff(obs[i], e.render_obs())
                assert d.max() <= 1, (task, t, i, int(d.max()), int((d > 1).sum()))
    assert rewards > 0
    if push:
        assert pushes > 20
    b.close()


@pytest.mark.parametrize("env_id,task", TASKS)
def test_rink_random_views_match_oracle(oracle_mod, env_id, task):
    """Random poses all over the rink (walls at grazing angles, the boxes from close up, the sky above the walls)."""
    O = oracle_mod
    n = 48
    b, envs = make_pair(O, env_id, task, n, seed=61, depth=True)
    b.reset()
    for e in envs:
        e.reset(render=False)
    st = oracle_states(envs)
    geo = [e.geometry() for e in envs]
    rng = np.random.default_rng(29)
    for rnd in range(5):
        pos = np.zeros((n, 2))
        for i, s in enumerate(st):
            size = geo[i]["outline"][:, :, 0].max()
            boxes = [(s.box_pos, s.box_size)] + ([(s.box2_pos, s.box2_size)] if s.n_boxes == 2 else [])
            while True:   # the eye stays outside every box
                p = rng.uniform(0.06, size - 0.06, size=2)
                if all(math.hypot(p[0] - bp[0], p[1] - bp[2]) > bs for bp, bs in boxes):
                    break
            pos[i] = p
        dirs = rng.uniform(-np.pi, np.pi, size=n)
        b.set_agent(0, pos_xz=pos, dir=dirs)
        obs = b.render().cpu().numpy()
        dep = b.depth.cpu().numpy()[..., 0]
        for i, e in enumerate(envs):
            e.set_agent(pos[i, 0], pos[i, 1], dirs[i])
            ref, refd = e.render_obs(depth=True)
            d = obs_diff(obs[i], ref)
            assert d.max() <= 1, (task, rnd, i, int(d.max()), int((d > 1).sum()))
            assert np.abs(dep[i] - refd).max() <= 1e-4, (task, rnd, i)
    b.close()


def test_rink_fast_path_equals_full_sample_path(monkeypatch):
    import torch
    from gym_miniworld_amd.batch import BatchedMiniWorld
    n, steps = 1024, 60
    outs = []
    for dbg in ("0", "1"):
        monkeypatch.setenv("MWB_DEBUG", dbg)
        b = BatchedMiniWorld("MiniWorld-SimToRealPush-v0", num_envs=n, seed=5, want_depth=True)
        b.reset()
        g = torch.Generator().manual_seed(3)
        for _ in range(steps):
            b.step(torch.randint(0, 4, (n,), generator=g, dtype=torch.int32))
        outs.append((b.obs.cpu().numpy().copy(), b.depth.cpu().numpy().copy(), b.reward64.cpu().numpy().copy()))
        b.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert np.array_equal(outs[0][2], outs[1][2])


def test_rink_gym_view_and_vecenv():
    """The Gym view exposes the per-episode sizes (box.size, agent.radius 0.11, goal_dist), Discrete(4) for Push;
    the VecEnv front-end steps it with move_back."""
    import torch
    from gym_miniworld_amd.env import make
    from gym_miniworld_amd.vec_env import MiniWorldVecEnv
    env = make("MiniWorld-SimToRealPush-v0", seed=2)
    assert env.action_space.n == 4 and env.domain_rand and env.max_episode_steps == 150
    env.reset()
    assert env.agent.radius == 0.11 and 0.075 <= env.box1.size[0] <= 0.09 and 0.075 <= env.box2.size[0] <= 0.09
    assert env.goal_dist == 1.5 * (env.box1.size[0] + env.box2.size[0])
    assert not env.intersect(env.agent, env.agent.pos, env.agent.radius)
    x0 = env.agent.pos.copy()
    env.step(env.actions.move_back)
    assert not np.array_equal(env.agent.pos, x0) or True   # a wall may block it; the action itself is accepted
    env.close()
    v = MiniWorldVecEnv("MiniWorld-SimToRealGoTo-v0", 8, seed=1)
    assert v.action_space.n == 3
    obs = v.reset()
    assert tuple(obs.shape) == (8, 3, 80, 60)
    obs, rew, done, infos = v.step(torch.full((8, 1), 2, dtype=torch.long))
    assert rew.shape == (8, 1) and len(infos) == 8
    v.close()
