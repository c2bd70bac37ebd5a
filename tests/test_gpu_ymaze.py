"""GPU parity tests of YMaze (envs/ymaze.py): rooms that are convex polygons with 3 or 4 arbitrary
edges - a corridor, a triangular hub, two arms rotated by -+120 degrees and the two back-face-culled connector rooms
connect_rooms builds where the arms overlap the hub - through the C ABI against the CPU oracle (itself bit-exact on
the reference's own YMaze / YMazeLeft / YMazeRight trajectories, tests/golden/state_YMaze*.npz, its polygon
renderer equal to the brute-force rendition of the reference's GL stream, glstream_YMaze_dr*.json, and pinned to
the reference's ymaze_0.jpg screenshot, refimg_ymaze.npz).

Bars as elsewhere: world (rotated outlines, edge normals, portals, segments), placement, RNG stream, rewards,
dones, info['goal_pos'] bit-exact; observations within +-1/255, depth within 1e-4 m; the interior-pixel fast path
equal to the 8-sample path bit for bit."""
import math

import numpy as np
import pytest

from test_gpu_parity import assert_state_equal, obs_diff, oracle_states

pytestmark = pytest.mark.gpu

IDS = [("MiniWorld-YMaze-v0", [0, 0, 0, 0]), ("MiniWorld-YMazeLeft-v0", [1, 3.9, -7.0, 0]), ("MiniWorld-YMazeRight-v0", [1, 3.9, 7.0, 0])]


def make_pair(O, env_id, args, n, seed, dr, depth=False, layout="HWC", **kw):
    from gym_miniworld_amd.batch import BatchedMiniWorld
    b = BatchedMiniWorld(env_id, num_envs=n, seed=seed, domain_rand=dr, want_depth=depth, layout=layout, **kw)
    envs = [O.OracleEnv("YMaze", seed=seed + i, domain_rand=dr, task_args=args) for i in range(n)]
    return b, envs


def assert_poly_table(rooms, g):
    """the polygon room table (include/miniworld_batch.h at mwb_get_geometry) against the oracle's float64 world"""
    o = g["outline"]
    assert rooms.shape == (6, 52) and o.shape[0] == 6
    flags = rooms[:, 2].view(np.int32)
    ne = flags & 255
    assert list(ne) == [4, 3, 4, 4, 4, 4]
    assert list((flags >> 8) & 1) == [0, 0, 0, 0, 1, 1]   # the two connectors run clockwise: nothing of them is drawn
    assert np.array_equal(rooms[:, 0], g["wall_height"].astype(np.float32))
    tex = rooms[:, 1].view(np.int32)
    assert np.array_equal(np.stack([tex & 255, (tex >> 8) & 255, (tex >> 16) & 255], axis=1), g["tex_ids"])
    for i in range(6):
        for k in range(int(ne[i])):
            ed = rooms[i, 4 + 12 * k:16 + 12 * k]
            p0, p1 = o[i, k], o[i, (k + 1) % ne[i]]
            assert np.array_equal(ed[0:2], p0.astype(np.float32)), (i, k)
            dvec = (p1 - p0) / math.sqrt((p1[0] - p0[0]) ** 2 + 0.0 + (p1[1] - p0[1]) ** 2)
            assert np.abs(ed[2:4] - dvec).max() < 1e-6 and abs(ed[4] - dvec[1]) < 1e-6 and abs(ed[5] + dvec[0]) < 1e-6, (i, k)
            nbr = int(ed[9:10].view(np.int32)[0])
            cnt = int(g["portal_count"][i, k])
            assert (nbr >= 0) == (cnt == 1), (i, k, nbr, cnt)
            if cnt:
                start, end, _, max_y = g["portals"][i, k, 0]
                assert np.array_equal(ed[6:9], np.array([start, end, max_y], np.float32)), (i, k)
    # portals into a culled connector lead on to the arm behind it
    hub = rooms[1]
    assert sorted(int(hub[4 + 12 * k + 9:4 + 12 * k + 10].view(np.int32)[0]) for k in range(3)) == [0, 2, 3]


@pytest.mark.parametrize("env_id,args", IDS)
@pytest.mark.parametrize("dr", [0, 1])
def test_ymaze_reset_state_geometry_and_first_obs(oracle_mod, env_id, args, dr):
    O = oracle_mod
    n = 16
    b, envs = make_pair(O, env_id, args, n, seed=900, dr=dr, depth=True)
    assert b.max_episode_steps == 280 and b.n_actions == 3 and b.n_boxes == 1 and b.has_goal_pos   # ymaze.py:21,26,92
    obs = b.reset().cpu().numpy()
    dep = b.depth.cpu().numpy()[..., 0]
    for e in envs:
        e.reset(render=False)
    assert_state_equal(b.get_state(), oracle_states(envs), exact_pose=True, tag=env_id)
    for i in (0, n - 1):
        rooms, segs = b.get_geometry(i)
        g = envs[i].geometry()
        assert np.array_equal(segs, g["wall_segs"]) and segs.shape[0] == 19
        assert_poly_table(rooms, g)
    for i, e in enumerate(envs):
        ref, refd = e.render_obs(depth=True)
        d = obs_diff(obs[i], ref)
        assert d.max() <= 1, (env_id, dr, i, int(d.max()), int((d > 1).sum()))
        assert np.abs(dep[i] - refd).max() <= 1e-4, (env_id, dr, i)
    b.reset()
    for e in envs:
        e.reset(render=False)
    assert_state_equal(b.get_state(), oracle_states(envs), exact_pose=True, tag=env_id + " 2nd reset")
    b.close()


def seek_action(s, rng):
    """through the corridor to the hub, then towards the box"""
    ax, az = s.agent_pos[0], s.agent_pos[2]
    bx, bz = s.box_pos[0], s.box_pos[2]
    tx, tz = (0.3, 0.0) if ax < -0.6 else (bx, bz)
    want = math.atan2(-(tz - az), tx - ax)
    diff = (want - s.agent_dir + math.pi) % (2 * math.pi) - math.pi
    if rng.random() < 0.05:
        return int(rng.integers(0, 3))
    if abs(diff) > math.radians(10):
        return 0 if diff > 0 else 1
    return 2


@pytest.mark.parametrize("env_id,args", IDS)
@pytest.mark.parametrize("dr", [0, 1])
def test_ymaze_rollout_exact(oracle_mod, env_id, args, dr):
    """box-seeking rollouts with auto-reset: reward, done, step count and info['goal_pos'] exact at every step, the full
    state every 25 steps, the step's own frame (rendered from the device's own state) every 25 steps"""
    import torch
    O = oracle_mod
    n, steps = 16, 600
    b, envs = make_pair(O, env_id, args, n, seed=5100, dr=dr, depth=True)
    b.reset()
    for e in envs:
        e.reset(render=False)
    rng = np.random.default_rng(3)
    rewards = n_done = in_arm = 0
    for t in range(steps):
        sts = oracle_states(envs)
        a = np.array([seek_action(s, rng) for s in sts], dtype=np.int32)
        b.step(torch.from_numpy(a))
        rew, done, eps, gpos = b.reward64.cpu().numpy(), b.done.cpu().numpy(), b.ep_steps.cpu().numpy(), b.goal_pos.cpu().numpy()
        for i, e in enumerate(envs):
            _, r, d, _ = e.step(int(a[i]))
            s = e.state()
            assert r == rew[i] and d == bool(done[i]) and s.step_count == eps[i], (env_id, dr, t, i, r, rew[i], d, done[i])
            assert list(s.box_pos) == list(gpos[i]), (t, i)
            rewards += r > 0
            in_arm += s.agent_pos[0] > 1.5
            if d:
                e.reset(render=False)
                n_done += 1
        if t % 25 == 24 or t == steps - 1:
            assert_state_equal(b.get_state(), oracle_states(envs), exact_pose=True, tag="%s dr%d t=%d" % (env_id, dr, t))
            obs, dep = b.obs.cpu().numpy(), b.depth.cpu().numpy()[..., 0]
            for i, e in enumerate(envs):
                ref, refd = e.render_obs(depth=True)
                assert obs_diff(obs[i], ref).max() <= 1 and np.abs(dep[i] - refd).max() <= 1e-4, (env_id, dr, t, i)
    assert rewards > 0 and n_done > 0 and in_arm > 50   # boxes were reached, episodes ended, the arms were walked
    b.close()


def random_poses(rng, n):
    """poses all over the Y: corridor, hub, both arms (arm axes at -+120 degrees), looking anywhere"""
    pos = np.zeros((n, 3))
    for i in range(n):
        where = rng.integers(0, 4)
        if where == 0:
            pos[i] = [rng.uniform(-8.6, -1.6), 0, rng.uniform(-1.5, 1.5)]
        elif where == 1:
            pos[i] = [rng.uniform(-0.9, 0.9), 0, rng.uniform(-0.5, 0.5)]
        else:
            sgn = 1.0 if where == 2 else -1.0
            along, across = rng.uniform(1.8, 8.6), rng.uniform(-1.5, 1.5)
            ang = math.radians(60.0)   # arm axis: (cos 60, -+sin 60) from the origin
            pos[i] = [along * math.cos(ang) - sgn * across * math.sin(ang), 0, sgn * (along * math.sin(ang) + 0.0) + across * math.cos(ang)]
    return pos, rng.uniform(-math.pi, math.pi, n)


@pytest.mark.parametrize("dr", [0, 1])
def test_ymaze_random_views_and_fast_path(oracle_mod, monkeypatch, dr):
    """random poses (injected through mwb_set_state) in every room of the Y: frames against the oracle, and the
    interior-pixel fast path against the 8-sample path bit for bit (obs and depth)"""
    O = oracle_mod
    n = 48
    fast, envs = make_pair(O, "MiniWorld-YMaze-v0", [0, 0, 0, 0], n, seed=31, dr=dr, depth=True)
    monkeypatch.setenv("MWB_DEBUG", "9")
    full, _ = make_pair(O, "MiniWorld-YMaze-v0", [0, 0, 0, 0], n, seed=31, dr=dr, depth=True)
    monkeypatch.delenv("MWB_DEBUG")
    fast.reset(); full.reset()
    for e in envs:
        e.reset(render=False)
    rng = np.random.default_rng(17)
    import torch
    worst = 0
    for rnd in range(6):
        pos, ang = random_poses(rng, n)
        for h in (fast, full):
            h.set_state(0, agent_pos=pos, agent_dir=ang)
        a, b_ = fast.render(), full.render()
        assert torch.equal(a, b_) and torch.equal(fast.depth, full.depth), (dr, rnd)
        obs, dep = a.cpu().numpy(), fast.depth.cpu().numpy()[..., 0]
        for i, e in enumerate(envs):
            e.set_agent(pos[i, 0], pos[i, 2], ang[i])
            ref, refd = e.render_obs(depth=True)
            d = obs_diff(obs[i], ref)
            worst = max(worst, int(d.max()))
            assert d.max() <= 1 and np.abs(dep[i] - refd).max() <= 1e-4, (dr, rnd, i, int(d.max()), int((d > 1).sum()), pos[i], ang[i])
    fast.close(); full.close()


def test_ymaze_gym_view_vecenv_and_cwh(oracle_mod):
    import torch
    from gym_miniworld_amd.env import MiniWorldEnv
    from gym_miniworld_amd.vec_env import MiniWorldVecEnv
    env = MiniWorldEnv("MiniWorld-YMazeLeft-v0", seed=4)
    env.reset()
    assert env.action_space.n == 3 and len(env.rooms) == 6 and env.rooms[1].num_walls == 3
    # place_entity(min_x = max_x = 3.9, ...) draws uniform(3.9 + r, 3.9 - r): within a box radius of the goal (ymaze.py:65-72)
    assert abs(env.box.pos[0] - 3.9) <= env.box.radius and abs(env.box.pos[2] + 7.0) <= env.box.radius
    e = oracle_mod.OracleEnv("YMaze", seed=4, task_args=[1, 3.9, -7.0, 0])
    e.reset(render=False)
    rng = np.random.default_rng(1)
    got = False
    for t in range(280):
        a = seek_action(e.state(), rng)
        obs, r, d, info = env.step(a)
        _, ro, do, _ = e.step(a)
        assert (r, d) == (ro, do) and list(info["goal_pos"]) == list(e.state().box_pos)
        if d:
            got = r > 0
            break
    assert got
    env.close()
    v = MiniWorldVecEnv("MiniWorld-YMaze-v0", 8, seed=5, to_float=False)
    o0 = v.reset()
    h = oracle_mod.OracleEnv("YMaze", seed=5, task_args=[0, 0, 0, 0])
    ref = h.reset()
    got0 = o0[0].cpu().numpy().transpose(2, 1, 0)   # [3, W, H] -> HWC
    assert obs_diff(got0, ref).max() <= 1
    for t in range(10):
        obs, rew, done, infos = v.step(torch.randint(0, 3, (8, 1)))
    assert obs.shape == (8, 3, 80, 60) and "goal_pos" in infos[0]
    v.close()
