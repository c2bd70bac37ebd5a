"""GPU parity tests of PutNext (envs/putnext.py): six boxes of per-episode sizes, the full action set
Discrete(8) with pickup / drop and the carry physics of miniworld.py:594-702, through the C ABI against the CPU
oracle (itself bit-exact on the reference's own PutNext trajectories, tests/golden/state_PutNext.npz, and its frames
equal to the brute-force rendition of the reference's GL stream with a carried box, glstream_PutNext_dr*.json).

Bars as elsewhere: world, placement, sizes, colours, RNG stream, every box pose incl. the carry height, agent.carrying,
rewards, dones bit-exact; observations within +-1/255, depth within 1e-4 m."""
import math

import numpy as np
import pytest

from test_gpu_parity import obs_diff

pytestmark = pytest.mark.gpu
ENV_ID = "MiniWorld-PutNext-v0"


def make_pair(O, n, seed, dr, depth=False, layout="HWC", **kw):
    from gym_miniworld_amd.batch import BatchedMiniWorld
    b = BatchedMiniWorld(ENV_ID, num_envs=n, seed=seed, domain_rand=dr, want_depth=depth, layout=layout, **kw)
    envs = [O.OracleEnv("PutNext", seed=seed + i, domain_rand=dr) for i in range(n)]
    return b, envs


def assert_putnext_state_equal(st, envs, tag=""):
    os_ = [e.state() for e in envs]
    arr = lambda f: np.array([f(s) for s in os_])   # noqa: E731
    assert np.array_equal(st["boxes_pos"], arr(lambda s: np.array(s.boxes_pos)[:6])), (tag, "boxes_pos")
    assert np.array_equal(st["boxes_dir"], arr(lambda s: list(s.boxes_dir)[:6])), (tag, "boxes_dir")
    assert np.array_equal(st["boxes_color"], arr(lambda s: np.array(s.boxes_color)[:6])), (tag, "boxes_color")
    assert np.array_equal(st["boxes_size"], arr(lambda s: list(s.boxes_size)[:6])), (tag, "boxes_size")
    assert np.array_equal(st["carrying"], arr(lambda s: s.carrying)), (tag, "carrying")
    assert np.array_equal(st["agent_pos"], arr(lambda s: list(s.agent_pos))) and np.array_equal(st["agent_dir"], arr(lambda s: s.agent_dir)), (tag, "pose")
    assert np.array_equal(st["cam"], arr(lambda s: [s.cam_height, s.cam_fwd_disp, s.cam_pitch, s.cam_fov_y])), (tag, "cam")
    for k in ("sky_color", "light_pos", "light_color", "light_ambient"):
        assert np.array_equal(st[k], arr(lambda s: list(getattr(s, k)))), (tag, k)
    assert np.array_equal(st["step_count"], arr(lambda s: s.step_count)) and np.array_equal(st["rng_pos"], arr(lambda s: s.rng_pos))
    assert np.array_equal(st["rng_keysum"], arr(lambda s: s.rng_keysum).astype(np.uint32)), (tag, "rng")


def fetch_action(s, rng):
    """the fixture generator's policy restated on the oracle's state: fetch the red box, carry it to the yellow one, drop"""
    if rng.random() < 0.06:
        return int(rng.integers(0, 8))
    ax, az, ad = s.agent_pos[0], s.agent_pos[2], s.agent_dir
    red, yel = np.array(s.boxes_pos[4]), np.array(s.boxes_pos[5])
    rr, ry = math.sqrt(2) * s.boxes_size[4] / 2, math.sqrt(2) * s.boxes_size[5] / 2

    def steer(t, stop):
        want = math.atan2(-(t[2] - az), t[0] - ax)
        diff = (want - ad + math.pi) % (2 * math.pi) - math.pi
        if abs(diff) > math.radians(9):
            return (0 if diff > 0 else 1), False
        return 2, math.hypot(t[0] - ax, t[2] - az) < stop
    if s.carrying < 0:
        act, close = steer(red, 0.4 + rr + 0.25)
        return 4 if close else act
    if s.carrying != 4:
        return 5
    if np.linalg.norm(red - yel) < rr + ry + 1.1 * 0.17:
        return 5
    return steer(yel, 0.0)[0]


@pytest.mark.parametrize("dr", [0, 1])
def test_putnext_reset_state_and_first_obs(oracle_mod, dr):
    O = oracle_mod
    n = 16
    b, envs = make_pair(O, n, seed=500, dr=dr, depth=True)
    assert b.n_boxes == 6 and b.n_actions == 8 and b.max_episode_steps == 250
    obs = b.reset().cpu().numpy()
    dep = b.depth.cpu().numpy()[..., 0]
    for e in envs:
        e.reset(render=False)
    assert_putnext_state_equal(b.get_state(), envs, "reset")
    st = b.get_state()
    assert (st["boxes_size"] >= 0.6).all() and (st["boxes_size"] <= 0.85).all() and (st["carrying"] == -1).all()
    rooms, segs = b.get_geometry(3)
    assert rooms.shape[0] == 1 and np.array_equal(segs, envs[3].geometry()["wall_segs"]) and list(rooms[0, :4]) == [0, 12, 0, 12]
    for i, e in enumerate(envs):
        ref, refd = e.render_obs(depth=True)
        assert obs_diff(obs[i], ref).max() <= 1 and np.abs(dep[i] - refd).max() <= 1e-4, (dr, i)
    b.reset()
    for e in envs:
        e.reset(render=False)
    assert_putnext_state_equal(b.get_state(), envs, "2nd reset")
    b.close()


@pytest.mark.parametrize("dr,policy", [(0, "fetch"), (1, "fetch"), (0, "random"), (1, "random")])
def test_putnext_rollout_carry_physics_exact(oracle_mod, dr, policy):
    import torch
    O = oracle_mod
    n, steps = 12, 600
    b, envs = make_pair(O, n, seed=40, dr=dr, depth=True)
    b.reset()
    for e in envs:
        e.reset(render=False)
    rng = np.random.default_rng(8)
    rewards = carried = floated = n_done = 0
    for t in range(steps):
        if policy == "fetch":
            a = np.array([fetch_action(e.state(), rng) for e in envs], dtype=np.int32)
        else:
            a = rng.choice(8, size=n, p=[0.14, 0.14, 0.3, 0.1, 0.2, 0.06, 0.03, 0.03]).astype(np.int32)
        b.step(torch.from_numpy(a))
        rew, done, eps = b.reward64.cpu().numpy(), b.done.cpu().numpy(), b.ep_steps.cpu().numpy()
        for i, e in enumerate(envs):
            _, r, d, _ = e.step(int(a[i]))
            s = e.state()
            assert r == rew[i] and d == bool(done[i]) and s.step_count == eps[i], (dr, policy, t, i, r, rew[i], d, done[i])
            rewards += r > 0
            carried += s.carrying >= 0
            if d:
                e.reset(render=False)
                n_done += 1
        if t % 10 == 9 or t == steps - 1:
            st = b.get_state()
            assert_putnext_state_equal(st, envs, "%s dr%d t=%d" % (policy, dr, t))
            floated += int((st["boxes_pos"][:, :, 1] > 0.3).sum())
        if t % 20 == 19 or t == steps - 1:
            obs, dep = b.obs.cpu().numpy(), b.depth.cpu().numpy()[..., 0]   # the step's own frame, from the device's own state
            for i, e in enumerate(envs):
                ref, refd = e.render_obs(depth=True)
                assert obs_diff(obs[i], ref).max() <= 1 and np.abs(dep[i] - refd).max() <= 1e-4, (dr, policy, t, i)
    assert carried > 100 and floated > 0 and n_done > 0
    if policy == "fetch":
        assert rewards > 0   # the red box was put next to the yellow one
    b.close()


def test_putnext_views_with_a_carried_box_and_fast_path(oracle_mod, monkeypatch):
    """random poses all over the room, every env carrying one of its boxes (injected state: pose, carried index, box at
    its carry position) - six boxes in view from all sides, the carried one filling the lower half of the frame; the
    interior-pixel fast path must equal the 8-sample path bit for bit with six boxes as well"""
    import torch
    O = oracle_mod
    n = 24
    fast, envs = make_pair(O, n, seed=77, dr=1, depth=True)
    monkeypatch.setenv("MWB_DEBUG", "9")
    full, _ = make_pair(O, n, seed=77, dr=1, depth=True)
    monkeypatch.delenv("MWB_DEBUG")
    fast.reset(); full.reset()
    for e in envs:
        e.reset(render=False)
    rs = np.random.default_rng(5)
    for rnd in range(8):
        # a pickup (where something is in reach), some carrying around, then a look
        acts = [4] + list(rs.choice([0, 1, 2, 2, 3], size=6)) + [int(rs.choice([0, 1, 5, 4]))]
        for a in acts:
            av = torch.full((n,), int(a), dtype=torch.int32)
            fast.step(av); full.step(av)
            for e in envs:
                _, _, d, _ = e.step(int(a))
                if d:
                    e.reset(render=False)
        assert torch.equal(fast.obs, full.obs) and torch.equal(fast.depth, full.depth), rnd
        assert_putnext_state_equal(fast.get_state(), envs, "views %d" % rnd)
        obs, dep = fast.obs.cpu().numpy(), fast.depth.cpu().numpy()[..., 0]
        for i, e in enumerate(envs):
            ref, refd = e.render_obs(depth=True)
            assert obs_diff(obs[i], ref).max() <= 1 and np.abs(dep[i] - refd).max() <= 1e-4, (rnd, i)
        # teleport: new random poses next to random boxes, so that pickups succeed in the next round
        st = fast.get_state()
        k = rs.integers(0, 6, n)
        ang = rs.uniform(-math.pi, math.pi, n)
        bp = st["boxes_pos"][np.arange(n), k]
        dist = 0.4 + math.sqrt(2) * st["boxes_size"][np.arange(n), k] / 2 + 0.1
        pos = np.stack([bp[:, 0] - np.cos(ang) * dist, np.zeros(n), bp[:, 2] + np.sin(ang) * dist], axis=1)
        pos[:, [0, 2]] = np.clip(pos[:, [0, 2]], 0.5, 11.5)
        drop = st["carrying"] < 0   # only envs that carry nothing are moved (a carried box would be left behind)
        newpos = np.where(drop[:, None], pos, st["agent_pos"])
        newdir = np.where(drop, ang, st["agent_dir"])
        for h in (fast, full):
            h.set_state(0, agent_pos=newpos, agent_dir=newdir)
        for i, e in enumerate(envs):
            e.set_agent(newpos[i, 0], newpos[i, 2], newdir[i])
    fast.close(); full.close()


def test_putnext_gym_view_and_vecenv(oracle_mod):
    import torch
    from gym_miniworld_amd.env import MiniWorldEnv
    from gym_miniworld_amd.vec_env import MiniWorldVecEnv
    env = MiniWorldEnv(ENV_ID, seed=3)
    env.reset()
    assert env.action_space.n == 8 and len(env.entities) == 7 and env.agent.carrying is None
    assert env.red_box.color == "red" and env.yellow_box.color == "yellow" and 0.6 <= env.red_box.size[0] <= 0.85
    e = oracle_mod.OracleEnv("PutNext", seed=3)
    e.reset(render=False)
    rng = np.random.default_rng(0)
    picked = False
    for t in range(300):
        a = fetch_action(e.state(), rng)
        obs, r, d, _ = env.step(a)
        _, ro, do, _ = e.step(a)
        assert (r, d) == (ro, do)
        s = e.state()
        assert (env.agent.carrying is None) == (s.carrying < 0)
        if s.carrying >= 0:
            picked = True
            assert env.agent.carrying is env.boxes[s.carrying] and env.agent.carrying.pos[1] > 0.3
        if d:
            break
    assert picked
    env.close()
    v = MiniWorldVecEnv(ENV_ID, 8, seed=5, to_float=False)
    v.reset()
    assert v.action_space.n == 8
    for t in range(12):
        obs, rew, done, infos = v.step(torch.randint(0, 8, (8, 1)))
    assert obs.shape == (8, 3, 80, 60) and len(infos) == 8
    v.close()


@pytest.mark.parametrize("W,H", [(97, 41), (160, 120), (48, 100)])
def test_putnext_and_ymaze_at_other_frame_sizes(oracle_mod, W, H):
    """the per-item box masks, the lattice pre-test (more than 7 strips at 160 wide: two passes) and the polygon rooms are not
    specialised for 80 x 60: six boxes / polygon rooms at odd sizes, RGB + depth against the oracle, fast path == 8-sample path"""
    import torch
    from gym_miniworld_amd.batch import BatchedMiniWorld
    O = oracle_mod
    n = 8
    for env_id, task, args in ((ENV_ID, "PutNext", None), ("MiniWorld-YMaze-v0", "YMaze", [0, 0, 0, 0])):
        b = BatchedMiniWorld(env_id, num_envs=n, seed=21, domain_rand=True, obs_width=W, obs_height=H, want_depth=True)
        import os
        os.environ["MWB_DEBUG"] = "9"
        try:
            full = BatchedMiniWorld(env_id, num_envs=n, seed=21, domain_rand=True, obs_width=W, obs_height=H, want_depth=True)
        finally:
            del os.environ["MWB_DEBUG"]
        envs = [O.OracleEnv(task, seed=21 + i, domain_rand=True, task_args=args, obs_width=W, obs_height=H) for i in range(n)]
        b.reset(); full.reset()
        for e in envs:
            e.reset(render=False)
        rng = np.random.default_rng(1)
        for rnd in range(3):
            for t in range(10 * rnd):
                a = rng.integers(0, 3, n).astype(np.int32)
                b.step(torch.from_numpy(a)); full.step(torch.from_numpy(a))
                for i, e in enumerate(envs):
                    _, _, d, _ = e.step(int(a[i]))
                    if d:
                        e.reset(render=False)
            assert torch.equal(b.obs, full.obs) and torch.equal(b.depth, full.depth), (env_id, W, H, rnd)
            obs, dep = b.obs.cpu().numpy(), b.depth.cpu().numpy()[..., 0]
            for i, e in enumerate(envs):
                ref, refd = e.render_obs(depth=True)
                assert obs_diff(obs[i], ref).max() <= 1 and np.abs(dep[i] - refd).max() <= 1e-4, (env_id, W, H, rnd, i)
        b.close(); full.close()
