"""GPU parity tests of the T-maze family (envs/tmaze.py): TMaze, TMazeLeft / Right,
TMazeDynamic and the two-box variants with info['feature'], through the C ABI against the CPU oracle
(itself pinned bit-exactly to reference vectors, tests/golden/state_TMaze*.npz).

Same bars as test_gpu_parity.py: world, placement, RNG stream, rewards (incl. the negative penalty
reward), dones, features, goal positions and the goal-alternation counters bit-exact; observations
within +-1/255, depth within 1e-4 m.
"""
import math

import numpy as np
import pytest

from test_gpu_parity import assert_state_equal, obs_diff, oracle_states

pytestmark = pytest.mark.gpu

# env id, oracle task, task_args (None = the id's own), i.e. the registered classes ...
IDS = [
    ("MiniWorld-TMaze-v0", "TMaze", [0, 0, 0, 0]),
    ("MiniWorld-TMazeLeft-v0", "TMaze", [1, 10, -6, 0]),
    ("MiniWorld-TMazeRight-v0", "TMaze", [1, 10, 6, 0]),
    ("MiniWorld-TMazeDynamic-v0", "TMaze", [1, 10, -6, 100]),
    ("MiniWorld-TMazeTwoBoxDynamic-v0", "TMazeTwoBox", [0, 0, 0, 100]),
    ("MiniWorld-TMazeTwoBoxDynamicFeatures100K-v0", "TMazeTwoBox", [1, 0, 0, 100000]),
    ("MiniWorld-TMazeTwoBoxDynamicFeaturesDebug-v0", "TMazeTwoBox", [1, 0, 0, 9000000000000]),
]
# ... and the same classes with a short sub_task_length so that the alternation rules fire in a rollout
SHORT = [
    ("TMaze", [0, 0, 0, 0]), ("TMaze", [1, 10, -6, 2]), ("TMazeTwoBox", [0, 0, 0, 2]), ("TMazeTwoBox", [1, 0, 0, 150]),
]


def make_pair(O, task, args, n, seed, dr, env_id=None, depth=False, layout="HWC"):
    from gym_miniworld_amd.batch import BatchedMiniWorld
    if env_id is not None:
        b = BatchedMiniWorld(env_id, num_envs=n, seed=seed, domain_rand=dr, want_depth=depth, layout=layout)
    else:
        b = BatchedMiniWorld(None, num_envs=n, seed=seed, domain_rand=dr, want_depth=depth, layout=layout,
                             task=task, task_args=args)
    envs = [O.OracleEnv(task, seed=seed + i, domain_rand=dr, task_args=args) for i in range(n)]
    return b, envs


def assert_tmaze_state_equal(st, os_, two_box, exact_pose=True, tag=""):
    assert_state_equal(st, os_, exact_pose=exact_pose, tag=tag)
    assert np.array_equal(st["goal_idx"], np.array([s.goal_idx for s in os_])), (tag, "goal_idx")
    assert np.array_equal(st["episode_count"], np.array([s.episode_count for s in os_])), (tag, "episode_count")
    assert np.array_equal(st["task_step_count"], np.array([s.task_step_count for s in os_])), (tag, "task_step_count")
    if two_box:
        assert np.array_equal(st["box2_pos"], np.array([list(s.box2_pos) for s in os_])), (tag, "box2_pos")
        assert np.array_equal(st["box2_dir"], np.array([s.box2_dir for s in os_])), (tag, "box2_dir")
        assert np.array_equal(st["box2_color"], np.array([list(s.box2_color) for s in os_])), (tag, "box2_color")


@pytest.mark.parametrize("env_id,task,args", IDS)
@pytest.mark.parametrize("dr", [0, 1])
def test_tmaze_reset_state_geometry_and_first_obs(oracle_mod, env_id, task, args, dr):
    O = oracle_mod
    n = 16
    b, envs = make_pair(O, task, args, n, seed=300, dr=dr, env_id=env_id, depth=True)
    assert b.max_episode_steps == 280 and envs[0].state().max_episode_steps == 280   # tmaze.py:20,142
    obs = b.reset().cpu().numpy()
    dep = b.depth.cpu().numpy()[..., 0]
    for e in envs:
        e.reset(render=False)
    two = task == "TMazeTwoBox"
    assert_tmaze_state_equal(b.get_state(), oracle_states(envs), two, tag=env_id)
    for i in (0, n - 1):
        rooms, segs = b.get_geometry(i)
        g = envs[i].geometry()
        assert rooms.shape[0] == 2 and np.array_equal(segs, g["wall_segs"]) and segs.shape[0] == 8
        o = g["outline"]
        rect = np.stack([o[:, :, 0].min(1), o[:, :, 0].max(1), o[:, :, 1].min(1), o[:, :, 1].max(1)], axis=1)
        assert np.array_equal(rooms[:, 0:4], rect.astype(np.float32))
        tex = rooms[:, 5].view(np.int32)
        assert np.array_equal(np.stack([tex & 255, (tex >> 8) & 255, (tex >> 16) & 255], axis=1), g["tex_ids"])
    for i, e in enumerate(envs):
        ref, refd = e.render_obs(depth=True)
        d = obs_diff(obs[i], ref)
        assert d.max() <= 1, (env_id, dr, i, int(d.max()), int((d > 1).sum()))
        assert np.abs(dep[i] - refd).max() <= 1e-4, (env_id, dr, i)
    b.reset()   # a second reset continues the RNG streams and the episode counters
    for e in envs:
        e.reset(render=False)
    assert_tmaze_state_equal(b.get_state(), oracle_states(envs), two, tag=env_id + " 2nd reset")
    b.close()


def seek_action(s, target_xz, rng):
    """Turn towards a target, then walk; leaves the stem of the T through its mouth first."""
    ax, az = s.agent_pos[0], s.agent_pos[2]
    tx, tz = (10.0, 0.0) if ax < 9.2 else target_xz
    want = math.atan2(-(tz - az), tx - ax)
    diff = (want - s.agent_dir + math.pi) % (2 * math.pi) - math.pi
    if rng.random() < 0.05:
        return int(rng.integers(0, 3))
    if abs(diff) > math.radians(10):
        return 0 if diff > 0 else 1
    return 2


@pytest.mark.parametrize("task,args", SHORT)
@pytest.mark.parametrize("dr", [0, 1])
def test_tmaze_rollout_rewards_features_and_alternation_exact(oracle_mod, task, args, dr):
    """Box-seeking rollouts with auto-reset: reward (positive at the goal box, negative at the penalty box),
    done, step_count, info['feature'], info['goal_pos'] exact at every step; the full state incl. goal_idx and
    the alternation counters, and the observation (oracle pose injected), every 25 steps."""
    import torch
    O = oracle_mod
    n, steps = 16, 700
    two = task == "TMazeTwoBox"
    b, envs = make_pair(O, task, args, n, seed=4000, dr=dr)
    b.reset()
    for e in envs:
        e.reset(render=False)
    rng = np.random.default_rng(11)
    pos_rewards = neg_rewards = flips = 0
    goal_prev = [e.state().goal_idx for e in envs]
    for t in range(steps):
        sts = oracle_states(envs)
        a = np.zeros(n, np.int32)
        for i, s in enumerate(sts):
            # even envs walk to the red box (box 0), odd ones to the blue box where there is one
            tgt = s.box2_pos if (two and i % 2) else s.box_pos
            a[i] = seek_action(s, (tgt[0], tgt[2]), rng)
        b.step(torch.from_numpy(a))
        rew, done = b.reward64.cpu().numpy(), b.done.cpu().numpy()
        eps, feat, gpos = b.ep_steps.cpu().numpy(), b.feature.cpu().numpy(), b.goal_pos.cpu().numpy()
        for i, e in enumerate(envs):
            _, r, d, _ = e.step(int(a[i]))
            s = e.state()
            assert r == rew[i] and d == bool(done[i]) and s.step_count == eps[i], (task, args, dr, t, i, r, rew[i], d, done[i])
            assert list(s.feature) == list(feat[i]), (t, i, list(s.feature), feat[i])
            gb = s.box2_pos if (two and s.goal_idx == 1) else s.box_pos
            assert list(gb) == list(gpos[i]), (t, i)
            pos_rewards += r > 0
            neg_rewards += r < 0
            if d:
                e.reset(render=False)
                g = e.state().goal_idx
                flips += g != goal_prev[i]
                goal_prev[i] = g
        if t % 25 == 24 or t == steps - 1:
            os_ = oracle_states(envs)
            assert_tmaze_state_equal(b.get_state(), os_, two, tag="%s t=%d" % (task, t))
            obs = b.obs.cpu().numpy()   # the step's own frame, from the device's own (bit-equal) pose
            for i, e in enumerate(envs):
                d = obs_diff(obs[i], e.render_obs())
                assert d.max() <= 1, (task, dr, t, i, int(d.max()), int((d > 1).sum()))
    assert pos_rewards > 0
    if two:
        assert neg_rewards > 0
    if args[3]:
        assert flips > 0   # the alternation rule fired
    b.close()


@pytest.mark.parametrize("task,args", [("TMaze", [0, 0, 0, 0]), ("TMazeTwoBox", [1, 0, 0, 150])])
@pytest.mark.parametrize("dr", [0, 1])
def test_tmaze_random_views_in_the_bar_match_oracle(oracle_mod, task, args, dr):
    """Random poses inside the bar of the T and at its mouth (both boxes, the portal and the stem in view from
    all sides): every pixel within +-1 LSB, depth within 1e-4 m."""
    O = oracle_mod
    n = 48
    b, envs = make_pair(O, task, args, n, seed=77, dr=dr, depth=True)
    b.reset()
    for e in envs:
        e.reset(render=False)
    st = oracle_states(envs)
    rng = np.random.default_rng(23)
    for rnd in range(5):
        pos = np.zeros((n, 2))
        for i, s in enumerate(st):
            boxes = [s.box_pos] + ([s.box2_pos] if s.n_boxes == 2 else [])
            while True:   # a free spot: inside a room, the eye outside every box
                if rng.random() < 0.25:
                    p = np.array([rng.uniform(5.0, 8.5), rng.uniform(-1.5, 1.5)])
                else:
                    p = np.array([rng.uniform(8.5, 11.5), rng.uniform(-7.5, 7.5)])
                if all(math.hypot(p[0] - bx[0], p[1] - bx[2]) > 0.75 for bx in boxes):
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
            assert d.max() <= 1, (task, dr, rnd, i, int(d.max()), int((d > 1).sum()))
            assert np.abs(dep[i] - refd).max() <= 1e-4, (task, dr, rnd, i)
    b.close()


def test_two_box_fast_path_equals_full_sample_path(monkeypatch):
    """The corner-ray interior classification with two boxes against the kernel's own 8-sample path
    (MWB_DEBUG=1) on a large batch stepped identically."""
    import torch
    from gym_miniworld_amd.batch import BatchedMiniWorld
    n, steps = 1024, 60
    outs = []
    for dbg in ("0", "1"):
        monkeypatch.setenv("MWB_DEBUG", dbg)
        b = BatchedMiniWorld("MiniWorld-TMazeTwoBoxDynamicFeatures100K-v0", num_envs=n, seed=5, domain_rand=True,
                             want_depth=True)
        b.reset()
        g = torch.Generator().manual_seed(3)
        # start every env somewhere in the bar so that boxes are in view
        rs = np.random.default_rng(1)
        b.set_agent(0, pos_xz=np.stack([rs.uniform(8.6, 11.4, n), rs.uniform(-4.5, 4.5, n)], axis=1),
                    dir=rs.uniform(-np.pi, np.pi, n))
        for _ in range(steps):
            b.step(torch.randint(0, 3, (n,), generator=g, dtype=torch.int32))
        outs.append((b.obs.cpu().numpy().copy(), b.depth.cpu().numpy().copy(), b.reward64.cpu().numpy().copy()))
        b.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert np.array_equal(outs[0][2], outs[1][2])


def test_task_state_hook_and_large_counters(oracle_mod):
    """mwb_set_task_state: counters far beyond 32 bits (FeaturesDebug's 9e12 threshold) and an episode count
    just below a multiple of sub_task_length behave as in the oracle."""
    O = oracle_mod
    from gym_miniworld_amd.batch import BatchedMiniWorld
    args = [1, 0, 0, 9000000000000]
    b = BatchedMiniWorld(None, num_envs=4, seed=9, task="TMazeTwoBox", task_args=args)
    envs = [O.OracleEnv("TMazeTwoBox", seed=9 + i, task_args=args) for i in range(4)]
    tsc = np.array([0, 9000000000000, 9000000000001, 2 ** 62], np.int64)
    b.set_task_state(0, task_step_count=tsc, goal_idx=np.array([0, 1, 0, 1], np.int32))
    for e, c, g in zip(envs, tsc, (0, 1, 0, 1)):
        e.set_counters(0, int(c), g)
        e.reset(render=False)
    b.reset()
    assert_tmaze_state_equal(b.get_state(), oracle_states(envs), True)
    assert list(b.get_state()["goal_idx"]) == [0, 1, 1, 0]   # strictly greater than the threshold flips
    b.close()
    args = [0, 0, 0, 100]
    b = BatchedMiniWorld("MiniWorld-TMazeTwoBoxDynamic-v0", num_envs=3, seed=1)
    envs = [O.OracleEnv("TMazeTwoBox", seed=1 + i, task_args=args) for i in range(3)]
    assert list(b.get_state()["episode_count"]) == [1, 1, 1]   # the constructor's own reset (miniworld.py:523)
    epc = np.array([98, 99, 2 ** 40 * 100 - 1], np.int64)
    b.set_task_state(0, episode_count=epc)
    for e, c in zip(envs, epc):
        e.set_counters(int(c), 0, 0)
        e.reset(render=False)
    b.reset()
    assert_tmaze_state_equal(b.get_state(), oracle_states(envs), True)
    assert list(b.get_state()["goal_idx"]) == [0, 1, 1]
    with pytest.raises(Exception):
        b.set_task_state(0, goal_idx=np.array([2], np.int32))
    b.close()


def test_intersect_reports_both_boxes(oracle_mod):
    O = oracle_mod
    from gym_miniworld_amd.batch import BatchedMiniWorld
    b = BatchedMiniWorld("MiniWorld-TMazeTwoBoxDynamic-v0", num_envs=2, seed=3)
    b.reset()
    e = O.OracleEnv("TMazeTwoBox", seed=3, task_args=[0, 0, 0, 100])
    e.reset(render=False)
    s = e.state()
    assert b.intersect(0, s.box_pos[0] + 0.3, s.box_pos[2]) == 2       # the red box (entity 0)
    assert b.intersect(0, s.box2_pos[0], s.box2_pos[2] - 0.3) == 3     # the blue box (entity 1)
    assert b.intersect(0, 7.9, 3.0) == 1 and b.intersect(0, 3.0, 0.0) == 0
    for (x, z) in ((s.box_pos[0] + 0.3, s.box_pos[2]), (s.box2_pos[0], s.box2_pos[2] - 0.3), (7.9, 3.0), (3.0, 0.0)):
        assert bool(b.intersect(0, x, z)) == bool(e.intersect_agent(x, z))
    b.close()


def test_tmaze_vecenv_infos(oracle_mod):
    """VecEnv front-end: infos carry 'goal_pos' (tmaze.py:66,206) and, for the *Features* classes, 'feature'
    (tmaze.py:311-318) of the transition, also across an auto-reset."""
    import torch
    from gym_miniworld_amd.vec_env import MiniWorldVecEnv
    O = oracle_mod
    n = 4
    v = MiniWorldVecEnv("MiniWorld-TMazeTwoBoxDynamicFeatures100K-v0", n, seed=21, to_float=False)
    envs = [O.OracleEnv("TMazeTwoBox", seed=21 + i, task_args=[1, 0, 0, 100000]) for i in range(n)]
    v.reset()
    for e in envs:
        e.reset(render=False)
    rng = np.random.default_rng(2)
    seen_feature = False
    for t in range(400):
        a = np.array([seek_action(e.state(), (e.state().box2_pos[0], e.state().box2_pos[2]) if i % 2 else
                                  (e.state().box_pos[0], e.state().box_pos[2]), rng) for i, e in enumerate(envs)])
        _, rews, dones, infos = v.step(torch.from_numpy(a).unsqueeze(1))
        for i, e in enumerate(envs):
            _, r, d, _ = e.step(int(a[i]))
            s = e.state()
            assert np.float32(r) == rews[i, 0].item() and d == dones[i]
            assert list(infos[i]["feature"]) == list(s.feature)
            gb = s.box2_pos if s.goal_idx == 1 else s.box_pos
            assert list(infos[i]["goal_pos"]) == list(gb)
            seen_feature |= bool(s.feature[0] or s.feature[1])
            if d:
                e.reset(render=False)
    assert seen_feature
    v.close()
