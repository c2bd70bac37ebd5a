"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on identical seeds.

Bars (BASELINE.json north_star): rewards / dones / step counters / RNG stream bit-exact; world
geometry and placement bit-exact (float64); observations within +-1/255 per channel; depth within
1e-4 m.  Poses are bit-exact too: the step / prep kernels evaluate sin / cos with the bit-for-bit restatement
of glibc's routines (csrc/mwb_glibc_trig.h, tests/test_glibc_trig.py), so nothing is injected inside rollouts.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [
    ("MiniWorld-Hallway-v0", "Hallway", None, 0),
    ("MiniWorld-OneRoom-v0", "OneRoom", None, 0),
    ("MiniWorld-FourRooms-v0", "FourRooms", None, 0),
    ("MiniWorld-Maze-v0", "Maze", None, 0),
    ("MiniWorld-MazeS3-v0", "Maze", [3, 3, 3], 0),
    ("MiniWorld-OneRoomS6-v0", "OneRoom", [6], 100),
]


def make_pair(O, env_id, task, args, mes, n, seed, dr, depth=False, layout="HWC"):
    from gym_miniworld_amd.batch import BatchedMiniWorld
    b = BatchedMiniWorld(env_id, num_envs=n, seed=seed, domain_rand=dr, want_depth=depth, layout=layout)
    envs = [O.OracleEnv(task, seed=seed + i, domain_rand=dr, task_args=args, max_episode_steps=mes) for i in range(n)]
    return b, envs


def oracle_states(envs):
    return [e.state() for e in envs]


def assert_state_equal(st, os_, exact_pose=True, tag=""):
    n = len(os_)
    f = lambda name: np.array([list(getattr(s, name)) if hasattr(getattr(s, name), "__len__") else getattr(s, name) for s in os_])  # noqa: E731
    for k in ("box_pos", "box_dir", "box_color", "sky_color", "light_pos", "light_color", "light_ambient"):
        assert np.array_equal(st[k], f(k)), (tag, k)
    cam = np.array([[s.cam_height, s.cam_fwd_disp, s.cam_pitch, s.cam_fov_y] for s in os_])
    assert np.array_equal(st["cam"], cam), (tag, "cam")
    assert np.array_equal(st["step_count"], f("step_count")), (tag, "step_count")
    assert np.array_equal(st["rng_pos"], f("rng_pos")), (tag, "rng_pos")
    assert np.array_equal(st["rng_keysum"], f("rng_keysum").astype(np.uint32)), (tag, "rng_keysum")
    assert np.array_equal(st["n_rooms"], f("n_rooms")), (tag, "n_rooms")
    assert np.array_equal(st["n_segs"], f("n_segs")), (tag, "n_segs")
    assert exact_pose, "poses are bit-exact since round 2; no caller may ask for less"
    assert np.array_equal(st["agent_pos"], f("agent_pos")), (tag, "agent_pos")
    assert np.array_equal(st["agent_dir"], f("agent_dir")), (tag, "agent_dir")
    assert n == len(st["agent_dir"])


@pytest.mark.parametrize("env_id,task,args,mes", CASES)
@pytest.mark.parametrize("dr", [0, 1])
def test_reset_state_and_geometry_bit_exact(oracle_mod, env_id, task, args, mes, dr):
    O = oracle_mod
    n = 24
    b, envs = make_pair(O, env_id, task, args, mes, n, seed=100, dr=dr)
    b.reset()
    for e in envs:
        e.reset(render=False)
    assert_state_equal(b.get_state(), oracle_states(envs), tag=env_id)
    for i in (0, 5, n - 1):
        rooms, segs = b.get_geometry(i)
        g = envs[i].geometry()
        assert np.array_equal(segs, g["wall_segs"]), (env_id, i, "segs")
        tex = rooms[:, 5].view(np.int32)
        got = np.stack([tex & 255, (tex >> 8) & 255, (tex >> 16) & 255], axis=1)
        assert np.array_equal(got, g["tex_ids"]), (env_id, i, "tex ids")
        o = g["outline"]
        rect = np.stack([o[:, :, 0].min(1), o[:, :, 0].max(1), o[:, :, 1].min(1), o[:, :, 1].max(1)], axis=1)
        assert np.array_equal(rooms[:, 0:4], rect.astype(np.float32)), (env_id, i, "rects")
        assert np.array_equal(rooms[:, 4], g["wall_height"].astype(np.float32))
    # a second reset continues every env's RNG stream exactly
    b.reset()
    for e in envs:
        e.reset(render=False)
    assert_state_equal(b.get_state(), oracle_states(envs), tag=env_id + " 2nd reset")
    b.close()


def obs_diff(a, b):
    d = np.abs(a.astype(np.int16) - b.astype(np.int16))
    return d


@pytest.mark.parametrize("env_id,task,args,mes", CASES)
@pytest.mark.parametrize("dr", [0, 1])
def test_render_after_reset_matches_oracle(oracle_mod, env_id, task, args, mes, dr):
    O = oracle_mod
    n = 16
    b, envs = make_pair(O, env_id, task, args, mes, n, seed=7, dr=dr, depth=True)
    obs = b.reset().cpu().numpy()
    dep = b.depth.cpu().numpy()[..., 0]
    for i, e in enumerate(envs):
        e.reset(render=False)
        ref, refd = e.render_obs(depth=True)
        d = obs_diff(obs[i], ref)
        assert d.max() <= 1, (env_id, dr, i, int(d.max()), int((d > 1).sum()))
        assert np.abs(dep[i] - refd).max() <= 1e-4, (env_id, dr, i, float(np.abs(dep[i] - refd).max()))
    assert 0 < obs.mean() < 255   # run_tests.py:21-22 sanity
    b.close()


@pytest.mark.parametrize("env_id,task,args,mes,steps", [
    ("MiniWorld-Hallway-v0", "Hallway", None, 0, 300),
    ("MiniWorld-OneRoom-v0", "OneRoom", None, 0, 400),
    ("MiniWorld-FourRooms-v0", "FourRooms", None, 0, 300),
    ("MiniWorld-MazeS3-v0", "Maze", [3, 3, 3], 0, 300),
    ("MiniWorld-Maze-v0", "Maze", None, 0, 120),
])
@pytest.mark.parametrize("dr", [0, 1])
def test_rollout_rewards_dones_exact_and_obs_close(oracle_mod, env_id, task, args, mes, steps, dr):
    """Random-action rollout with auto-reset: reward / done / step_count exact every step; every 25 steps the
    whole state - the accumulated pose included, bit for bit - and the observation rendered from the device's
    own state."""
    import torch
    O = oracle_mod
    n = 16
    b, envs = make_pair(O, env_id, task, args, mes, n, seed=2024, dr=dr)
    b.reset()
    for e in envs:
        e.reset(render=False)
    rng = np.random.default_rng(5)
    n_done = 0
    for t in range(steps):
        # bias towards moving forward so that walls, the box and resets are actually reached
        a = rng.choice(3, size=n, p=[0.2, 0.2, 0.6]).astype(np.int32)
        b.step(torch.from_numpy(a))
        rew = b.reward64.cpu().numpy()
        done = b.done.cpu().numpy()
        eps = b.ep_steps.cpu().numpy()
        for i, e in enumerate(envs):
            _, r, d, _ = e.step(int(a[i]))
            assert r == rew[i] and d == bool(done[i]) and e.state().step_count == eps[i], (env_id, dr, t, i, r, rew[i], d, done[i])
            if d:
                e.reset(render=False)
                n_done += 1
        if t % 25 == 24 or t == steps - 1:
            os_ = oracle_states(envs)
            assert_state_equal(b.get_state(), os_, tag="%s t=%d" % (env_id, t))
            obs = b.obs.cpu().numpy()   # the frame the step itself left behind
            for i, e in enumerate(envs):
                d = obs_diff(obs[i], e.render_obs())
                assert d.max() <= 1, (env_id, dr, t, i, int(d.max()), int((d > 1).sum()))
    assert n_done > 0 or steps < 200
    b.close()


def test_cwh_layout_is_transpose_of_hwc(oracle_mod):
    """TransposeImage (pytorch-a2c-ppo-acktr/envs.py:106-107): obs.transpose(2, 1, 0)."""
    from gym_miniworld_amd.batch import BatchedMiniWorld
    a = BatchedMiniWorld("MiniWorld-FourRooms-v0", num_envs=8, seed=3, layout="HWC")
    c = BatchedMiniWorld("MiniWorld-FourRooms-v0", num_envs=8, seed=3, layout="CWH")
    oa = a.reset().cpu().numpy()
    oc = c.reset().cpu().numpy()
    assert oc.shape == (8, 3, 80, 60)
    assert np.array_equal(oc, oa.transpose(0, 3, 2, 1))
    a.close(); c.close()


def test_skip_mask_dummy_semantics(oracle_mod):
    """vec_env/subproc_vec_env.py:26-31,58-67: mask[i] != 0 -> no step, reward -99, done False."""
    import torch
    from gym_miniworld_amd.batch import BatchedMiniWorld
    b = BatchedMiniWorld("MiniWorld-OneRoom-v0", num_envs=8, seed=11)
    b.reset()
    before = b.get_state()
    obs0 = b.obs.cpu().numpy().copy()
    mask = torch.tensor([1, 0, 1, 0, 1, 0, 1, 0], dtype=torch.uint8)
    b.step(torch.full((8,), 2, dtype=torch.int32), skip_mask=mask)
    after = b.get_state()
    rew = b.reward.cpu().numpy()
    assert np.all(rew[0::2] == -99) and np.all(b.done.cpu().numpy()[0::2] == 0)
    assert np.array_equal(after["step_count"][0::2], before["step_count"][0::2])
    assert np.array_equal(after["step_count"][1::2], before["step_count"][1::2] + 1)
    assert np.array_equal(after["agent_pos"][0::2], before["agent_pos"][0::2])
    assert np.array_equal(b.obs.cpu().numpy()[0::2], obs0[0::2])
    b.close()


def test_large_batch_properties():
    """BASELINE sizes (4096 OneRoom envs): size-independent properties - run_tests.py:51-59 (agent
    never leaves the room), run_tests.py:69-72 (spawn never intersects), shape/dtype, brightness."""
    import torch
    from gym_miniworld_amd.batch import BatchedMiniWorld
    n = 4096
    b = BatchedMiniWorld("MiniWorld-OneRoom-v0", num_envs=n, seed=1)
    obs = b.reset()
    assert obs.shape == (n, 60, 80, 3) and obs.dtype == torch.uint8
    st = b.get_state()
    for i in range(0, n, 257):
        assert b.intersect(i, st["agent_pos"][i, 0], st["agent_pos"][i, 2]) == 0
    g = torch.Generator().manual_seed(0)
    for _ in range(60):
        b.step(torch.randint(0, 3, (n,), generator=g, dtype=torch.int32))
    st = b.get_state()
    assert np.all(st["agent_pos"][:, [0, 2]] >= 0) and np.all(st["agent_pos"][:, [0, 2]] <= 10)
    m = b.obs.float().mean().item()
    assert 20 < m < 235
    # envs are independent: the first 16 envs of the big batch equal a 16-env batch with the same seeds
    small = BatchedMiniWorld("MiniWorld-OneRoom-v0", num_envs=16, seed=1)
    small.reset()
    g = torch.Generator().manual_seed(0)
    for _ in range(60):
        a = torch.randint(0, 3, (n,), generator=g, dtype=torch.int32)
        small.step(a[:16])
    assert torch.equal(small.obs, b.obs[:16])
    b.close(); small.close()


@pytest.mark.parametrize("env_id,task,n,depth,dr", [
    ("MiniWorld-Maze-v0", "Maze", 8192, True, False),          # BASELINE configs[2]: Maze, 8192 envs, RGB + depth
    ("MiniWorld-FourRooms-v0", "FourRooms", 16384, False, True),   # BASELINE configs[3]: FourRooms + domain randomisation
])
def test_baseline_configs_at_full_size(oracle_mod, env_id, task, n, depth, dr):
    """The BASELINE.json configurations at their full batch size, through size-independent properties: every env is
    rendered by every step (a zeroed buffer comes back non-zero everywhere, also with the cost-ordered dispatch and
    the half-frame tail), a step's frames equal a plain render of the same state, the first 16 envs equal a 16-env
    batch with the same seeds (shard independence), a sample of envs equals the oracle (state bit-exact, obs +-1 LSB,
    depth 1e-4), agents stay inside the world, spawns do not intersect (run_tests.py:51-72)."""
    import torch
    from gym_miniworld_amd.batch import BatchedMiniWorld
    O = oracle_mod
    b = BatchedMiniWorld(env_id, num_envs=n, seed=1, domain_rand=dr, want_depth=depth)
    small = BatchedMiniWorld(env_id, num_envs=16, seed=1, domain_rand=dr, want_depth=depth)
    sample = [0, 1, 777, n // 2, n - 2, n - 1]
    refs = {i: O.OracleEnv(task, seed=1 + i, domain_rand=dr) for i in sample}
    obs = b.reset()
    small.reset()
    for e in refs.values():
        e.reset(render=False)
    assert obs.shape == (n, 60, 80, 3) and int((obs.reshape(n, -1).max(dim=1).values == 0).sum()) == 0
    st = b.get_state()
    for i in range(0, n, 1021):
        assert b.intersect(i, st["agent_pos"][i, 0], st["agent_pos"][i, 2]) == 0
    g = torch.Generator().manual_seed(0)
    for t in range(30):
        a = torch.randint(0, 3, (n,), generator=g, dtype=torch.int32)
        b.obs.zero_()
        b.step(a)
        small.step(a[:16])
        for i, e in refs.items():
            _, r, dn, _ = e.step(int(a[i]))
            if dn:
                e.reset(render=False)
        assert int((b.obs.reshape(n, -1).max(dim=1).values == 0).sum()) == 0, (env_id, t, "an env was not rendered")
    assert torch.equal(small.obs, b.obs[:16]) and torch.equal(small.reward64, b.reward64[:16])
    if depth:
        assert torch.equal(small.depth, b.depth[:16])
    step_obs = b.obs.clone()
    b.render()
    assert torch.equal(step_obs, b.obs)
    st = b.get_state()
    lim = (-8.0, 8.0) if task == "FourRooms" else (0.0, 8 * 3.25)
    assert np.all(st["agent_pos"][:, [0, 2]] >= lim[0]) and np.all(st["agent_pos"][:, [0, 2]] <= lim[1])
    obs = b.obs.cpu().numpy()
    dep = b.depth.cpu().numpy()[..., 0] if depth else None
    for i, e in refs.items():
        s = e.state()
        assert list(st["agent_pos"][i]) == list(s.agent_pos) and st["agent_dir"][i] == s.agent_dir and st["rng_pos"][i] == s.rng_pos
        ref, refd = e.render_obs(depth=True)
        assert np.abs(obs[i].astype(np.int16) - ref.astype(np.int16)).max() <= 1, (env_id, i)
        if depth:
            assert np.abs(dep[i] - refd).max() <= 1e-4
    assert 20 < b.obs.float().mean().item() < 235
    b.close(); small.close()


@pytest.mark.parametrize("env_id,task,args,mes", [c for c in CASES if c[1] in ("FourRooms", "Maze", "Hallway")])
@pytest.mark.parametrize("dr", [0, 1])
def test_random_view_sweep_matches_oracle(oracle_mod, env_id, task, args, mes, dr):
    """Many viewpoints per world: the agent keeps its spawn position and is turned to random headings
    (identical float64 pose injected on both sides), so oblique views of far portals, lintels and
    the box are covered.  Every pixel within +-1 LSB, depth within 1e-4 m."""
    O = oracle_mod
    n = 48
    b, envs = make_pair(O, env_id, task, args, mes, n, seed=900, dr=dr, depth=True)
    b.reset()
    for e in envs:
        e.reset(render=False)
    st = oracle_states(envs)
    pos = np.array([[s.agent_pos[0], s.agent_pos[2]] for s in st])
    rng = np.random.default_rng(17)
    for rnd in range(6):
        dirs = rng.uniform(-np.pi, np.pi, size=n)
        b.set_agent(0, pos_xz=pos, dir=dirs)
        obs = b.render().cpu().numpy()
        dep = b.depth.cpu().numpy()[..., 0]
        for i, e in enumerate(envs):
            e.set_agent(pos[i, 0], pos[i, 1], dirs[i])
            ref, refd = e.render_obs(depth=True)
            d = obs_diff(obs[i], ref)
            assert d.max() <= 1, (env_id, dr, rnd, i, int(d.max()), int((d > 1).sum()))
            assert np.abs(dep[i] - refd).max() <= 1e-4, (env_id, dr, rnd, i)
    b.close()


@pytest.mark.parametrize("env_id,n,steps,dr", [("MiniWorld-Maze-v0", 2048, 40, 0), ("MiniWorld-FourRooms-v0", 2048, 40, 1),
                                               ("MiniWorld-MazeS3-v0", 1024, 40, 1), ("MiniWorld-Hallway-v0", 1024, 30, 1)])
def test_fast_path_equals_full_sample_path(env_id, n, steps, dr, monkeypatch):
    """The corner-ray interior classification must never change a result: tens of thousands of frames
    rendered with it are bit-identical (obs and depth) to the same frames rendered with every pixel
    resolved by the full 8-sample path (MWB_DEBUG=1 disables the classification at mwb_create), each of its eight
    traversals started in the eye's room (bit 3, MWB_DEBUG=9: no skipping of the portal crossings the corner rays share)."""
    import torch
    from gym_miniworld_amd.batch import BatchedMiniWorld
    fast = BatchedMiniWorld(env_id, num_envs=n, seed=31, domain_rand=dr, want_depth=True)
    monkeypatch.setenv("MWB_DEBUG", "9")
    full = BatchedMiniWorld(env_id, num_envs=n, seed=31, domain_rand=dr, want_depth=True)
    monkeypatch.setenv("MWB_DEBUG", "1")
    full_skip = BatchedMiniWorld(env_id, num_envs=n, seed=31, domain_rand=dr, want_depth=True)
    monkeypatch.delenv("MWB_DEBUG")
    assert torch.equal(fast.reset(), full.reset()) and torch.equal(fast.depth, full.depth)
    assert torch.equal(full_skip.reset(), full.obs) and torch.equal(full_skip.depth, full.depth)
    g = torch.Generator().manual_seed(3)
    for t in range(steps):
        a = torch.randint(0, 3, (n,), generator=g, dtype=torch.int32)
        fast.step(a)
        full.step(a)
        full_skip.step(a)
        assert torch.equal(full_skip.obs, full.obs) and torch.equal(full_skip.depth, full.depth), (env_id, t, "prefix skipping changed a frame")
        same = torch.equal(fast.obs, full.obs) and torch.equal(fast.depth, full.depth)
        if not same:
            bad = (fast.obs != full.obs).flatten(1).any(1).nonzero().flatten().tolist()
            raise AssertionError((env_id, t, "envs with differing pixels", bad[:8], len(bad)))
        assert torch.equal(fast.reward64, full.reward64)
    fast.close(); full.close(); full_skip.close()


@pytest.mark.parametrize("env_id,kwargs,task,args,mes", [
    ("MiniWorld-OneRoomS6Fast-v0", {}, "OneRoom", [6], 50),          # envs/oneroom.py:52-66: params override, no DR
    ("MiniWorld-MazeS3Fast-v0", {}, "Maze", [3, 3, 3], 300),          # envs/maze.py:123-141
    ("MiniWorld-MazeS2-v0", {}, "Maze", [2, 2, 3], 0),                # envs/maze.py:115-117
    ("MiniWorld-Hallway-v0", {"task_args": [6]}, "Hallway", [6], 0),  # Hallway(length=6), hallway.py:13
    ("MiniWorld-Maze-v0", {"task_args": [2, 4, 2.5], "max_episode_steps": 200}, "Maze", [2, 4, 2.5], 200),
    ("MiniWorld-Maze-v0", {"task_args": [10, 11, 3]}, "Maze", [10, 11, 3], 0),   # 219 rooms: > 64 KB of LDS in reset
])
def test_constructor_variants_match_oracle(oracle_mod, env_id, kwargs, task, args, mes):
    """Constructor parameters of the task classes and custom DomainParams tables reach the kernels:
    state bit-exact after reset, rewards/dones exact over a rollout, first observation +-1 LSB."""
    import torch
    from gym_miniworld_amd.batch import BatchedMiniWorld, ENV_SPECS
    O = oracle_mod
    n = 12
    b = BatchedMiniWorld(env_id, num_envs=n, seed=77, **kwargs)
    spec_params = ENV_SPECS[env_id][3]
    table = spec_params().to_table() if spec_params else None
    envs = [O.OracleEnv(task, seed=77 + i, domain_rand=b.domain_rand, task_args=args, max_episode_steps=mes, params=table)
            for i in range(n)]
    obs = b.reset().cpu().numpy()
    for i, e in enumerate(envs):
        assert np.abs(obs[i].astype(int) - e.reset().astype(int)).max() <= 1
    assert_state_equal(b.get_state(), oracle_states(envs), tag=env_id)
    assert b.max_episode_steps == envs[0].state().max_episode_steps
    rng = np.random.default_rng(2)
    for t in range(120):
        a = rng.choice(3, size=n, p=[0.2, 0.2, 0.6]).astype(np.int32)
        b.step(torch.from_numpy(a))
        rew, done = b.reward64.cpu().numpy(), b.done.cpu().numpy()
        for i, e in enumerate(envs):
            _, r, d, _ = e.step(int(a[i]))
            assert r == rew[i] and d == bool(done[i]), (env_id, t, i)
            if d:
                e.reset(render=False)
    assert_state_equal(b.get_state(), oracle_states(envs), tag=env_id + " end")
    b.close()


def test_shard_offset_reproduces_the_global_batch():
    """Multi-GPU sharding contract: a handle created with first_env_index = k owns envs k.. of the global
    batch (seed base + global index), so its results equal the corresponding slice of one big handle."""
    import torch
    from gym_miniworld_amd.batch import BatchedMiniWorld
    big = BatchedMiniWorld("MiniWorld-FourRooms-v0", num_envs=48, seed=5, domain_rand=True)
    shard = BatchedMiniWorld("MiniWorld-FourRooms-v0", num_envs=16, seed=5, domain_rand=True, first_env_index=32)
    assert torch.equal(big.reset()[32:], shard.reset())
    g = torch.Generator().manual_seed(9)
    for _ in range(50):
        a = torch.randint(0, 3, (48,), generator=g, dtype=torch.int32)
        big.step(a); shard.step(a[32:])
        assert torch.equal(big.obs[32:], shard.obs) and torch.equal(big.reward64[32:], shard.reward64)
        assert torch.equal(big.done[32:], shard.done)
    big.close(); shard.close()


@pytest.mark.parametrize("W,H", [(64, 48), (160, 120), (33, 17), (128, 96)])
@pytest.mark.parametrize("layout", ["HWC", "CWH"])
def test_other_observation_sizes_match_oracle(oracle_mod, W, H, layout):
    """obs_width / obs_height are constructor arguments of the reference (miniworld.py:459-460); the kernels are
    not specialised for 80x60: odd sizes, both layouts, RGB + depth against the oracle at the same size."""
    from gym_miniworld_amd.batch import BatchedMiniWorld
    O = oracle_mod
    n = 6
    for env_id, task in (("MiniWorld-FourRooms-v0", "FourRooms"), ("MiniWorld-MazeS3-v0", "Maze")):
        args = [3, 3, 3] if task == "Maze" else None
        b = BatchedMiniWorld(env_id, num_envs=n, seed=40, domain_rand=True, obs_width=W, obs_height=H, want_depth=True,
                             layout=layout)
        envs = [O.OracleEnv(task, seed=40 + i, domain_rand=True, task_args=args, obs_width=W, obs_height=H) for i in range(n)]
        obs = b.reset().cpu().numpy()
        dep = b.depth.cpu().numpy()[..., 0]
        assert obs.shape == ((n, H, W, 3) if layout == "HWC" else (n, 3, W, H))
        for i, e in enumerate(envs):
            e.reset(render=False)
            ref, refd = e.render_obs(depth=True)
            got = obs[i] if layout == "HWC" else obs[i].transpose(2, 1, 0)
            d = obs_diff(got, ref)
            assert d.max() <= 1, (env_id, W, H, layout, i, int(d.max()), int((d > 1).sum()))
            assert np.abs(dep[i] - refd).max() <= 1e-4
        b.close()


@pytest.mark.parametrize("env_id,task,args", [("MiniWorld-FourRooms-v0", "FourRooms", None), ("MiniWorld-MazeS3-v0", "Maze", [3, 3, 3]),
                                              ("MiniWorld-Hallway-v0", "Hallway", None)])
def test_extreme_camera_parameters_match_oracle(oracle_mod, env_id, task, args):
    """A custom DomainParams table far outside the defaults: steep camera pitch both ways, narrow to very wide
    field of view, eyes from ankle height to above the 2.2 m lintels, the camera pushed forward or back, strong
    light variation.  Portals, lintels and the box are then seen under angles the default tables never produce."""
    from gym_miniworld_amd.batch import BatchedMiniWorld
    from gym_miniworld_amd.params import DEFAULT_PARAMS
    O = oracle_mod
    p = DEFAULT_PARAMS.copy()
    p.set("cam_pitch", 0, -25, 25)
    p.set("cam_fov_y", 60, 35, 95)
    p.set("cam_height", 1.5, 0.25, 2.6)
    p.set("cam_fwd_disp", 0, -0.2, 0.2)
    p.set("light_pos", [0, 2.5, 0], [-60, -10, -60], [60, 20, 60])
    p.set("light_color", [0.7, 0.7, 0.7], [0.1, 0.1, 0.1], [1.5, 1.5, 1.5])
    p.set("obj_color_bias", [0, 0, 0], [-0.9, -0.9, -0.9], [0.9, 0.9, 0.9])
    n = 40
    b = BatchedMiniWorld(env_id, num_envs=n, seed=321, domain_rand=True, params=p, want_depth=True)
    envs = [O.OracleEnv(task, seed=321 + i, domain_rand=True, task_args=args, params=p.to_table()) for i in range(n)]
    b.reset()
    for e in envs:
        e.reset(render=False)
    st = oracle_states(envs)
    assert_state_equal(b.get_state(), st, tag=env_id)
    assert min(s.cam_pitch for s in st) < -15 and max(s.cam_pitch for s in st) > 15 and max(s.cam_fov_y for s in st) > 85
    pos = np.array([[s.agent_pos[0], s.agent_pos[2]] for s in st])
    rng = np.random.default_rng(8)
    for rnd in range(4):
        dirs = np.array([s.agent_dir for s in st]) if rnd == 0 else rng.uniform(-np.pi, np.pi, size=n)
        b.set_agent(0, pos_xz=pos, dir=dirs)
        obs = b.render().cpu().numpy()
        dep = b.depth.cpu().numpy()[..., 0]
        for i, e in enumerate(envs):
            e.set_agent(pos[i, 0], pos[i, 1], dirs[i])
            ref, refd = e.render_obs(depth=True)
            d = obs_diff(obs[i], ref)
            assert d.max() <= 1, (env_id, rnd, i, int(d.max()), int((d > 1).sum()))
            assert np.abs(dep[i] - refd).max() <= 1e-4, (env_id, rnd, i)
    b.close()
