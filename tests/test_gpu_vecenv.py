"""GPU: the VecEnv / Gym front-ends reproduce the reference harness' batch semantics
(vec_env/subproc_vec_env.py:5-33,58-75; envs.py:96-165) on top of the HIP path, and the
reference's own smoke tests (run_tests.py) hold."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_vecenv_matches_per_env_oracle_with_autoreset(oracle_mod):
    import torch
    from gym_miniworld_amd.vec_env import MiniWorldVecEnv
    O = oracle_mod
    n, seed = 6, 40
    v = MiniWorldVecEnv("MiniWorld-Hallway-v0", n, seed=seed, transpose=True, to_float=True, torch_api=True)
    assert v.observation_space.shape == (3, 80, 60) and v.action_space.n == 3
    refs = [O.OracleEnv("Hallway", seed=seed + i) for i in range(n)]   # env i seeded seed + i, envs.py:36
    obs = v.reset()
    assert obs.dtype == torch.float32 and tuple(obs.shape) == (n, 3, 80, 60)
    for i, r in enumerate(refs):
        d = np.abs(obs[i].cpu().numpy() - r.reset().transpose(2, 1, 0).astype(np.float32))
        assert d.max() <= 1
    rng = np.random.default_rng(0)
    seen_done = False
    for t in range(260):
        a = rng.choice(3, size=(n, 1), p=[0.15, 0.15, 0.7])
        obs, rew, done, infos = v.step(torch.from_numpy(a))
        assert rew.shape == (n, 1) and rew.dtype == torch.float32 and rew.device.type == "cpu"
        assert done.dtype == bool and len(infos) == n
        for i, r in enumerate(refs):
            _, rr, dd, _ = r.step(int(a[i, 0]))
            assert np.float32(rr) == rew[i, 0].item() and dd == done[i]
            if dd:   # worker: `if done: ob = env.reset()` - the returned obs is the NEW episode's first
                seen_done = True
                first = r.reset().transpose(2, 1, 0).astype(np.float32)
                assert np.abs(obs[i].cpu().numpy() - first).max() <= 1
    assert seen_done
    v.close()


@pytest.mark.parametrize("env_id", ["MiniWorld-OneRoomS6-v0", "MiniWorld-CollectHealth-v0", "MiniWorld-PickupObjs-v0"])
def test_frame_stack_semantics(env_id):
    """VecPyTorchFrameStack (envs.py:135-165) restated in numpy on the same observation stream (the fused window: also the entity
    tasks' render kernel writes it)."""
    import torch
    from gym_miniworld_amd.vec_env import MiniWorldVecEnv
    n = 5
    kw = {"max_episode_steps": 40} if "PickupObjs" in env_id else {}
    v = MiniWorldVecEnv(env_id, n, seed=3, frame_stack=4, **kw)
    plain = MiniWorldVecEnv(env_id, n, seed=3, frame_stack=0, **kw)
    assert v.observation_space.shape == (12, 80, 60)
    st = v.reset().cpu().numpy()
    ob = plain.reset().cpu().numpy()
    ref = np.zeros((n, 12, 80, 60), np.float32)
    ref[:, -3:] = ob
    assert np.array_equal(st, ref)
    g = torch.Generator().manual_seed(1)
    any_done = False
    for t in range(130):
        a = torch.randint(0, 3, (n, 1), generator=g)
        st, _, done, _ = v.step(a)
        ob, _, done2, _ = plain.step(a)
        assert np.array_equal(done, done2)
        ref[:, :-3] = ref[:, 3:].copy()
        ref[done] = 0
        ref[:, -3:] = ob.cpu().numpy()
        any_done |= bool(done.any())
        assert np.array_equal(st.cpu().numpy(), ref)
    assert any_done
    v.close(); plain.close()


def test_mask_and_feature_info():
    import torch
    from gym_miniworld_amd.vec_env import make_vec_envs
    v = make_vec_envs("MiniWorld-FourRooms-v0", 1, 4, device="cuda:0")   # the fork's main.py:66 call shape
    obs = v.reset()
    assert tuple(obs.shape) == (4, 12, 80, 60) and obs.device.type == "cuda"
    mask = np.array([0, 1, 0, 1])
    obs, rew, done, infos = v.step(torch.full((4, 1), 2, dtype=torch.long), mask)
    assert rew[1, 0] == -99 and rew[3, 0] == -99 and not done[1] and not done[3]
    assert all("feature" in i and len(i["feature"]) == 2 for i in infos)   # main.py:614-619
    v.close()


def test_gym_view_reference_smoke_tests():
    """run_tests.py:11-28 and 51-59 through the single-env Gym view."""
    from gym_miniworld_amd.env import make
    env = make("MiniWorld-Hallway-v0", seed=0)
    first = env.reset()
    for _ in range(10):
        obs, _, _, _ = env.step(0)
    assert 0 < first.mean() < 255
    assert first.shape == env.observation_space.shape == obs.shape
    d = env.render_depth()
    assert d.shape == (60, 80, 1) and d.min() > 0
    env.close()
    env = make("MiniWorld-OneRoom-v0", seed=5)
    for _ in range(6):
        env.reset()
        room = env.rooms[0]
        assert not env.intersect(env.agent, env.agent.pos, env.agent.radius)   # run_tests.py:69
        for _ in range(30):
            _, _, done, _ = env.step(env.actions.move_forward)
            x, _, z = env.agent.pos
            assert room.min_x <= x <= room.max_x and room.min_z <= z <= room.max_z
            if done:
                break
    env.close()


def test_gym_view_returns_terminal_obs_without_autoreset(oracle_mod):
    from gym_miniworld_amd.env import make
    O = oracle_mod
    env = make("MiniWorld-OneRoomS6-v0", seed=9)
    ref = O.OracleEnv("OneRoom", seed=9, task_args=[6], max_episode_steps=100)
    env.reset(); ref.reset(render=False)
    for t in range(100):
        obs, r, d, _ = env.step(1)
        _, rr, dd, _ = ref.step(1)
        assert (r, d) == (rr, dd)
    assert d and env.step_count == 100   # timeout at max_episode_steps, state kept (no reset happened)
    assert np.abs(obs.astype(int) - ref.render_obs().astype(int)).max() <= 1
    env.close()


def test_fused_stack_u8_and_bandwidth_shape():
    """The uint8 variant of the fused stack equals the float one; the stack tensor is library-owned."""
    import torch
    from gym_miniworld_amd.vec_env import MiniWorldVecEnv
    n = 64
    f = MiniWorldVecEnv("MiniWorld-OneRoomS6-v0", n, seed=8, frame_stack=4, to_float=True)
    u = MiniWorldVecEnv("MiniWorld-OneRoomS6-v0", n, seed=8, frame_stack=4, to_float=False)
    assert torch.equal(f.reset(), u.reset().float())
    g = torch.Generator().manual_seed(2)
    for t in range(110):
        a = torch.randint(0, 3, (n, 1), generator=g)
        sf, rf, df, _ = f.step(a)
        su, ru, du, _ = u.step(a)
        assert su.dtype == torch.uint8 and torch.equal(sf, su.float()) and np.array_equal(df, du)
    f.close(); u.close()


def test_domain_rand_toggled_after_construction(oracle_mod):
    """run_tests.py:64-66 sets `env.domain_rand = True` on a constructed env; from the next reset on it behaves
    like an env constructed with domain randomisation (draw order, textures, per-step parameter draws)."""
    from gym_miniworld_amd.env import make
    O = oracle_mod
    env = make("MiniWorld-FourRooms-v0", seed=12)
    assert env.domain_rand is False
    env.domain_rand = True
    assert env.domain_rand is True
    ref = O.OracleEnv("FourRooms", seed=12, domain_rand=True)
    obs = env.reset()
    ref.reset(render=False)
    assert np.array_equal(env.agent.pos, np.array(ref.state().agent_pos)) and list(env.sky_color) == list(ref.state().sky_color)
    assert np.abs(obs.astype(int) - ref.render_obs().astype(int)).max() <= 1
    for a in (2, 2, 0, 2, 1, 2, 2):
        _, r, d, _ = env.step(a)
        _, rr, dd, _ = ref.step(a)
        assert (r, d) == (rr, dd)
    s = ref.state()
    assert env._b.get_state()["rng_pos"][0] == s.rng_pos   # three draws per step were consumed
    env.domain_rand = False
    env.reset(); ref2 = None
    assert list(env.sky_color) == [0.25, 0.82, 1.0]   # defaults again (params.py:111)
    env.close()


def test_every_covered_env_id_loads_and_steps_like_run_tests():
    """run_tests.py:61-79: for each registered id (here: the covered ones) - gym.make, domain_rand = True, random
    restarts, the spawn never intersects anything, random actions, reset on done."""
    from gym_miniworld_amd.batch import ENV_SPECS
    from gym_miniworld_amd.env import make
    rng = np.random.default_rng(0)
    for env_id in sorted(ENV_SPECS):
        env = make(env_id, seed=3)
        env.domain_rand = True
        for _ in range(4):
            obs = env.reset()
            if isinstance(obs, dict):   # Sign: {"obs": ..., "goal": ...} (sign.py:130-133)
                obs = obs["obs"]
            assert obs.shape == env.observation_space.shape and 0 < obs.mean() < 255, env_id
            assert not env.intersect(env.agent, env.agent.pos, env.agent.radius), env_id
            for _ in range(20):
                obs, reward, done, info = env.step(int(rng.integers(0, env.action_space.n)))
                if done:
                    env.reset()
        env.close()


def test_held_stack_view_lifetime():
    """The sliding-window stack hands out a new VIEW every step.  What an older view is worth: only until the next
    step in general - an env that ends later gets its history planes zeroed in place - and, for envs that keep running,
    MWB_STACK_SLACK_FRAMES + 1 - nstack = 5 further steps from ANY window position (then the wrap copy or a new frame gets there)."""
    import torch
    from gym_miniworld_amd import _lib
    from gym_miniworld_amd.vec_env import MiniWorldVecEnv
    n, nstack = 4, 4
    keep = _lib.STACK_SLACK_FRAMES + 1 - nstack
    assert keep == 5
    v = MiniWorldVecEnv("MiniWorld-OneRoomS6-v0", n, seed=11, frame_stack=nstack, max_episode_steps=10 ** 6)
    held = [(0, v.reset(), None)]
    held[0] = (0, held[0][1], held[0][1].clone())
    turn = torch.zeros((n, 1), dtype=torch.int64)                     # turning only: nobody reaches the box, no episode ends
    clobbered_at = set()
    for t in range(1, 40):
        st, _, done, _ = v.step(turn)
        assert not done.any()
        for (t0, view, copy) in held:
            same = bool(torch.equal(view, copy))
            if t - t0 <= keep:
                assert same, (t0, t)                                  # every window position keeps the promise
            elif not same:
                clobbered_at.add(t - t0)
        held = [h for h in held if t - h[0] <= keep + 3] + [(t, st, st.clone())]
    assert clobbered_at and min(clobbered_at) == keep + 1             # and the bound is tight: some view dies on step 6
    v.close()
