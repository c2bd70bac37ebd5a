"""GPU: the tasks with a general entity list - PickupObjs, RoomObjs, CollectHealth, ThreeRooms, Sign, Sidewalk, WallGap (mesh entities, image /
text frames, entities that leave the list or re-enter it at its end) - through the C ABI against the oracle: state bit for bit
(entity kinds, dimensions incl. the NumPy-2 float32 radii, poses, LIST ORDER, counters, RNG), rewards / dones / step counts at every
step, frames within +-1 LSB and depth within 1e-4 m incl. the step's own frame of a pick-up (the object still in the agent's hands)."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# env id -> (oracle task, task_args, params)
TASKS = {"MiniWorld-PickupObjs-v0": ("PickupObjs", [12, 5, 0, 0]), "MiniWorld-RoomObjs-v0": ("RoomObjs", [10, 0, 0, 0]),
         "MiniWorld-CollectHealth-v0": ("CollectHealth", [16, 0, 0, 0]), "MiniWorld-ThreeRooms-v0": ("ThreeRooms", None), "MiniWorld-Sign-v0": ("Sign", [10, 0, 0, 0]),
         "MiniWorld-Sidewalk-v0": ("Sidewalk", None), "MiniWorld-WallGap-v0": ("WallGap", None)}


def obs_diff(a, b):
    return np.abs(a.astype(np.int16) - b.astype(np.int16))


def make_pair(O, env_id, n, seed, dr, **kw):
    from gym_miniworld_amd.batch import BatchedMiniWorld, ENV_SPECS
    task, args = TASKS[env_id]
    if task == "Sign":
        dr = 0
    b = BatchedMiniWorld(env_id, num_envs=n, seed=seed, domain_rand=dr, want_depth=True, **kw)
    prm = ENV_SPECS[env_id][3]
    envs = [O.OracleEnv(task, seed=seed + i, domain_rand=dr, task_args=args, params=prm().to_table() if prm else None) for i in range(n)]
    return b, envs


def assert_state_equal(b, st, envs, tag=""):
    ss = [e.state() for e in envs]
    E = b.n_boxes
    arr = lambda f: np.array([f(s) for s in ss])   # noqa: E731
    assert np.array_equal(st["agent_pos"], arr(lambda s: list(s.agent_pos))) and np.array_equal(st["agent_dir"], arr(lambda s: s.agent_dir)), (tag, "pose")
    assert np.array_equal(st["ent_kind"], arr(lambda s: list(s.ents_kind[:E]))) and np.array_equal(st["ent_geom"] * (st["ent_kind"] == 1), arr(lambda s: [g if k == 1 else 0 for g, k in zip(s.ents_mesh[:E], s.ents_kind[:E])])), (tag, "kinds")
    assert np.array_equal(st["ent_alive"], arr(lambda s: list(s.ents_alive[:E]))) and np.array_equal(st["ent_static"], arr(lambda s: list(s.ents_static[:E]))), (tag, "flags")
    assert np.array_equal(st["ent_radius"], arr(lambda s: list(s.ents_radius[:E]))) and np.array_equal(st["ent_rad_f32"], arr(lambda s: list(s.ents_rad_f32[:E]))), (tag, "radius")
    assert np.array_equal(st["ent_height"], arr(lambda s: list(s.ents_height[:E]))) and np.array_equal(st["ent_scale"], arr(lambda s: list(s.ents_scale[:E]))), (tag, "dims")
    assert np.array_equal(st["ent_order"], arr(lambda s: list(s.order[:E + 1]))), (tag, "order", st["ent_order"][0], list(ss[0].order[:E + 1]))
    alive = st["ent_alive"].astype(bool)
    op, od, oc = arr(lambda s: np.array(s.boxes_pos)[:E]), arr(lambda s: list(s.boxes_dir)[:E]), arr(lambda s: np.array(s.boxes_color)[:E])
    assert np.array_equal(st["boxes_pos"][alive], op[alive]) and np.array_equal(st["boxes_dir"][alive], od[alive]), (tag, "entity poses")
    solid = st["ent_kind"] < 2   # Box.color_vec / the mesh's Kd; frames have no colour
    assert np.array_equal(st["boxes_color"][solid], oc[solid]), (tag, "colours")
    assert np.array_equal(st["carrying"], arr(lambda s: s.carrying)), (tag, "carrying")
    assert np.array_equal(st["task_f"], arr(lambda s: s.health)) and np.array_equal(st["task_i"], arr(lambda s: s.num_picked)), (tag, "counters")
    assert np.array_equal(st["cam"], arr(lambda s: [s.cam_height, s.cam_fwd_disp, s.cam_pitch, s.cam_fov_y])), (tag, "cam")
    for k in ("sky_color", "light_pos", "light_color", "light_ambient"):
        assert np.array_equal(st[k], arr(lambda s: list(getattr(s, k)))), (tag, k)
    assert np.array_equal(st["step_count"], arr(lambda s: s.step_count)) and np.array_equal(st["rng_pos"], arr(lambda s: s.rng_pos)), (tag, "step / rng pos")
    assert np.array_equal(st["rng_keysum"], arr(lambda s: s.rng_keysum).astype(np.uint32)), (tag, "rng")
    if b.task == "Sign":
        assert np.array_equal(st["text_tex"], arr(lambda s: list(s.ents_tex[6]))), (tag, "text")


def policy_action(s, b, rng, n_act):
    """walk up to the nearest thing that can be picked up and pick it up (the fixture generator's `collect`); some noise"""
    if rng.random() < 0.08:
        return int(rng.integers(0, n_act))
    ax, az, ad = s.agent_pos[0], s.agent_pos[2], s.agent_dir
    best, bd = None, 1e9
    for i in range(s.n_boxes):
        if s.ents_alive[i] and not s.ents_static[i] and s.ents_kind[i] < 2:
            dd = math.hypot(s.boxes_pos[i][0] - ax, s.boxes_pos[i][2] - az)
            if dd < bd:
                best, bd = i, dd
    if best is None:
        return int(rng.integers(0, n_act))
    t = s.boxes_pos[best]
    want = math.atan2(-(t[2] - az), t[0] - ax)
    diff = (want - ad + math.pi) % (2 * math.pi) - math.pi
    if abs(diff) > math.radians(9):
        return 0 if diff > 0 else 1
    if n_act > 4 and bd < 1.5 * s.agent_radius + 1.2 * s.agent_radius + s.ents_radius[best] - 0.05:
        return 4
    return 2


@pytest.mark.parametrize("env_id", list(TASKS))
@pytest.mark.parametrize("dr", [0, 1])
def test_reset_state_and_first_obs(oracle_mod, env_id, dr):
    O = oracle_mod
    n = 12
    b, envs = make_pair(O, env_id, n, seed=900, dr=dr)
    obs = b.reset().cpu().numpy()
    dep = b.depth.cpu().numpy()[..., 0]
    for e in envs:
        e.reset(render=False)
    assert_state_equal(b, b.get_state(), envs, "reset")
    rooms, segs = b.get_geometry(2)
    assert np.array_equal(segs, envs[2].geometry()["wall_segs"])
    for i, e in enumerate(envs):
        ref, refd = e.render_obs(depth=True)
        d = obs_diff(obs[i], ref)
        assert d.max() <= 1 and np.abs(dep[i] - refd).max() <= 1e-4, (env_id, dr, i, int(d.max()), float(np.abs(dep[i] - refd).max()))
    b.reset()
    for e in envs:
        e.reset(render=False)
    assert_state_equal(b, b.get_state(), envs, "2nd reset")
    b.close()


@pytest.mark.parametrize("env_id,dr,policy", [("MiniWorld-PickupObjs-v0", 0, "collect"), ("MiniWorld-PickupObjs-v0", 1, "collect"),
                                              ("MiniWorld-RoomObjs-v0", 1, "collect"), ("MiniWorld-CollectHealth-v0", 0, "collect"),
                                              ("MiniWorld-CollectHealth-v0", 1, "random"), ("MiniWorld-Sign-v0", 0, "collect"),
                                              ("MiniWorld-Sidewalk-v0", 1, "collect"), ("MiniWorld-WallGap-v0", 0, "collect"),
                                              ("MiniWorld-PickupObjs-v0", 1, "random"), ("MiniWorld-ThreeRooms-v0", 0, "collect"),
                                              ("MiniWorld-ThreeRooms-v0", 1, "random")])
def test_rollout_exact_and_step_frames(oracle_mod, env_id, dr, policy):
    import torch
    O = oracle_mod
    n, steps = 10, 420
    b, envs = make_pair(O, env_id, n, seed=31, dr=dr)
    b.reset()
    for e in envs:
        e.reset(render=False)
    rng = np.random.default_rng(5)
    n_done = n_removed = n_carry = n_reward = 0
    for t in range(steps):
        if policy == "collect":
            a = np.array([policy_action(e.state(), b, rng, b.n_actions) for e in envs], dtype=np.int32)
        else:
            a = rng.integers(0, b.n_actions, size=n).astype(np.int32)
        b.step(torch.from_numpy(a))
        rew, done, eps = b.reward64.cpu().numpy(), b.done.cpu().numpy(), b.ep_steps.cpu().numpy()
        check_frame = t % 15 == 14 or t == steps - 1
        obs = dep = None
        if check_frame:
            obs, dep = b.obs.cpu().numpy(), b.depth.cpu().numpy()[..., 0]
        if b.has_health:
            health = b.feature.cpu().numpy()[:, 0]
        for i, e in enumerate(envs):
            before = e.state()
            _, r, d, _ = e.step(int(a[i]))
            s = e.state()
            assert r == rew[i] and d == bool(done[i]) and s.step_count == eps[i], (env_id, dr, t, i, r, rew[i], d, done[i])
            if b.has_health:
                assert health[i] == s.health, (t, i)
            changed = list(before.order) != list(s.order)
            n_removed += changed
            n_carry += s.carrying >= 0
            n_reward += r > 0
            if d:
                e.reset(render=False)
                n_done += 1
            if check_frame or changed:   # the step's own frame (an entity the rule just removed / respawned is still where the frame saw it)
                if obs is None:
                    obs, dep = b.obs.cpu().numpy(), b.depth.cpu().numpy()[..., 0]
                ref, refd = e.render_obs(depth=True, step_frame=not d)
                df = obs_diff(obs[i], ref)
                assert df.max() <= 1 and np.abs(dep[i] - refd).max() <= 1e-4, (env_id, dr, t, i, int(df.max()), int((df > 1).sum()))
        if t % 10 == 9 or t == steps - 1:
            assert_state_equal(b, b.get_state(), envs, "%s dr%d t=%d" % (env_id, dr, t))
    assert n_done > 0 or "RoomObjs" in env_id
    if "PickupObjs" in env_id and policy == "collect":
        assert n_removed > 5 and n_reward > 5
    if "CollectHealth" in env_id and policy == "collect":
        assert n_removed > 5
    if "RoomObjs" in env_id:
        assert n_carry > 20
    b.check()
    b.close()


@pytest.mark.parametrize("env_id", list(TASKS))
def test_random_views_and_fast_path(oracle_mod, env_id, monkeypatch):
    """random poses all over the world (entities close up, from all sides, meshes overlapping each other on screen): frames
    +-1 / depth 1e-4 against the oracle, and the interior-pixel fast path equal to the 8-sample path bit for bit"""
    O = oracle_mod
    n = 16
    fast, envs = make_pair(O, env_id, n, seed=77, dr=1)
    fast.reset()
    for e in envs:
        e.reset(render=False)
    monkeypatch.setenv("MWB_DEBUG", "1")
    slow, _ = make_pair(O, env_id, n, seed=77, dr=1)
    monkeypatch.delenv("MWB_DEBUG")
    slow.reset()
    rng = np.random.default_rng(3)
    worst = 0
    for rep in range(6):
        st = fast.get_state()
        pos, dirs = np.zeros((n, 2)), rng.uniform(-math.pi, math.pi, n)
        for i, e in enumerate(envs):
            s = e.state()
            g = e.geometry()
            # a point inside the first room, looking roughly at a random entity half of the time
            o = g["outline"][int(rng.integers(0, 3)) if "ThreeRooms" in env_id else 0]   # ThreeRooms: any of the three rooms, through both openings
            lo, hi = o.min(axis=0) + 0.45, o.max(axis=0) - 0.45
            pos[i] = rng.uniform(lo, hi)
            if rng.random() < 0.6:
                k = int(rng.integers(0, s.n_boxes))
                t = s.boxes_pos[k]
                dirs[i] = math.atan2(-(t[2] - pos[i, 1]), t[0] - pos[i, 0]) + rng.normal(0, 0.25)
            e.set_agent(pos[i, 0], pos[i, 1], dirs[i])
        for bb in (fast, slow):
            bb.set_agent(0, pos_xz=pos, dir=dirs)
            bb.render()
        obs, dep = fast.obs.cpu().numpy(), fast.depth.cpu().numpy()[..., 0]
        assert np.array_equal(obs, slow.obs.cpu().numpy()) and np.array_equal(dep, slow.depth.cpu().numpy()[..., 0]), (env_id, rep)
        for i, e in enumerate(envs):
            ref, refd = e.render_obs(depth=True)
            df = obs_diff(obs[i], ref)
            worst = max(worst, int(df.max()))
            assert df.max() <= 1 and np.abs(dep[i] - refd).max() <= 1e-4, (env_id, rep, i, int(df.max()), int((df > 1).sum()))
        del st
    fast.close(); slow.close()


@pytest.mark.parametrize("env_id", ["MiniWorld-CollectHealth-v0", "MiniWorld-Sidewalk-v0"])
def test_pair_list_overflow_path_equals_the_list_path(env_id, monkeypatch):
    """a round's (ray, mesh) pairs beyond the LDS list's capacity are walked on the spot by their own lane: with the capacity cut to 8
    pairs (MWB_EXP bit 5) nearly every pair goes that way - frames and depth must not change by a bit"""
    import torch
    from gym_miniworld_amd.batch import BatchedMiniWorld
    n = 12
    a = BatchedMiniWorld(env_id, num_envs=n, seed=41, domain_rand=1, want_depth=True)
    monkeypatch.setenv("MWB_EXP", "32")
    b = BatchedMiniWorld(env_id, num_envs=n, seed=41, domain_rand=1, want_depth=True)
    monkeypatch.delenv("MWB_EXP")
    a.reset(); b.reset()
    rng = np.random.default_rng(9)
    for t in range(40):
        act = torch.from_numpy(rng.integers(0, 3, n).astype(np.int32))
        a.step(act); b.step(act)
        if t % 5 == 4:
            assert torch.equal(a.obs, b.obs) and torch.equal(a.depth, b.depth), (env_id, t)
    a.close(); b.close()
