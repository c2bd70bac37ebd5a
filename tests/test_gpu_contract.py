"""GPU: error behaviour of the C ABI and the bench.py / __graft_entry__ contracts the driver relies on."""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_refuses_misuse_loudly():
    import torch
    from gym_miniworld_amd import _lib
    from gym_miniworld_amd.batch import BatchedMiniWorld
    b = BatchedMiniWorld("MiniWorld-OneRoom-v0", num_envs=4)      # not seeded
    with pytest.raises(_lib.MwbError, match="mwb_seed"):
        b.reset()                                                    # the reference would seed from entropy; we refuse
    b.seed(3)
    b.reset()
    with pytest.raises(_lib.MwbError, match="stack"):
        b.stack_enable(4)                                            # HWC handle: the stack is channel-first
    with pytest.raises(_lib.MwbError):
        b.intersect(99, 0.0, 0.0)
    L = _lib.load()
    assert L.mwb_step(b.h, None, None, None) == -1 and b"null actions" in L.mwb_last_error()
    b.close()
    with pytest.raises(KeyError):
        BatchedMiniWorld("MiniWorld-RemoteBot-v0", num_envs=1)       # out of scope ids are named as such (the ZMQ robot bridge)
    h = ctypes.c_void_p()
    cfg = _lib.MwbConfig()
    cfg.abi_version, cfg.task, cfg.num_envs, cfg.obs_width, cfg.obs_height, cfg.use_default_params = 1, 3, 4, 80, 60, 1
    cfg.task_args[0], cfg.task_args[1] = 30, 30                      # 900 cells: beyond the 160 KB LDS staging limit
    assert L.mwb_create(ctypes.byref(cfg), ctypes.byref(h)) == -1


def test_bench_emits_the_contract_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2",
                          "--envs-per-gpu", "256", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["unit"] == "env-steps/s" and d["value"] > 0 and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12


def test_graph_capture_of_a_step():
    """mwb_step (incl. the fork/join onto the side stream) can be captured into a HIP graph and replayed."""
    import torch
    from gym_miniworld_amd.batch import BatchedMiniWorld
    n = 256
    a = BatchedMiniWorld("MiniWorld-FourRooms-v0", num_envs=n, seed=4)
    b = BatchedMiniWorld("MiniWorld-FourRooms-v0", num_envs=n, seed=4)
    a.reset(); b.reset()
    acts = torch.zeros(n, dtype=torch.int32, device="cuda")
    a.step(acts); b.step(acts)            # eager warm-up, both handles in lock-step
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):             # capture only records; nothing executes
        b.step(acts)
    gen = torch.Generator().manual_seed(1)
    for t in range(60):
        acts.copy_(torch.randint(0, 3, (n,), generator=gen, dtype=torch.int32))
        a.step(acts)
        g.replay()
        assert torch.equal(a.obs, b.obs) and torch.equal(a.reward64, b.reward64) and torch.equal(a.done, b.done), t
    a.close(); b.close()


def test_cost_ordered_dispatch_renders_every_env_exactly_once():
    """mwb_step dispatches the bulk render's workgroups by decreasing measured frame cost (a map rebuilt every step
    on the side stream, the cheapest envs as half-frame workgroups).  Whatever the map, every env must be rendered
    exactly once: the observations a step leaves behind equal a plain mwb_render of the same state, for every env."""
    import torch
    from gym_miniworld_amd.batch import BatchedMiniWorld
    for env_id, n, layout in (("MiniWorld-Maze-v0", 3000, "HWC"), ("MiniWorld-FourRooms-v0", 1500, "CWH")):
        b = BatchedMiniWorld(env_id, num_envs=n, seed=11, domain_rand=True, want_depth=True, layout=layout)
        b.reset()
        g = torch.Generator().manual_seed(0)
        for t in range(40):
            b.step(torch.randint(0, 3, (n,), generator=g, dtype=torch.int32))
            if t % 8 == 7:
                obs_step, dep_step = b.obs.clone(), b.depth.clone()
                b.obs.zero_()   # a workgroup that never ran would leave zeros behind
                b.render()
                assert torch.equal(b.obs, obs_step) and torch.equal(b.depth, dep_step), (env_id, t)
                b.obs.zero_()
                b.step(torch.randint(0, 3, (n,), generator=g, dtype=torch.int32))
                assert int((b.obs.reshape(n, -1).max(dim=1).values == 0).sum()) == 0, (env_id, t, "an env was not rendered")
        b.close()


def test_set_state_round_trips_and_the_rng_stream_can_be_injected(oracle_mod):
    """mwb_get_state / mwb_set_state: a snapshot (incl. the full MT19937 state) restores the batch exactly; an RNG
    state injected from another env makes the next episode of env 0 the episode that env would have had."""
    import torch
    from gym_miniworld_amd import _lib
    from gym_miniworld_amd.batch import BatchedMiniWorld
    n = 6
    b = BatchedMiniWorld("MiniWorld-FourRooms-v0", num_envs=n, seed=50)   # no domain randomisation: the room tables
    b.reset()                                                               # (not part of the snapshot) never change
    snap = b.get_state(rng_state=True)
    assert snap["rng_state"].shape == (n, 625) and np.array_equal(snap["rng_state"][:, 624], snap["rng_pos"])
    assert np.array_equal(snap["rng_state"][:, :624].astype(np.uint64).sum(axis=1).astype(np.uint32), snap["rng_keysum"])
    g = torch.Generator().manual_seed(3)
    acts = [torch.randint(0, 3, (n,), generator=g, dtype=torch.int32) for _ in range(30)]
    for a in acts:
        b.step(a)
    after = b.get_state(rng_state=True)
    obs_after = b.obs.clone()
    keys = ("agent_pos", "agent_dir", "boxes_pos", "boxes_dir", "boxes_color", "boxes_size", "cam", "sky_color", "light_pos",
            "light_color", "light_ambient", "step_count", "goal_idx", "episode_count", "task_step_count", "goal_dist", "rng_state")
    b.set_state(0, **{k: snap[k] for k in keys})
    again = b.get_state(rng_state=True)
    for k in keys:
        assert np.array_equal(again[k], snap[k]), k
    for a in acts:   # same actions from the restored snapshot: same trajectory (nobody finished an episode in 30 steps,
        b.step(a)    # or if somebody did, the restored RNG regenerates the same world)
    redo = b.get_state(rng_state=True)
    for k in keys:
        assert np.array_equal(redo[k], after[k]), k
    assert torch.equal(b.obs, obs_after)
    # env 0 continues env 3's stream: its next reset builds env 3's next world
    b.set_state(0, rng_state=redo["rng_state"][3:4])
    b.reset()
    st = b.get_state(rng_state=True)
    assert np.array_equal(st["agent_pos"][0], st["agent_pos"][3]) and np.array_equal(st["boxes_pos"][0], st["boxes_pos"][3])
    assert np.array_equal(st["rng_state"][0], st["rng_state"][3])
    with pytest.raises(_lib.MwbError, match="position"):
        bad = redo["rng_state"][:1].copy(); bad[0, 624] = 700
        b.set_state(0, rng_state=bad)
    with pytest.raises(KeyError):
        b.set_state(0, n_rooms=[1])
    b.close()


def test_intersect_skips_the_querying_entity_and_sees_the_agent(oracle_mod):
    """MiniWorldEnv.intersect(ent, pos, radius), miniworld.py:933-959: walls first, then every entity but `ent`
    in list order - the agent is an obstacle for a box, a box is not an obstacle for itself."""
    from gym_miniworld_amd.batch import BatchedMiniWorld
    from gym_miniworld_amd.env import MiniWorldEnv
    b = BatchedMiniWorld("MiniWorld-TMazeTwoBoxDynamic-v0", num_envs=2, seed=3)
    b.reset()
    st = b.get_state()
    red, blue, ag = st["boxes_pos"][0, 0], st["boxes_pos"][0, 1], st["agent_pos"][0]
    br = float(np.sqrt(2 * 0.8 * 0.8) / 2)
    assert b.intersect(0, red[0], red[2], br, ent=0) == 0            # the red box at its own position: itself is skipped
    assert b.intersect(0, red[0], red[2], br, ent=1) == 2            # ... the blue box asking about the same spot hits red
    assert b.intersect(0, red[0], red[2], br, ent=-1) == 2
    assert b.intersect(0, ag[0] + 0.1, ag[2], br, ent=0) == 4        # a box near the agent: the agent (entity 2) blocks
    assert b.intersect(0, ag[0] + 0.1, ag[2], 0.4) == 0              # the agent asking: itself is skipped (default ent)
    assert b.intersect(0, blue[0], blue[2] - 0.3, 0.4) == 3
    b.close()
    env = MiniWorldEnv("MiniWorld-TMazeTwoBoxDynamic-v0", seed=3)
    env.reset()
    assert env.intersect(env.red_box, env.red_box.pos, env.red_box.radius) is None
    assert env.intersect(env.blue_box, env.red_box.pos, env.red_box.radius) is env.red_box
    assert env.intersect(env.red_box, env.agent.pos, env.red_box.radius) is env.agent
    assert env.intersect(env.agent, env.agent.pos, env.agent.radius) is None
    assert env.intersect(env.agent, np.array([7.9, 0, 3.0]), 0.4) is True
    env.close()


@pytest.mark.parametrize("env_id,task,targs", [("MiniWorld-TMazeTwoBoxDynamic-v0", "TMazeTwoBox", [0, 0, 0, 100]),
                                               ("MiniWorld-SimToRealPush-v0", "SimToRealPush", None)])
def test_two_box_tasks_render_large_observations(oracle_mod, env_id, task, targs):
    """200x150 needs more than the default 64 KB of dynamic LDS (the frame is assembled in LDS): the two-box render
    kernels must opt in like the one-box ones (round-1 advisor finding); 640x480 goes through the tiled path."""
    from gym_miniworld_amd.batch import BatchedMiniWorld
    from gym_miniworld_amd.params import sim_to_real_params
    O = oracle_mod
    prm = sim_to_real_params(push=True).to_table() if task == "SimToRealPush" else None
    n = 3
    b = BatchedMiniWorld(env_id, num_envs=n, seed=9, obs_width=200, obs_height=150, want_depth=True, domain_rand=True)
    obs = b.reset().cpu().numpy()
    dep = b.depth.cpu().numpy()[..., 0]
    for i in range(n):
        e = O.OracleEnv(task, seed=9 + i, domain_rand=True, task_args=targs, params=prm, obs_width=200, obs_height=150)
        e.reset(render=False)
        ref, refd = e.render_obs(depth=True)
        d = np.abs(obs[i].astype(np.int16) - ref.astype(np.int16))
        assert d.max() <= 1, (env_id, i, int(d.max()))
        assert np.abs(dep[i] - refd).max() <= 1e-4
    b.close()
    # 640 x 480 fits neither one workgroup's LDS nor its 16-bit pixel queue: refused until round 3, rendered in tiles since
    b = BatchedMiniWorld(env_id, num_envs=1, seed=9, obs_width=640, obs_height=480, want_depth=True, domain_rand=True)
    obs = b.reset().cpu().numpy()
    dep = b.depth.cpu().numpy()[..., 0]
    e = O.OracleEnv(task, seed=9, domain_rand=True, task_args=targs, params=prm, obs_width=640, obs_height=480)
    e.reset(render=False)
    ref, refd = e.render_obs(depth=True)
    assert np.abs(obs[0].astype(np.int16) - ref.astype(np.int16)).max() <= 1 and np.abs(dep[0] - refd).max() <= 1e-4
    b.close()


def test_shard_exchange_single_gpu_self_test():
    """world 1 on the GPU: the staging path of the per-step exchange (obs + aux pack from a BatchedMiniWorld, consumed
    one round behind) and the action scatter reproduce the env's own outputs; bench.py --force-gather drives the same path."""
    import torch
    from gym_miniworld_amd.batch import BatchedMiniWorld
    from gym_miniworld_amd.distributed import ShardExchange, unpack_aux
    n = 64
    b = BatchedMiniWorld("MiniWorld-TMazeTwoBoxDynamicFeatures100K-v0", num_envs=n, seed=5, layout="CWH")
    b.reset()
    g = ShardExchange(tuple(b.obs.shape), b.obs.dtype, b.device, 1, rank=0)
    gen = torch.Generator().manual_seed(0)
    prev = None
    for t in range(40):
        a_all = torch.randint(0, 3, (n, 1), generator=gen)
        a = g.scatter_actions(a_all)
        assert a.dtype == torch.int32 and a.device.type == "cuda" and a.cpu().tolist() == a_all.reshape(-1).tolist()
        b.step(a)
        snap = (b.obs.clone(), b.reward64.clone(), b.done.clone(), b.ep_steps.clone(), b.feature.clone(), b.goal_pos.clone())
        g.push(b.obs, env=b)
        if prev is not None:
            po, pa = g.previous()
            u = unpack_aux(pa)
            assert torch.equal(po, prev[0]) and torch.equal(u["reward"], prev[1]) and torch.equal(u["done"], prev[2] != 0)
            assert torch.equal(u["ep_steps"], prev[3]) and torch.equal(u["feature"].float(), prev[4]) and torch.equal(u["goal_pos"], prev[5])
        lo, la = g.latest()
        assert torch.equal(lo, snap[0]) and torch.equal(la[:, 0], snap[1])
        prev = snap
    g.drain()
    b.close()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--envs-per-gpu", "256",
                          "--no-cpu-baseline", "--no-vecenv", "--force-gather"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.strip().splitlines() if l.startswith("{")][0])
    assert "exchange" in d["config"]["parallelism"] and d["value"] > 0


def test_vecenv_leg_of_the_bench_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "2", "--envs-per-gpu", "512",
                          "--no-cpu-baseline", "--workload", "tmaze_features8192"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.strip().splitlines() if l.startswith("{")][0])
    v = d["vecenv"]
    assert "error" not in v, v
    assert v["vecenv_u8"]["value"] > 0 and v["make_vec_envs_f32_stack4"]["value"] > 0 and v["vecenv_u8"]["infos"] == "LazyInfos"


def test_vecenv_graph_replay_equals_eager_steps():
    """MiniWorldVecEnv(graph=True) (optional: off by default) replays the captured step; results equal the eager path's"""
    import torch
    from gym_miniworld_amd.vec_env import MiniWorldVecEnv
    n = 64
    a = MiniWorldVecEnv("MiniWorld-FourRooms-v0", n, seed=9, to_float=False, graph=True)
    b = MiniWorldVecEnv("MiniWorld-FourRooms-v0", n, seed=9, to_float=False, graph=False)
    assert torch.equal(a.reset(), b.reset())
    g = torch.Generator().manual_seed(1)
    for t in range(12):
        act = torch.randint(0, 3, (n, 1), generator=g)
        oa, ra, da, _ = a.step(act)
        ob, rb, db, _ = b.step(act)
        assert torch.equal(oa, ob) and torch.equal(ra, rb) and np.array_equal(da, db), t
    assert a._graph is not None and b._graph is None
    a.close(); b.close()


def test_round2_advisor_findings():
    """int64 actions outside int32 are unknown actions (never aliased to a real one); mwb_set_state refuses NaN / empty boxes;
    a fused frame stack cannot be enabled once an observation exists."""
    import torch
    from gym_miniworld_amd import _lib
    from gym_miniworld_amd.batch import BatchedMiniWorld
    n = 6
    b = BatchedMiniWorld("MiniWorld-OneRoom-v0", num_envs=n, seed=5, layout="CWH")
    b.reset()
    s0 = b.get_state()
    # 2^32 + 2 has the low word of move_forward, -1 / 2^40 / int64 min are no actions either: nobody moves or turns
    weird = torch.tensor([2 ** 32 + 2, 2 ** 32 + 0, -1, 2 ** 40 + 1, -2 ** 63, 2 ** 31 + 2], dtype=torch.int64, device=b.device)
    b.step(weird)
    s1 = b.get_state()
    assert np.array_equal(s0["agent_pos"], s1["agent_pos"]) and np.array_equal(s0["agent_dir"], s1["agent_dir"])
    assert list(s1["step_count"]) == [1] * n                                  # the step itself counts (miniworld.py:663)
    b.step(torch.full((n,), 2, dtype=torch.int64, device=b.device))           # a proper LongTensor action still moves
    s2 = b.get_state()
    assert not np.array_equal(s1["agent_pos"], s2["agent_pos"])
    for bad in (dict(agent_pos=np.full((1, 3), np.nan)), dict(agent_dir=np.array([np.inf])), dict(boxes_size=np.zeros((1, 1))),
                dict(boxes_size=np.full((1, 1), -0.8)), dict(cam=np.array([[1.5, 0.0, np.nan, 60.0]]))):
        with pytest.raises(_lib.MwbError, match="mwb_set_state"):
            b.set_state(0, **bad)
    assert np.array_equal(b.get_state()["agent_pos"], s2["agent_pos"])        # a refused call changed nothing
    with pytest.raises(_lib.MwbError, match="before the first"):
        b.stack_enable(4, fused=True)                                         # an observation exists: the window would lack it
    b.stack_enable(4, fused=False)                                            # the non-fused forms rebuild theirs from the obs buffer
    b.close()
    import gym_miniworld_amd.distributed as D
    assert not hasattr(D, "ObsGatherer")                                      # the round-1 alias was not API compatible: gone
