"""GPU: the agent's view at any size (mwb_render_view) - render_obs(frame_buffer) with another frame buffer than the observation's,
above all the 800 x 600 human view of render(mode='rgb_array') (miniworld.py:505,1160-1205,1317-1335), rendered in tiles of one
workgroup each - and with it the reference's ONLY assertion on pixel values, run_tests.py:18-23."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_reference_pixel_invariant_run_tests_18_23():
    """first_obs = env.reset(); first_render = env.render('rgb_array'); 0 < mean < 255 for both and |difference of the means| < 5"""
    from gym_miniworld_amd.env import make
    for env_id, seed in (("MiniWorld-Hallway-v0", 0), ("MiniWorld-Hallway-v0", 7), ("MiniWorld-FourRooms-v0", 3), ("MiniWorld-PickupObjs-v0", 2)):
        env = make(env_id, seed=seed)
        first_obs = env.reset()
        first_render = env.render("rgb_array")
        assert first_render.shape == (600, 800, 3) and first_render.dtype == np.uint8
        m0, m1 = first_obs.mean(), first_render.mean()
        assert m0 > 0 and m0 < 255
        assert m1 > 0 and m1 < 255
        assert abs(m0 - m1) < 5, (env_id, seed, m0, m1)
        # the big frame box-filtered 10 x 10 is the small one up to aliasing
        small = first_render.reshape(60, 10, 80, 10, 3).mean(axis=(1, 3))
        assert np.abs(small - first_obs).mean() < 6, (env_id, float(np.abs(small - first_obs).mean()))
        env.close()


@pytest.mark.parametrize("env_id,task,args,dr", [("MiniWorld-Hallway-v0", "Hallway", None, 0), ("MiniWorld-FourRooms-v0", "FourRooms", None, 1),
                                                  ("MiniWorld-PutNext-v0", "PutNext", None, 1), ("MiniWorld-YMaze-v0", "YMaze", [0, 0, 0, 0], 0),
                                                  ("MiniWorld-PickupObjs-v0", "PickupObjs", [12, 5, 0, 0], 1), ("MiniWorld-Maze-v0", "Maze", None, 0)])
def test_view_equals_oracle_at_other_sizes(oracle_mod, env_id, task, args, dr):
    """tiles of 75 x 60 pixels: sizes that are no multiple of the tile, one smaller than a tile, the reference's window size"""
    from gym_miniworld_amd.batch import BatchedMiniWorld
    O = oracle_mod
    n = 3
    b = BatchedMiniWorld(env_id, num_envs=n, seed=21, domain_rand=dr, want_depth=True)
    obs = b.reset().cpu().numpy()
    same, same_d = b.render_view(80, 60, depth=True)
    assert np.array_equal(same.cpu().numpy(), obs) and np.array_equal(same_d.cpu().numpy(), b.depth.cpu().numpy())   # tiled == untiled, bit for bit
    for (W, H) in ((200, 150), (64, 48), (800, 600) if task in ("Hallway", "PickupObjs") else (333, 211)):
        img, dep = b.render_view(W, H, depth=True)
        img, dep = img.cpu().numpy(), dep.cpu().numpy()[..., 0]
        for i in range(n if W < 800 else 1):
            e = O.OracleEnv(task, seed=21 + i, domain_rand=dr, task_args=args, obs_width=W, obs_height=H)
            e.reset(render=False)
            ref, refd = e.render_obs(depth=True)
            d = np.abs(img[i].astype(np.int16) - ref.astype(np.int16))
            assert d.max() <= 1 and np.abs(dep[i] - refd).max() <= 1e-4, (env_id, W, H, i, int(d.max()), int((d > 1).sum()))
    assert np.array_equal(b.obs.cpu().numpy(), obs)   # the observation buffers are untouched
    b.close()


@pytest.mark.parametrize("env_id,task,args,layout", [("MiniWorld-Hallway-v0", "Hallway", None, "HWC"), ("MiniWorld-FourRooms-v0", "FourRooms", None, "CWH"),
                                                     ("MiniWorld-PutNext-v0", "PutNext", None, "HWC")])
def test_observations_too_large_for_one_workgroup_are_rendered_in_tiles(oracle_mod, env_id, task, args, layout):
    """obs_width / obs_height are constructor arguments of the reference without an upper limit (miniworld.py:459-460): a 320 x 240
    observation (230 KB, more than a workgroup's LDS) is rendered in tiles by reset and by every step - frames +-1 and depth
    1e-4 m against the oracle at that size, through auto-resets, in both layouts"""
    import torch
    from gym_miniworld_amd.batch import BatchedMiniWorld
    O = oracle_mod
    n, W, H = 6, 320, 240
    b = BatchedMiniWorld(env_id, num_envs=n, seed=5, domain_rand=1, want_depth=True, obs_width=W, obs_height=H, layout=layout, max_episode_steps=7)
    envs = [O.OracleEnv(task, seed=5 + i, domain_rand=1, task_args=args, obs_width=W, obs_height=H, max_episode_steps=7) for i in range(n)]
    b.reset()
    for e in envs:
        e.reset(render=False)
    rng = np.random.default_rng(2)

    def check(tag):
        obs, dep = b.obs.cpu().numpy(), b.depth.cpu().numpy()[..., 0]
        if layout == "CWH":
            obs = obs.transpose(0, 3, 2, 1)   # [N, 3, W, H] -> [N, H, W, 3]
        for i, e in enumerate(envs):
            ref, refd = e.render_obs(depth=True)
            d = np.abs(obs[i].astype(np.int16) - ref.astype(np.int16))
            assert d.max() <= 1 and np.abs(dep[i] - refd).max() <= 1e-4, (env_id, tag, i, int(d.max()), int((d > 1).sum()))
    check("reset")
    n_done = 0
    for t in range(12):
        a = rng.integers(0, 3, n).astype(np.int32)
        b.step(torch.from_numpy(a))
        done = b.done.cpu().numpy()
        for i, e in enumerate(envs):
            _, _, dn, _ = e.step(int(a[i]))
            assert dn == bool(done[i])
            if dn:
                e.reset(render=False)
                n_done += 1
        if t % 4 == 3 or t == 6:
            check("t=%d" % t)
    assert n_done >= n
    b.close()
