"""GPU leg of the reference-image pin: the HIP renderer, through the C ABI (mwb_set_state + mwb_render),
against the frames the reference holds under /root/reference/images (committed as tests/golden/refimg_*.npz;
statistics and tolerances in tests/refimg_stats.py; the CPU leg is tests/test_refimage_pin.py)."""
import math
import os

import numpy as np
import pytest

import refimg_stats as RS

pytestmark = pytest.mark.gpu


def _render(fx, env_id, W, H, box=None, depth=False):
    from gym_miniworld_amd.batch import BatchedMiniWorld
    b = BatchedMiniWorld(env_id, num_envs=2, seed=1, obs_width=W, obs_height=H, want_depth=depth)
    b.reset()
    x, z, d = fx["fit_pose"]
    bx, bz, bd = box if box is not None else RS.hidden_box_pose(fx)
    b.set_state(0, agent_pos=[[x, 0.0, z]] * 2, agent_dir=[d] * 2, boxes_pos=[[[bx, 0.0, bz]]] * 2, boxes_dir=[[bd]] * 2)
    st = b.get_state()
    assert np.array_equal(st["agent_pos"][1], [x, 0.0, z]) and st["agent_dir"][0] == d and st["box_pos"][0, 0] == bx
    img = b.render().cpu().numpy()
    dep = b.depth.cpu().numpy()[..., 0] if depth else None
    b.close()
    assert np.array_equal(img[0], img[1])
    return img[0], (dep[0] if depth else None)


@pytest.mark.parametrize("name", RS.CASES)
def test_hip_render_matches_the_reference_screenshot(name):
    fx = RS.load(name)
    assert RS.pose_inside_hud_interval(fx)
    for key, mkey, W, H, block, tol in RS.views(fx):
        img, _ = _render(fx, RS.ENV_IDS[name], W, H)
        RS.check(RS.stats(img, fx[key], fx[mkey], block), tol, "%s/%s" % (name, key))


@pytest.mark.parametrize("name", RS.ENT_CASES)
def test_hip_render_matches_the_reference_screenshots_of_the_entity_tasks(name):
    """the entity tasks' scenes: rooms, the building, the cones, the ImageFrame - through mwb_set_state + mwb_render"""
    from gym_miniworld_amd.batch import BatchedMiniWorld
    fx = RS.load(name)
    env_id, _, _, keep, _ = RS.ENT_CASES[name]
    x, z, d = fx["fit_pose"]
    bx, bz = RS.far_behind(fx)
    for key, mkey, W, H, block, tol in RS.ent_views(name):
        b = BatchedMiniWorld(env_id, num_envs=2, seed=1, obs_width=W, obs_height=H)
        b.reset()
        st = b.get_state()
        pos, dirs = st["boxes_pos"].copy(), st["boxes_dir"].copy()
        for k in range(b.n_boxes):
            if RS.KIND_NAMES.get(int(st["ent_kind"][0, k]), "frame") not in keep:
                pos[0, k] = [bx, 0.0, bz]
                dirs[0, k] = 0.0
        b.set_state(0, agent_pos=[[x, 0.0, z]] * 2, agent_dir=[d] * 2, boxes_pos=pos, boxes_dir=dirs)
        img = b.render().cpu().numpy()[0]
        b.close()
        RS.check(RS.stats(img, fx[key], fx[mkey], block), tol, "%s/%s" % (name, key))


def test_hip_box_face_colours_match_the_screenshots():
    fx = RS.load("hallway")
    x, z, d = fx["fit_pose"]
    img, _ = _render(fx, RS.ENV_IDS["hallway"], 160, 120, box=(x + 2.5 * math.cos(d), z - 2.5 * math.sin(d), 0.0))
    px = img.reshape(-1, 3).astype(np.float64)
    red = px[(px[:, 0] > px[:, 1] + 80) & (px[:, 0] > px[:, 2] + 80)]
    assert len(red) > 100
    thr = 0.5 * (np.percentile(red[:, 0], 5) + np.percentile(red[:, 0], 95))
    top, side = np.median(red[red[:, 0] > thr], axis=0), np.median(red[red[:, 0] <= thr], axis=0)
    for ref in (fx["box_faces"], RS.load("oneroom")["box_faces"]):
        assert abs(top[0] - ref[0, 0]) <= 12 and top[1] <= 6 and top[2] <= 6, (top, ref[0])
        assert abs(side[0] - ref[1, 0]) <= 6 and side[1] <= 6 and side[2] <= 6, (side, ref[1])


def test_hip_depth_is_planar_linear_like_the_depth_map_screenshot():
    assert RS.depth_is_planar_and_linear(np.load(os.path.join(RS.GOLDEN, "refimg_depth_map.npz"))["lum160"]) > 0
    fx = {"fit_pose": np.array([0.0, 0.0, 0.0]), "hud_pos": np.array([0.0, 0.0]), "hud_angle": 0}
    _, dep = _render(fx, RS.ENV_IDS["hallway"], 160, 120, depth=True)
    rows = np.arange(100, 120)
    band = dep[rows][:, 50:115].astype(np.float64)
    z = 1.5 / (((rows + 0.5 + 3.0 / 16.0) - 60.0) / 60.0 * math.tan(math.radians(30.0)))
    assert np.abs(band - z[:, None]).max() < 0.008
