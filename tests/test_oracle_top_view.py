"""render_top_view (miniworld.py:1087-1158), CPU legs.

(1) The oracle's top view against a brute-force float64 rendition (tests/soup_renderer.py, orthographic mode) of the GL
    call stream the UNMODIFIED reference issues for render_top_view() - room polygons of the display list, glOrtho, the
    fixed modelview, the boxes and the agent's triangle with whatever normal was current (tests/golden/gltop_*.json,
    generator: gen_fixtures.capture_gl_top): YMaze (polygon rooms, up-facing ceilings of the reversed connector
    slivers), FourRooms with domain randomisation, PutNext (six boxes); agent moved off its spawn pose in two of them.
(2) The same view against a frame the reference itself holds: the 800 x 600 main area (kept box-filtered to 400 x 300) of /root/reference/images/
    ymaze_0.jpg is a top view (manual_control.py --top_view); pose from the HUD text as fitted for the observation inset
    (tests/golden/refimg_ymaze.npz), box unknown and masked.  A coarse pin (JPEG, 16 samples there, 8 here): floorplan
    overlap, sky / floor colours, checkerboard phase, the agent's triangle where and as dark as the reference drew it."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
STREAMS = [("YMaze", 0, "YMaze", [0, 0, 0, 0]), ("FourRooms", 1, "FourRooms", None), ("PutNext", 1, "PutNext", None)]


def posed_env(O, g, task, args, dr):
    env = O.OracleEnv(task, seed=g["seed"], domain_rand=dr, task_args=args)
    env.reset(render=False)
    env.set_agent(g["agent_pos"][0], g["agent_pos"][2], g["agent_dir"])
    return env


def load(name, dr):
    with open(os.path.join(GOLD, "gltop_%s_dr%d.json" % (name, dr))) as fh:
        return json.load(fh)


@pytest.mark.parametrize("name,dr,task,args", STREAMS)
def test_top_view_inputs_equal_the_reference_stream(oracle_mod, name, dr, task, args):
    """extents, glOrtho frame, entity poses and the agent's triangle of the reference's own frame"""
    g = load(name, dr)
    env = posed_env(oracle_mod, g, task, args, dr)
    geo, s = env.geometry(), env.state()
    o = geo["outline"]
    ext = [np.nanmin(o[:, :, 0]), np.nanmax(o[:, :, 0]), np.nanmin(o[:, :, 1]), np.nanmax(o[:, :, 1])]
    assert ext == g["extents"]
    l, r, b, t, n, f = g["misc"]["glOrtho"]
    w, h = ext[1] - ext[0] + 2, ext[3] - ext[2] + 2
    assert abs((r - l) / (t - b) - 80 / 60) < 1e-12 and (abs((r - l) - w) < 1e-12 or abs((t - b) - h) < 1e-12)
    assert abs((l + r) / 2 - (ext[0] + ext[1]) / 2) < 1e-12 and abs(-(b + t) / 2 - (ext[2] + ext[3]) / 2) < 1e-12
    tri = g["polys"][-1]
    assert tri["mode"] == "GL_TRIANGLES" and tri["color"] == [1.0, 0.0, 0.0] and tri["norms"][0] == [0.0, -1.0, 0.0]
    ax, az, ad = s.agent_pos[0], s.agent_pos[2], s.agent_dir
    rad = s.agent_radius
    dv, rv = np.array([np.cos(ad), 0, -np.sin(ad)]) * rad, np.array([np.sin(ad), 0, np.cos(ad)]) * rad
    p = np.array([ax, 1.6, az])
    want = [p + dv, p + 0.75 * (-rv - dv), p + 0.75 * (rv - dv)]   # glVertex order p0, p2, p1 (entity.py:511-514)
    assert np.abs(np.array(tri["verts"]) - np.array(want)).max() < 1e-6   # glVertex3f arguments pass through float32
    assert np.array_equal(np.array(g["boxes_pos"])[:, [0, 2]], np.array(s.boxes_pos)[:s.n_boxes][:, [0, 2]])


@pytest.mark.parametrize("name,dr,task,args", STREAMS)
@pytest.mark.parametrize("size", [(80, 60), (200, 150)])
def test_top_view_equals_bruteforce_rendition_of_the_reference_stream(oracle_mod, name, dr, task, args, size):
    import soup_renderer as SR
    O = oracle_mod
    g = load(name, dr)
    tex = O.load_textures()
    env = posed_env(O, g, task, args, dr)
    W, H = size
    a = env.render_top(W, H)
    with np.errstate(all="ignore"):
        b = SR.render_stream(g, {O.TEX_FILES[i]: tex[i][2] for i in tex}, W, H, ortho=True)
    d = np.abs(a.astype(int) - b.astype(int))
    assert (d.max(axis=2) > 1).mean() <= 1e-3 and d.mean() < 0.02, (int(d.max()), int((d.max(axis=2) > 1).sum()))
    assert (a.reshape(-1, 3) == a[0, 0]).all(axis=1).mean() < 0.9   # not an empty frame


def top_view_pin_stats(img, ref):
    """img, ref: 400 x 300 uint8 RGB.  Returns the statistics both legs assert on."""
    from scipy.ndimage import binary_erosion, gaussian_filter, label, center_of_mass
    f, r = ref.astype(np.float64), img.astype(np.float64)
    sky_ref = (f[..., 2] > 200) & (f[..., 0] < 120)
    sky_img = (r[..., 2] > 200) & (r[..., 0] < 120)
    red_ref = (f[..., 0] > 120) & (f[..., 1] < 70) & (f[..., 2] < 70)
    red_img = (r[..., 0] > 120) & (r[..., 1] < 70) & (r[..., 2] < 70)
    out = {"floorplan_iou": ((~sky_ref) & (~sky_img)).sum() / ((~sky_ref) | (~sky_img)).sum()}
    out["sky_diff"] = np.abs(f[sky_ref & sky_img].mean(axis=0) - r[sky_ref & sky_img].mean(axis=0)).max()
    floor = binary_erosion(~sky_ref & ~sky_img & ~red_ref & ~red_img, iterations=3)
    out["floor_mean_diff"] = np.abs(f[floor].mean(axis=0) - r[floor].mean(axis=0)).max()
    lf, lr = gaussian_filter(f.mean(axis=2), 0.7), gaussian_filter(r.mean(axis=2), 0.7)
    out["checker_corr"] = np.corrcoef(lf[floor], lr[floor])[0, 1]
    # the agent's triangle: the dark red blob (the box's top is bright red)
    def agent_blob(m, lum):
        lab, n = label(m)
        best = None
        for i in range(1, n + 1):
            sel = lab == i
            if sel.sum() < 10:
                continue
            if best is None or lum[sel].mean() < best[0]:
                best = (lum[sel].mean(), center_of_mass(sel), int(sel.sum()))
        return best
    out["agent_ref"], out["agent_img"] = agent_blob(red_ref, f[..., 0]), agent_blob(red_img, r[..., 0])
    return out


def check_top_view_pin(st):
    assert st["floorplan_iou"] > 0.985, st
    assert st["sky_diff"] <= 3.0 and st["floor_mean_diff"] <= 4.0, st
    assert st["checker_corr"] > 0.9, st
    (lum_r, com_r, area_r), (lum_i, com_i, area_i) = st["agent_ref"], st["agent_img"]
    assert abs(com_r[0] - com_i[0]) <= 2.0 and abs(com_r[1] - com_i[1]) <= 2.0, st      # pixels of 400 x 300 (the HUD rounds the pose to 1 cm ~ 0.15 px)
    assert abs(lum_r - lum_i) <= 10.0 and 0.8 <= area_i / area_r <= 1.25, st             # ambient-only red: lit with the stale (0, -1, 0) normal


def test_top_view_matches_the_reference_screenshot(oracle_mod):
    fx = np.load(os.path.join(GOLD, "refimg_ymaze.npz"))
    x, z, d = fx["fit_pose"]
    env = oracle_mod.OracleEnv("YMaze", seed=1, task_args=[0, 0, 0, 0])
    env.reset(render=False)
    env.set_agent(x, z, d)
    env.set_box(0, -8.0, 0.0, 0.0)   # the screenshot's box pose is unknown: ours goes under the agent's end of the corridor, both masked as red
    img = env.render_top(400, 300)
    st = top_view_pin_stats(img, fx["top400"])
    check_top_view_pin(st)


def test_maze_top_view_frame_matches_the_reference_screenshot(oracle_mod):
    """images/maze_top_view.jpg: the layout is random, the frame is not - glOrtho extents, cell pitch, gap width, the agent's
    triangle at the HUD position (tests/refimg_stats.maze_top_stats); two layouts of ours against the screenshot's"""
    import math
    import refimg_stats as RS
    fx = np.load(os.path.join(GOLD, "refimg_maze_top.npz"))
    ref = RS.maze_top_stats(fx["sky_mask"], fx["red_mask"], fx["hud_pos"])
    for seed in (1, 2):
        env = oracle_mod.OracleEnv("Maze", seed=seed)
        env.reset(render=False)
        env.set_agent(float(fx["hud_pos"][0]), float(fx["hud_pos"][1]), math.radians(float(fx["hud_angle"]) + 0.5))
        t = env.render_top(800, 600).astype(np.float64)
        sky = np.abs(t - t[5, 5]).max(axis=2) < 50
        red = (t[..., 0] > t[..., 1] + 60) & (t[..., 0] > t[..., 2] + 60)
        RS.check_maze_top(RS.maze_top_stats(sky, red, fx["hud_pos"]), ref)
    # teeth: the same statistics with another cell count do not describe the screenshot
    assert RS.maze_top_stats(fx["sky_mask"], fx["red_mask"], fx["hud_pos"], rows=7, cols=7, room=3.5, gap=0.25)["in_band"] < 0.6


# ------------------------------------------------------------------------------------------- get_visible_ents
VIS_STREAMS = [("PutNext", 0, "PutNext", None), ("PutNext", 1, "PutNext", None), ("TMazeTwoBoxFeatures", 1, "TMazeTwoBox", [1, 0, 0, 100000]),
               ("FourRooms", 0, "FourRooms", None), ("YMaze", 1, "YMaze", [0, 0, 0, 0]), ("Hallway", 1, "Hallway", None)]


@pytest.mark.parametrize("name,dr,task,args", VIS_STREAMS)
def test_visible_ents_equals_occlusion_queries_on_the_reference_polygons(oracle_mod, name, dr, task, args):
    """get_visible_ents (miniworld.py:1222-1315; the reference never calls it and holds no output of it: parity unpinned):
    the oracle's restatement (portal traversal + cube slabs) against the same queries evaluated on the reference's own room
    polygons (glstream_*.json) by the brute-force soup - from the stream's camera and from a ring of turned poses."""
    import math
    import test_oracle_render as TR
    import soup_renderer as SR
    O = oracle_mod
    g = TR.load_stream(name, dr)
    env = TR.posed_env(O, g, task, args, dr)
    s = env.state()
    boxes = [list(s.boxes_pos[b]) for b in range(s.n_boxes)]
    seen_any = 0
    for turn in range(8):
        ang = g["agent_dir"] + turn * math.pi / 4
        env.set_agent(g["agent_pos"][0], g["agent_pos"][2], ang)
        st = env.state()
        g2 = dict(g)
        g2["misc"] = dict(g["misc"])
        cp, cdir = np.array(st.cam_pos), np.array(st.cam_dir)
        g2["misc"]["gluLookAt"] = list(cp) + list(cp + cdir) + [0.0, 1.0, 0.0]
        want = SR.visible_cubes(g2, boxes)
        mask = env.visible_ents()
        got = {b for b in range(s.n_boxes) if (mask >> b) & 1}
        assert got == want, (name, dr, turn, got, want)
        seen_any += len(got)
    if name == "PutNext":
        assert seen_any > 0


@pytest.mark.parametrize("name,dr", [("PickupObjs", 0), ("PickupObjs", 1), ("RoomObjs", 1), ("ThreeRooms", 0), ("Sidewalk", 1), ("CollectHealth", 0)])
def test_visible_ents_of_an_entity_list_equals_the_queries_on_the_reference_polygons(oracle_mod, name, dr):
    """the tasks with a general entity list: one cube per entry of self.entities (whatever the entity is: mesh, frame, box), in list
    order, none for entities that left the list - against the queries evaluated on the streams' room polygons"""
    import math
    import test_oracle_ents_render as ER
    import soup_renderer as SR
    g = ER.load_stream(name, dr)
    env = ER.replay(oracle_mod, g, name, dr)
    s = env.state()
    slots = [int(q) for q in list(s.order)[:s.n_order] if 0 <= int(q) < s.n_boxes]
    positions = [list(s.boxes_pos[b]) for b in slots]
    seen_any = 0
    for turn in range(8):
        ang = s.agent_dir + turn * math.pi / 4
        env.set_agent(s.agent_pos[0], s.agent_pos[2], ang)
        st = env.state()
        g2 = dict(g)
        g2["misc"] = dict(g["misc"])
        cp, cdir = np.array(st.cam_pos), np.array(st.cam_dir)
        g2["misc"]["gluLookAt"] = list(cp) + list(cp + cdir) + [0.0, 1.0, 0.0]
        g2["polys"] = g["room_polys"]
        want = {slots[i] for i in SR.visible_cubes(g2, positions)}
        mask = env.visible_ents()
        got = {b for b in range(s.n_boxes) if (mask >> b) & 1}
        assert got == want, (name, dr, turn, got, want)
        seen_any += len(got)
    assert seen_any > 0
