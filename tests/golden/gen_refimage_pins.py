#!/usr/bin/env python3
"""Coarse PIXEL pin against frames the reference itself holds: /root/reference/images/<task>_0.jpg.

Each of those screenshots of manual_control.py shows, in one window (miniworld.py:1317-1425):
  * the 800x600 human view  = render_obs(self.vis_fb)   (miniworld.py:1330-1331), window cols 0..799
  * the agent's 80x60 observation = render_obs(), blitted at 256x192 in the top right corner (1389-1404)
  * the HUD text "pos: (%.2f, %.2f, %.2f) / angle: %d / steps" (1406-1412): the pose, rounded.
Both views are the path this repo re-implements, so they pin lighting, texture orientation / phase,
wall / floor / ceiling geometry and the camera - coarsely: JPEG, a pose known only to the HUD's
rounding (fitted here INSIDE that interval), and screenshots that predate the mip-mapped textures
(far-field aliasing differs).  +-1 LSB parity stays unverifiable (no GL in this pipeline).

This script crops the views, box-filters them to 160x120 and 80x60, fits the pose within the HUD
interval with the CPU oracle, and writes tests/golden/refimg_<task>.npz (data only).  Not used:
fourrooms_0.jpg (an older FourRooms: two boxes, narrower doors, HUD without % 360), maze_0.jpg (the
maze layout is random and the seed unknown), the tasks this repo does not cover.
depth_map.jpg (unknown task) contributes one property: the displayed depth is planar eye-space z.

Run (only where /root/reference exists):  python tests/golden/gen_refimage_pins.py
"""
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
IMAGES = "/root/reference/images"

# task of the oracle, HUD position (x, z), HUD angle, file
CASES = {
    "hallway": ("Hallway", (-0.02, -0.03), 1, "hallway_0.jpg"),
    "oneroom": ("OneRoom", (0.63, 8.42), 30, "oneroom_0.jpg"),
    "tmaze": ("TMaze", (3.41, -0.31), 359, "tmaze_0.jpg"),
    # ymaze_0.jpg shows the TOP view in the main area; only its 80x60 observation inset is the path pinned here
    "ymaze": ("YMaze", (-5.64, 0.48), 5, "ymaze_0.jpg"),
}
INSET_ONLY = {"ymaze"}
# Scenes with mesh entities / frames (round 3).  The world of these tasks is fixed apart from the entities the task places at random,
# which are unknown (unknown seed): those are moved out of view in the rendition and masked by their colour in the screenshot.  What is
# NOT random stays in view and is pinned with the rooms: Sidewalk's building and cones (sidewalk.py:43-60: fixed positions), WallGap's
# building (wallgap.py:44-52), ThreeRooms' ImageFrame (threerooms.py:47-52; objects.jpg is a ThreeRooms window).
#   name: (oracle task, task_args, HUD (x, z), HUD angle, file, colour classes masked, kinds of entity kept in view)
# Not used: sign.jpg (no HUD: pose unknown), textframe.jpg (a scene of no registered env).
ENT_CASES = {
    "pickupobjs": ("PickupObjs", [12, 5, 0, 0], (3.03, 4.12), 289, "pickupobjs_0.jpg", ("red", "green", "purple"), ()),
    "collecthealth": ("CollectHealth", [16, 0, 0, 0], (15.55, 12.52), 163, "collecthealth_0.jpg", ("red",), ()),
    "objects": ("ThreeRooms", None, (-1.17, 3.04), 291, "objects.jpg", ("red", "green", "blue"), ("frame",)),
    "sidewalk": ("Sidewalk", None, (-1.89, 0.41), 298, "sidewalk_0.jpg", ("pure_red",), ("mesh",)),
    "wallgap": ("WallGap", None, (-2.52, -5.01), 294, "wallgap_0.jpg", ("pure_red", "sky"), ("mesh",)),   # its sky is (110, 207, 255): a revision before today's sky_color
}
# window = 24-pixel title bar + 1-pixel frame left / right / below around the 1056x600 client area (the crop that
# minimises the fit residual of all three screenshots; one pixel off in either direction doubles it)
MAIN = (slice(24, 624), slice(1, 801))       # the 800x600 human view inside the 1058x625 screenshot
INSET = (slice(24, 216), slice(801, 1057))   # the 256x192 blit of the 80x60 observation


def box_down(a, f):
    h, w, c = a.shape
    return a.reshape(h // f, f, w // f, f, c).mean(axis=(1, 3))


def hud_angle_interval(a):
    """int(dir * 180 / pi) % 360 (miniworld.py:1409) truncates towards zero: the displayed integer a
    means dir in [a, a + 1) degrees, or, for a negative heading shown modulo 360, (a - 361, a - 360]."""
    return [(a, a + 1.0)] + ([(a - 361.0, a - 360.0)] if a > 180 else [])


def non_box_mask(img, grow):
    """False on and around saturated pixels: the boxes are the only coloured things in these scenes
    (their size / height differs between the screenshots' revision and today's Box(0.8))."""
    from scipy.ndimage import binary_dilation
    colored = (img.max(axis=2) - img.min(axis=2)) > 40
    return ~binary_dilation(colored, iterations=grow)


def colour_mask(img, classes, grow):
    """False on and around the pixels of the entities a task places at random, told by their colour class"""
    from scipy.ndimage import binary_dilation
    r, g, b = img[..., 0], img[..., 1], img[..., 2]
    rules = {"red": (r > g + 40) & (r > b + 40), "pure_red": (r > 90) & (g < 60) & (b < 60) & (r > g + 60),
             "green": (g > r + 60) & (g > b + 60), "blue": (b > r + 60) & (b > g + 60), "purple": (b > g + 40) & (r > g + 20),
             "sky": (b > 200) & (g > 170) & (r < 160)}
    m = np.zeros(r.shape, bool)
    for c in classes:
        m |= rules[c]
    return ~binary_dilation(m, iterations=grow)


def ent_scene(O, case, W, H):
    """the oracle's env of an ENT_CASES entry with every randomly placed entity moved far behind the camera"""
    task, args, (hx, hz), hang, _, _, keep = case
    e = O.OracleEnv(task, seed=1, obs_width=W, obs_height=H, task_args=args)
    e.reset(render=False)
    s = e.state()
    a0 = math.radians(hang + 0.5)
    for k in range(s.n_boxes):
        kind = {0: "box", 1: "mesh"}.get(int(s.ents_kind[k]), "frame")
        if kind not in keep:
            e.set_box(k, hx - 40.0 * math.cos(a0), hz + 40.0 * math.sin(a0), 0.0)
    return e


def fit_pose(task, hx, hz, hang, ref, mask, env=None):
    from scipy.optimize import minimize
    from oracle import oracle as O
    H, W, _ = ref.shape
    e = env
    if e is None:
        e = O.OracleEnv(task, seed=1, obs_width=W, obs_height=H, task_args=[0, 0, 0, 0] if task == "YMaze" else None)
        e.reset(render=False)
        a0 = math.radians(hang + 0.5)
        e.set_box(0, hx - 0.6 * math.cos(a0), hz + 0.6 * math.sin(a0), 0.0)   # behind the camera: not in view
    best = None
    for lo, hi in hud_angle_interval(hang):
        def cost(p, lo=lo, hi=hi):
            x = min(max(p[0], hx - 0.005), hx + 0.005)
            z = min(max(p[1], hz - 0.005), hz + 0.005)
            d = min(max(p[2], math.radians(lo)), math.radians(hi))
            pen = abs(x - p[0]) + abs(z - p[1]) + abs(d - p[2])
            e.set_agent(x, z, d)
            return np.abs(e.render_obs().astype(np.float64) - ref)[mask].mean() + 1e3 * pen
        c0 = min(((cost([hx, hz, math.radians(a)]), a) for a in np.linspace(lo + 0.02, hi - 0.02, 25)))
        p0 = [hx, hz, math.radians(c0[1])]
        simplex = [p0, [p0[0] + 0.003, p0[1], p0[2]], [p0[0], p0[1] + 0.003, p0[2]], [p0[0], p0[1], p0[2] + 0.002]]
        res = minimize(cost, p0, method="Nelder-Mead", options={"xatol": 1e-5, "fatol": 1e-4, "initial_simplex": simplex})
        if best is None or res.fun < best[0]:
            best = (res.fun, [min(max(res.x[0], hx - 0.005), hx + 0.005), min(max(res.x[1], hz - 0.005), hz + 0.005),
                              min(max(res.x[2], math.radians(lo)), math.radians(hi))])
    return best


def main():
    from PIL import Image
    only = set(sys.argv[1:])   # optional: the cases to regenerate
    for name, (task, (hx, hz), hang, fn) in CASES.items():
        if only and name not in only:
            continue
        im = np.asarray(Image.open(os.path.join(IMAGES, fn)).convert("RGB")).astype(np.float64)
        main_view = im[MAIN]
        inset_f = np.asarray(Image.fromarray(im[INSET].astype(np.uint8)).resize((80, 60), Image.BOX)).astype(np.float64)
        if name in INSET_ONLY:
            cost, pose = fit_pose(task, hx, hz, hang, inset_f, non_box_mask(inset_f, 2))
        else:
            ref400 = box_down(main_view, 2)
            cost, pose = fit_pose(task, hx, hz, hang, ref400, non_box_mask(ref400, 5))
        main160 = box_down(main_view, 5)
        main80 = box_down(main_view, 10)
        inset = np.asarray(Image.fromarray(im[INSET].astype(np.uint8)).resize((80, 60), Image.BOX)).astype(np.float64)
        # the lit faces of the reference's box: bright cluster = top face, dark cluster = the side(s) in view
        px = main_view.reshape(-1, 3)
        red = px[(px[:, 0] > px[:, 1] + 80) & (px[:, 0] > px[:, 2] + 80)]
        faces = np.zeros((2, 3))
        if len(red) > 200:
            thr = 0.5 * (np.percentile(red[:, 0], 5) + np.percentile(red[:, 0], 95))
            faces[0] = np.median(red[red[:, 0] > thr], axis=0)
            faces[1] = np.median(red[red[:, 0] <= thr], axis=0)
        out = os.path.join(HERE, "refimg_%s.npz" % name)
        np.savez_compressed(
            out, task=task, hud_pos=np.array([hx, hz]), hud_angle=hang, fit_pose=np.array(pose), fit_cost=cost,
            main160=np.rint(main160).astype(np.uint8), main80=np.rint(main80).astype(np.uint8), inset_only=name in INSET_ONLY,
            inset80=np.rint(inset).astype(np.uint8), mask160=non_box_mask(main160, 3), mask80=non_box_mask(main80, 2),
            mask_inset=non_box_mask(inset, 2), box_faces=faces,
            # ymaze_0.jpg's main area is a TOP VIEW (manual_control.py --top_view): kept box-filtered to 400 x 300
            **({"top400": np.rint(box_down(main_view, 2)).astype(np.uint8)} if name in INSET_ONLY else {}))
        print("%-8s pose %.4f %.4f %.3f deg  cost %.2f  box faces %s -> %s" %
              (name, pose[0], pose[1], math.degrees(pose[2]), cost, np.round(faces).tolist(), os.path.basename(out)))
    from oracle import oracle as O
    for name, case in ENT_CASES.items():
        if only and name not in only:
            continue
        task, args, (hx, hz), hang, fn, classes, keep = case
        im = np.asarray(Image.open(os.path.join(IMAGES, fn)).convert("RGB")).astype(np.float64)
        main_view = im[MAIN]
        ref200 = box_down(main_view, 4)   # 200 x 150: the building's hierarchy makes a 400 x 300 fit an hour's work on the CPU
        cost, pose = fit_pose(task, hx, hz, hang, ref200, colour_mask(ref200, classes, 4), env=ent_scene(O, case, 200, 150))
        main160, main80 = box_down(main_view, 5), box_down(main_view, 10)
        inset = np.asarray(Image.fromarray(im[INSET].astype(np.uint8)).resize((80, 60), Image.BOX)).astype(np.float64)
        out = os.path.join(HERE, "refimg_%s.npz" % name)
        np.savez_compressed(
            out, task=task, hud_pos=np.array([hx, hz]), hud_angle=hang, fit_pose=np.array(pose), fit_cost=cost,
            main160=np.rint(main160).astype(np.uint8), main80=np.rint(main80).astype(np.uint8), inset_only=False,
            inset80=np.rint(inset).astype(np.uint8), mask160=colour_mask(main160, classes, 3), mask80=colour_mask(main80, classes, 2),
            mask_inset=colour_mask(inset, classes, 2))
        print("%-13s pose %.4f %.4f %.3f deg  cost %.2f -> %s" % (name, pose[0], pose[1], math.degrees(pose[2]), cost, os.path.basename(out)))
    if not only or "maze_top" in only:
        # maze_top_view.jpg (manual_control.py --top_view on Maze): the layout is random, the FRAME is not - the glOrtho extents of
        # render_top_view (miniworld.py:1087-1158), the cell pitch 3.25 m and the 0.25 m gaps between rooms through which the sky shows
        # (maze.py:33-52), the agent's triangle at the HUD position.  Kept: where the sky colour shows, where red shows.  (The floor's
        # brightness falls off away from the corner at the origin: the positional light of that revision - colours are not compared.)
        im = np.asarray(Image.open(os.path.join(IMAGES, "maze_top_view.jpg")).convert("RGB")).astype(np.float64)
        mv = im[MAIN]
        sky = np.abs(mv - mv[5, 5]).max(axis=2) < 50
        red = (mv[..., 0] > mv[..., 1] + 60) & (mv[..., 0] > mv[..., 2] + 60)
        np.savez_compressed(os.path.join(HERE, "refimg_maze_top.npz"), sky_mask=sky, red_mask=red, hud_pos=np.array([17.10, 18.52]), hud_angle=168)
        print("maze_top_view -> refimg_maze_top.npz")
    if only and "depth_map" not in only:
        return
    # depth_map.jpg: grey level of the human view = displayed depth.  Kept: the 160x120 box-filtered luminance.
    im = np.asarray(Image.open(os.path.join(IMAGES, "depth_map.jpg")).convert("L")).astype(np.float64)
    d160 = box_down(im[MAIN][..., None], 5)[..., 0]
    np.savez_compressed(os.path.join(HERE, "refimg_depth_map.npz"), lum160=np.rint(d160).astype(np.uint8))
    print("depth_map -> refimg_depth_map.npz")


if __name__ == "__main__":
    main()
