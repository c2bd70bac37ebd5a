#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the UNMODIFIED reference
(/root/reference, read-only) in the build container.  `gym` and `pyglet` are absent here, so
they are replaced by the inert stand-ins of _inert_deps.py: every OpenGL entry point records
its arguments and computes nothing.  Consequently these fixtures pin the reference's *state*
arithmetic (geometry, RNG draw order, placement, trajectories, rewards, dones, camera vectors)
and the exact *inputs* it hands to OpenGL (vertex / texcoord / normal / colour / light / camera
streams).  They do NOT contain reference pixels - none can be produced here.

Run:  python tests/golden/gen_fixtures.py          (only in the container that has /root/reference)
Outputs: state_<task>.npz, glstream_<task>.json, gltop_<task>.json (render_top_view streams), math_kat.npz  (data only, no reference text).
"""
import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _inert_deps as deps  # noqa: E402

deps.install("/root/reference")

import gym_miniworld  # noqa: E402,F401
from gym_miniworld.envs import FourRooms, Hallway, Maze, OneRoom  # noqa: E402
from gym_miniworld.envs import (TMaze, TMazeDynamic, TMazeLeft, TMazeRight, TMazeTwoBoxDynamic,  # noqa: E402
                                TMazeTwoBoxDynamicFeatures100K, TMazeTwoBoxDynamicFeaturesDebug)
from gym_miniworld.envs import SimToRealGoTo, SimToRealPush  # noqa: E402
from gym_miniworld.envs import PutNext  # noqa: E402
from gym_miniworld.envs import YMaze, YMazeLeft, YMazeRight  # noqa: E402
from gym_miniworld.opengl import Texture  # noqa: E402
from gym_miniworld import math as ref_math  # noqa: E402

# name -> (reference class, constructor kwargs).  MazeS3 / OneRoomS6 / Hallway6 exercise the
# constructor parameters the reference exposes (maze.py:14-21, oneroom.py:14, hallway.py:13).
TASKS = {
    "Hallway": (Hallway, {}), "OneRoom": (OneRoom, {}), "FourRooms": (FourRooms, {}), "Maze": (Maze, {}),
    "MazeS3": (Maze, {"num_rows": 3, "num_cols": 3}),
    "MazeR2C4": (Maze, {"num_rows": 2, "num_cols": 4, "room_size": 2.5, "max_episode_steps": 200}),
    "OneRoomS6": (OneRoom, {"size": 6, "max_episode_steps": 100}),
    "Hallway6": (Hallway, {"length": 6}),
    # the T-maze family (envs/tmaze.py); small sub_task_length so that the goal alternation
    # rules fire within the recorded trajectories
    "TMaze": (TMaze, {}), "TMazeLeft": (TMazeLeft, {}), "TMazeRight": (TMazeRight, {}),
    "TMazeDynamic3": (TMazeDynamic, {"sub_task_length": 3}),
    "TMazeTwoBoxDynamic2": (TMazeTwoBoxDynamic, {"sub_task_length": 2}),
    "TMazeTwoBoxFeatures": (TMazeTwoBoxDynamicFeatures100K, {"sub_task_length": 150}),
    "TMazeTwoBoxFeaturesDebug": (TMazeTwoBoxDynamicFeaturesDebug, {}),
}
TASKS.update({"SimToRealGoTo": (SimToRealGoTo, {}), "SimToRealPush": (SimToRealPush, {})})   # no-ceiling rinks, own params
# non-rectangular rooms: a triangular hub, two arms rotated by 120 degrees, two sliver-shaped connectors
TASKS.update({"YMaze": (YMaze, {}), "YMazeLeft": (YMazeLeft, {}), "YMazeRight": (YMazeRight, {})})
NO_KWARGS = {"TMazeLeft", "TMazeRight", "TMazeDynamic3", "YMazeLeft", "YMazeRight"}   # constructors without **kwargs: no domain_rand

# (seed, domain_rand, policy, n_steps)
PLAN = {
    "Hallway": [(0, 0, "random", 600), (1, 0, "greedy", 200), (2, 1, "random", 600), (3, 1, "greedy", 200),
                (7, 0, "forward", 300)],
    "OneRoom": [(0, 0, "random", 500), (1, 0, "greedy", 300), (2, 1, "random", 500), (3, 1, "greedy", 300),
                (7, 0, "forward", 400), (4097, 1, "random", 200)],
    "FourRooms": [(0, 0, "random", 600), (1, 0, "greedy", 400), (2, 1, "random", 600), (3, 1, "greedy", 400),
                  (7, 1, "forward", 300)],
    "Maze": [(0, 0, "random", 1700), (1, 0, "greedy", 300), (2, 1, "random", 400), (3, 1, "forward", 300),
             (65535, 0, "random", 100)],
    "MazeS3": [(0, 0, "greedy", 500), (1, 1, "greedy", 500), (2, 0, "forward", 500), (3, 1, "random", 300)],
    "MazeR2C4": [(0, 0, "greedy", 450), (1, 1, "forward", 450)],
    "OneRoomS6": [(0, 0, "random", 250), (1, 1, "greedy", 150)],
    "Hallway6": [(0, 0, "random", 300), (1, 1, "greedy", 100)],
    "TMaze": [(0, 0, "random", 700), (1, 0, "greedy", 400), (2, 1, "greedy", 400), (3, 1, "forward", 400)],
    "TMazeLeft": [(0, 0, "greedy", 300), (1, 0, "random", 600)],
    "TMazeRight": [(0, 0, "greedy", 300), (1, 0, "forward", 400)],
    "TMazeDynamic3": [(0, 0, "greedy", 1200), (1, 0, "forward", 900)],
    "TMazeTwoBoxDynamic2": [(0, 0, "greedy", 700), (1, 1, "greedy_red", 700), (2, 0, "random", 600)],
    "TMazeTwoBoxFeatures": [(0, 0, "greedy", 600), (1, 1, "greedy_blue", 600), (2, 1, "forward", 500)],
    "TMazeTwoBoxFeaturesDebug": [(0, 0, "greedy_blue", 300), (1, 1, "greedy", 300)],
    # these two classes force domain_rand=True themselves (simtorealgoto.py:31, simtorealpush.py:30)
    "SimToRealGoTo": [(0, 1, "random", 400), (1, 1, "greedy", 400), (2, 1, "forward", 300)],
    "SimToRealPush": [(0, 1, "random", 500), (1, 1, "push", 900), (2, 1, "greedy", 600), (3, 1, "push", 900)],
    "YMaze": [(0, 0, "random", 600), (1, 0, "greedy", 500), (2, 1, "greedy", 500), (3, 1, "random", 600), (7, 0, "forward", 400)],
    "YMazeLeft": [(0, 0, "greedy", 400), (1, 0, "random", 500)],
    "YMazeRight": [(0, 0, "greedy", 400), (1, 0, "forward", 400)],
}


def tex_basename(tex):
    for path, t in Texture.tex_cache.items():
        if t is tex:
            return os.path.splitext(os.path.basename(path))[0]
    raise KeyError


def rng_fingerprint(env):
    st = env.rand.np_random.get_state()
    key = st[1]
    return np.array([st[2], int(key[0]), int(key[1]), int(key[623]), int(key.astype(np.uint64).sum() & 0xFFFFFFFF)],
                    dtype=np.int64)


def snapshot_world(env):
    R = len(env.rooms)
    def pad4(a):   # rooms are quadrilaterals except YMaze's triangular hub: its missing fourth row is NaN
        a = np.asarray(a)
        return a if a.shape[0] == 4 else np.concatenate([a.astype(float), np.full((4 - a.shape[0],) + a.shape[1:], np.nan)])
    outline = np.stack([pad4(np.stack([r.outline[:, 0], r.outline[:, 2]], axis=1)) for r in env.rooms])  # R,4,2
    heights = np.array([r.wall_height for r in env.rooms])
    pmax = max(1, max(len(p) for r in env.rooms for p in r.portals))
    portals = np.full((R, 4, pmax, 4), np.nan)
    pcount = np.zeros((R, 4), dtype=np.int32)
    for i, r in enumerate(env.rooms):
        for e in range(r.num_walls):
            pcount[i, e] = len(r.portals[e])
            for k, p in enumerate(r.portals[e]):
                portals[i, e, k] = [p["start_pos"], p["end_pos"], p["min_y"], p["max_y"]]
    segs = np.asarray(env.wall_segs)
    d = {
        "outline": outline, "wall_height": heights, "portals": portals, "portal_count": pcount,
        "wall_segs": np.concatenate([segs[:, 0, [0, 2]], segs[:, 1, [0, 2]]], axis=1),
        "segs_y_all_zero": np.array(bool(np.all(segs[:, :, 1] == 0))),
        "room_probs": np.asarray(env.room_probs),
        "quad_offsets": np.cumsum([0] + [r.wall_verts.shape[0] // 4 for r in env.rooms]).astype(np.int32),
        "wall_verts": np.concatenate([r.wall_verts for r in env.rooms]),
        "wall_norms": np.concatenate([r.wall_norms for r in env.rooms]),
        "wall_texcs": np.concatenate([r.wall_texcs for r in env.rooms]).astype(np.float32),
        "floor_texcs": np.stack([pad4(r.floor_texcs) for r in env.rooms]),
        "ceil_texcs": np.stack([pad4(r.ceil_texcs) for r in env.rooms]),
        "tex_names": np.array([[tex_basename(r.wall_tex), tex_basename(r.floor_tex), tex_basename(r.ceil_tex)]
                               for r in env.rooms]),
        "tex_width": np.array([[r.wall_tex.width, r.floor_tex.width, r.ceil_tex.width] for r in env.rooms],
                              dtype=np.int32),
    }
    if any(r.num_walls != 4 for r in env.rooms):
        d["n_edges"] = np.array([r.num_walls for r in env.rooms], dtype=np.int32)
        d["edge_dirs"] = np.stack([pad4(r.edge_dirs[:, [0, 2]]) for r in env.rooms])
        d["edge_norms"] = np.stack([pad4(r.edge_norms[:, [0, 2]]) for r in env.rooms])
    d.update(snapshot_entities(env))
    return d


def first_box(env):
    if hasattr(env, "box1"):
        return env.box1
    return env.box if hasattr(env, "box") else env.red_box


def snapshot_entities(env):
    a = env.agent
    box = first_box(env)
    extra = {}
    if hasattr(env, "blue_box"):
        b2 = env.blue_box
        extra = {"box2_pos": np.array(b2.pos, dtype=float), "box2_dir": np.array(float(b2.dir)),
                 "box2_color": np.array(b2.color_vec, dtype=float),
                 "goal_idx": np.array(int(env.goal_box_idx))}
    elif hasattr(env, "current_goal"):
        extra = {"goal_idx": np.array(int(env.current_goal))}
    if hasattr(env, "box2"):   # SimToRealPush: the yellow box
        b2 = env.box2
        extra = {"box2_pos": np.array(b2.pos, dtype=float), "box2_dir": np.array(float(b2.dir)),
                 "box2_color": np.array(b2.color_vec, dtype=float), "box2_size": np.array(float(b2.size[0])),
                 "goal_dist": np.array(float(env.goal_dist))}
    if type(env).__name__.startswith("SimToReal"):
        extra.update({"box_size": np.array(float(box.size[0])), "agent_radius": np.array(float(a.radius))})
    return {
        **extra,
        "box_pos": np.array(box.pos, dtype=float), "box_dir": np.array(float(box.dir)),
        "box_color": np.array(box.color_vec, dtype=float),
        "agent_pos": np.array(a.pos, dtype=float), "agent_dir": np.array(float(a.dir)),
        "cam": np.array([a.cam_height, a.cam_fwd_disp, a.cam_pitch, a.cam_fov_y], dtype=float),
        "sky_color": np.array(env.sky_color, dtype=float), "light_pos": np.array(env.light_pos, dtype=float),
        "light_color": np.array(env.light_color, dtype=float),
        "light_ambient": np.array(env.light_ambient, dtype=float),
        "rng": rng_fingerprint(env),
    }


def choose_action(env, policy, arng):
    if policy == "random":
        return int(arng.integers(0, env.action_space.n))
    if policy == "push":   # SimToRealPush: get behind the red box as seen from the yellow one, then drive at it
        b1, b2, a = env.box1.pos, env.box2.pos, env.agent
        away = (b1 - b2) / max(np.linalg.norm(b1 - b2), 1e-9)
        stage = b1 + away * 0.3
        tgt = b1 if np.linalg.norm(a.pos - stage) < 0.12 or np.dot(a.pos - b1, away) > 0.2 else stage
        want = math.atan2(-(tgt[2] - a.pos[2]), tgt[0] - a.pos[0])
        diff = (want - a.dir + math.pi) % (2 * math.pi) - math.pi
        if arng.random() < 0.05:
            return int(arng.integers(0, 4))
        if abs(diff) > math.radians(12):
            return 0 if diff > 0 else 1
        return 2
    if policy == "forward":
        return 2 if arng.random() < 0.9 else int(arng.integers(0, 2))
    # greedy: turn toward the box, then walk
    a = env.agent
    if policy == "greedy_red":
        b = env.red_box
    elif policy == "greedy_blue":
        b = env.blue_box
    elif hasattr(env, "boxes"):
        b = env.boxes[env.goal_box_idx]
    else:
        b = first_box(env)
    tgt = b.pos
    if type(env).__name__.startswith("TMaze") and a.pos[0] < 9.2:
        tgt = np.array([10.0, 0.0, 0.0])   # leave the stem of the T before heading for the arm
    if type(env).__name__.startswith("YMaze") and a.pos[0] < 0.3:
        tgt = np.array([0.7, 0.0, 0.0])    # through the hub of the Y first
    want = math.atan2(-(tgt[2] - a.pos[2]), tgt[0] - a.pos[0])
    diff = (want - a.dir + math.pi) % (2 * math.pi) - math.pi
    if abs(diff) > math.radians(10):
        return 0 if diff > 0 else 1
    return 2


def construct(cls, kwargs, dr):
    if cls.__name__.startswith("SimToReal"):
        return cls(**kwargs)
    if dr:
        return cls(domain_rand=True, **kwargs)
    return cls(**kwargs)


def run_case(cls, kwargs, seed, dr, policy, n_steps):
    env = construct(cls, kwargs, dr)
    env.seed(seed)
    env.reset()
    out = {"reset0/" + k: v for k, v in snapshot_world(env).items()}
    arng = np.random.default_rng(1000 + seed)
    actions = np.zeros(n_steps, dtype=np.int32)
    pos = np.zeros((n_steps, 2))
    dirs = np.zeros(n_steps)
    rew = np.zeros(n_steps)
    done = np.zeros(n_steps, dtype=np.uint8)
    stepc = np.zeros(n_steps, dtype=np.int32)
    rngpos = np.zeros((n_steps, 5), dtype=np.int64)
    campos = np.zeros((n_steps, 3))
    camdir = np.zeros((n_steps, 3))
    # state after the worker's auto-reset (subproc_vec_env.py worker: `if done: ob = env.reset()`)
    post = {k: [] for k in ("step", "agent_pos", "agent_dir", "box_pos", "box_dir", "box_color", "cam",
                            "sky_color", "light_pos", "light_color", "light_ambient", "rng", "n_rooms",
                            "segs_sum", "box2_pos", "box2_dir", "box2_color", "goal_idx", "box_size", "box2_size",
                            "agent_radius", "goal_dist")}
    boxes_t = np.zeros((n_steps, 2, 3))   # per-step (x, z, dir) of both boxes: SimToRealPush moves them
    feat = np.zeros((n_steps, 2))
    goal_pos = np.zeros((n_steps, 2))
    for t in range(n_steps):
        a = choose_action(env, policy, arng)
        actions[t] = a
        _, r, d, info = env.step(a)
        if "feature" in info:
            feat[t] = info["feature"]
        if "goal_pos" in info:
            goal_pos[t] = np.asarray(info["goal_pos"])[[0, 2]]
        fb = first_box(env)
        boxes_t[t, 0] = [fb.pos[0], fb.pos[2], fb.dir]
        if hasattr(env, "box2"):
            boxes_t[t, 1] = [env.box2.pos[0], env.box2.pos[2], env.box2.dir]
        pos[t] = env.agent.pos[[0, 2]]
        dirs[t] = env.agent.dir
        rew[t] = r
        done[t] = d
        stepc[t] = env.step_count
        rngpos[t] = rng_fingerprint(env)
        campos[t] = env.agent.cam_pos
        camdir[t] = env.agent.cam_dir
        if d:
            env.reset()
            s = snapshot_entities(env)
            post["step"].append(t)
            for k in s:
                post[k].append(s[k])
            post["n_rooms"].append(len(env.rooms))
            post["segs_sum"].append(float(np.sum(env.wall_segs)))
    if hasattr(env, "boxes") or hasattr(env, "goal_pos"):
        out.update({"traj/feature": feat, "traj/goal_pos": goal_pos})
    if type(env).__name__.startswith("SimToReal"):
        out["traj/boxes"] = boxes_t
    out.update({"traj/actions": actions, "traj/pos": pos, "traj/dir": dirs, "traj/reward": rew,
                "traj/done": done, "traj/step_count": stepc, "traj/rng": rngpos,
                "traj/cam_pos": campos, "traj/cam_dir": camdir})
    for k, v in post.items():
        out["post/" + k] = np.array(v) if len(v) else np.zeros((0,))
    out["final/outline"] = snapshot_world(env)["outline"]
    out["meta/max_episode_steps"] = np.array(env.max_episode_steps)
    out["meta/max_forward_step"] = np.array(env.max_forward_step)
    return out


# ---------------------------------------------------------------------------------------------------------------
# Tasks with several boxes and the carry actions: PutNext.  Every box of the entity list is
# recorded at every step (position incl. the carried height, heading), plus agent.carrying as an index.
CARRY_TASKS = {"PutNext": (PutNext, {})}
CARRY_PLAN = {"PutNext": [(0, 0, "putnext", 700), (1, 1, "putnext", 700), (2, 0, "random", 600), (3, 1, "random", 500),
                          (4, 0, "grab", 500)]}


def carry_index(env):
    c = env.agent.carrying
    return -1 if c is None else next(i for i, e in enumerate(env.entities) if e is c)


def snapshot_boxes(env):
    boxes = [e for e in env.entities if e is not env.agent]
    a = env.agent
    return {
        "boxes_pos": np.array([b.pos for b in boxes], dtype=float), "boxes_dir": np.array([float(b.dir) for b in boxes]),
        "boxes_color": np.array([b.color_vec for b in boxes], dtype=float), "boxes_size": np.array([float(b.size[0]) for b in boxes]),
        "boxes_radius": np.array([float(b.radius) for b in boxes]), "boxes_height": np.array([float(b.height) for b in boxes]),
        "carrying": np.array(carry_index(env)),
        "agent_pos": np.array(a.pos, dtype=float), "agent_dir": np.array(float(a.dir)),
        "cam": np.array([a.cam_height, a.cam_fwd_disp, a.cam_pitch, a.cam_fov_y], dtype=float),
        "sky_color": np.array(env.sky_color, dtype=float), "light_pos": np.array(env.light_pos, dtype=float),
        "light_color": np.array(env.light_color, dtype=float), "light_ambient": np.array(env.light_ambient, dtype=float),
        "rng": rng_fingerprint(env), "wall_segs_sum": np.array(float(np.sum(env.wall_segs))),
        "tex_names": np.array([[tex_basename(r.wall_tex), tex_basename(r.floor_tex), tex_basename(r.ceil_tex)] for r in env.rooms]),
    }


def carry_action(env, policy, arng):
    A = env.actions
    if policy == "random" or arng.random() < 0.06:
        return int(arng.integers(0, env.action_space.n))
    a = env.agent

    def steer(tgt, stop_dist):
        want = math.atan2(-(tgt[2] - a.pos[2]), tgt[0] - a.pos[0])
        diff = (want - a.dir + math.pi) % (2 * math.pi) - math.pi
        if abs(diff) > math.radians(9):
            return int(A.turn_left if diff > 0 else A.turn_right), False
        d = math.hypot(tgt[0] - a.pos[0], tgt[2] - a.pos[2])
        return int(A.move_forward), d < stop_dist
    if policy == "grab":   # pick up whatever is closest, carry it around (turns and back-ups included), drop it
        if a.carrying is None:
            b = min((e for e in env.entities if e is not a), key=lambda e: np.linalg.norm(e.pos - a.pos))
            act, close = steer(b.pos, a.radius + b.radius + 0.25)
            return int(A.pickup) if close else act
        r = arng.random()
        return int(A.drop) if r < 0.04 else int(arng.choice([A.turn_left, A.turn_right, A.move_forward, A.move_forward, A.move_back]))
    # putnext: fetch the red box, carry it to the yellow one, put it down next to it
    red, yel = env.red_box, env.yellow_box
    if a.carrying is None:
        act, close = steer(red.pos, a.radius + red.radius + 0.25)
        return int(A.pickup) if close else act
    if a.carrying is not red:
        return int(A.drop)
    if env.near(red, yel):
        return int(A.drop)
    act, _ = steer(yel.pos, 0.0)
    return act


def run_carry_case(cls, kwargs, seed, dr, policy, n_steps):
    env = construct(cls, kwargs, dr)
    env.seed(seed)
    env.reset()
    nb = len(env.entities) - 1
    out = {"reset0/" + k: v for k, v in snapshot_boxes(env).items()}
    out["reset0/wall_segs"] = np.concatenate([np.asarray(env.wall_segs)[:, 0, [0, 2]], np.asarray(env.wall_segs)[:, 1, [0, 2]]], axis=1)
    arng = np.random.default_rng(2000 + seed)
    rec = {"actions": np.zeros(n_steps, np.int32), "pos": np.zeros((n_steps, 3)), "dir": np.zeros(n_steps), "reward": np.zeros(n_steps),
           "done": np.zeros(n_steps, np.uint8), "step_count": np.zeros(n_steps, np.int32), "rng": np.zeros((n_steps, 5), np.int64),
           "boxes_pos": np.zeros((n_steps, nb, 3)), "boxes_dir": np.zeros((n_steps, nb)), "carrying": np.zeros(n_steps, np.int32),
           "cam_pos": np.zeros((n_steps, 3)), "cam_dir": np.zeros((n_steps, 3))}
    post = []
    pickups = drops = 0
    for t in range(n_steps):
        a = carry_action(env, policy, arng)
        before = carry_index(env)
        _, r, d, _ = env.step(a)
        after = carry_index(env)
        pickups += before < 0 <= after
        drops += after < 0 <= before
        rec["actions"][t] = a
        rec["pos"][t] = env.agent.pos; rec["dir"][t] = env.agent.dir
        rec["reward"][t] = r; rec["done"][t] = d; rec["step_count"][t] = env.step_count
        rec["rng"][t] = rng_fingerprint(env)
        boxes = [e for e in env.entities if e is not env.agent]
        rec["boxes_pos"][t] = [b.pos for b in boxes]; rec["boxes_dir"][t] = [b.dir for b in boxes]
        rec["carrying"][t] = after
        rec["cam_pos"][t] = env.agent.cam_pos; rec["cam_dir"][t] = env.agent.cam_dir
        if d:
            env.reset()
            post.append((t, snapshot_boxes(env)))
    for k, v in rec.items():
        out["traj/" + k] = v
    out["post/step"] = np.array([t for t, _ in post], dtype=np.int64)
    for k in (post[0][1] if post else {}):
        out["post/" + k] = np.array([s[k] for _, s in post])
    out["meta/max_episode_steps"] = np.array(env.max_episode_steps)
    out["meta/max_forward_step"] = np.array(env.max_forward_step)
    out["meta/n_actions"] = np.array(env.action_space.n)
    out["meta/pickups_drops"] = np.array([pickups, drops])
    return out


def parse_gl_log(log):
    polys, lights, misc = [], {}, {}
    cur = {"color": None, "normal": None, "texc": None, "tex_on": False, "mode": None, "xform": []}
    poly = None
    gl = sys.modules["pyglet.gl"]
    names = {getattr(gl, n): n for n in dir(gl) if n.startswith("GL_")}
    for name, args in log:
        if name == "glLightfv":
            lights[names[args[1]]] = list(args[2])
        elif name == "glEnable" and names.get(args[0]) == "GL_TEXTURE_2D":
            cur["tex_on"] = True
        elif name == "glDisable" and names.get(args[0]) == "GL_TEXTURE_2D":
            cur["tex_on"] = False
        elif name == "glColor3f":
            cur["color"] = [float(x) for x in args]
        elif name == "glNormal3f":
            cur["normal"] = [float(x) for x in args]
        elif name == "glTexCoord2f":
            cur["texc"] = [float(x) for x in args]
        elif name == "glTranslatef":
            cur["xform"].append(["translate"] + [float(x) for x in args])
        elif name == "glRotatef":
            cur["xform"].append(["rotate"] + [float(x) for x in args])
        elif name == "glPushMatrix":
            cur["xform"] = []
        elif name == "glPopMatrix":
            cur["xform"] = []
        elif name == "glBegin":
            poly = {"mode": names[args[0]], "color": cur["color"], "tex_on": cur["tex_on"],
                    "xform": list(cur["xform"]), "verts": [], "texcs": [], "norms": []}
        elif name == "glVertex3f":
            poly["verts"].append([float(x) for x in args])
            poly["texcs"].append(cur["texc"])
            poly["norms"].append(cur["normal"])
        elif name == "glEnd":
            polys.append(poly)
            poly = None
        elif name in ("gluPerspective", "gluLookAt", "glClearColor", "glClearDepth", "glOrtho"):
            misc[name] = [float(x) for x in args]
        elif name == "glLoadMatrixf":
            misc[name] = [float(x) for x in args[0]]
    return polys, lights, misc


def capture_gl_carry(cls, kwargs, seed, dr, policy="putnext", max_steps=400):
    """A frame with a CARRIED box: reset() under the recorder gives the rooms and lights (they live in the display list
    _render_static compiles, miniworld.py:1014-1057); the agent then fetches a box (unrecorded steps of the fixture
    policy); one more render_obs() under the recorder gives the boxes - the carried one translated to its carry height
    and turned with the agent - and the camera (Box.render entity.py:385-408, miniworld.py:1076-1083)."""
    env = construct(cls, kwargs, dr)
    env.seed(seed)
    deps.GL_LOG.clear()
    deps.GL_LOG_ENABLED[0] = True
    env.reset()
    deps.GL_LOG_ENABLED[0] = False
    reset_log = list(deps.GL_LOG)
    deps.GL_LOG.clear()
    arng = np.random.default_rng(3000 + seed)
    actions = []
    for t in range(max_steps):
        a = carry_action(env, policy, arng)
        actions.append(a)
        _, _, d, _ = env.step(a)
        assert not d
        if env.agent.carrying is not None and t > 5 and actions[-3:].count(int(env.actions.move_forward)) >= 2:
            break
    assert env.agent.carrying is not None
    deps.GL_LOG_ENABLED[0] = True
    env.render_obs()
    deps.GL_LOG_ENABLED[0] = False
    frame_log = list(deps.GL_LOG)
    deps.GL_LOG.clear()
    rooms, lights, _ = parse_gl_log(reset_log)
    boxes, _, misc = parse_gl_log(frame_log)
    nb = len(env.entities) - 1
    assert len(boxes) == nb and all(not p["tex_on"] for p in boxes)
    st = snapshot_boxes(env)
    return {
        "task": cls.__name__, "kwargs": kwargs, "seed": seed, "domain_rand": int(dr), "actions": [int(a) for a in actions],
        "lights": lights, "misc": misc, "polys": rooms[:len(rooms) - nb] + boxes,
        "room_tex": [[tex_basename(r.wall_tex), tex_basename(r.floor_tex), tex_basename(r.ceil_tex)] for r in env.rooms],
        "agent_pos": st["agent_pos"].tolist(), "agent_dir": float(st["agent_dir"]), "cam": st["cam"].tolist(),
        "boxes_pos": st["boxes_pos"].tolist(), "boxes_dir": st["boxes_dir"].tolist(), "boxes_color": st["boxes_color"].tolist(),
        "boxes_size": st["boxes_size"].tolist(), "carrying": int(st["carrying"]),
    }


def capture_gl(cls, kwargs, seed, dr, pose=None):
    """One reset() under the call recorder -> structured polygon list (renderer *input* parity).
    With pose = (x, z, dir) the agent is then moved there and render_obs() is recorded again: the
    camera calls of that second frame replace the first frame's (the polygons are the same)."""
    env = construct(cls, kwargs, dr)
    env.seed(seed)
    deps.GL_LOG.clear()
    deps.GL_LOG_ENABLED[0] = True
    env.reset()
    deps.GL_LOG_ENABLED[0] = False
    log = list(deps.GL_LOG)
    deps.GL_LOG.clear()
    if pose is not None:
        env.agent.pos = np.array([pose[0], 0.0, pose[1]])
        env.agent.dir = pose[2]
        deps.GL_LOG_ENABLED[0] = True
        env.render_obs()
        deps.GL_LOG_ENABLED[0] = False
        log += [(n, a) for (n, a) in deps.GL_LOG if n in ("gluPerspective", "gluLookAt")]
        deps.GL_LOG.clear()
    polys, lights, misc = parse_gl_log(log)
    ents = snapshot_entities(env)
    return {
        "task": cls.__name__, "kwargs": kwargs, "seed": seed, "domain_rand": int(dr),
        "lights": lights, "misc": misc, "polys": polys,
        "room_tex": [[tex_basename(r.wall_tex), tex_basename(r.floor_tex), tex_basename(r.ceil_tex)]
                     for r in env.rooms],
        "agent_pos": ents["agent_pos"].tolist(), "agent_dir": float(ents["agent_dir"]),
        "box_pos": ents["box_pos"].tolist(), "box_dir": float(ents["box_dir"]),
        "box_color": ents["box_color"].tolist(), "cam": ents["cam"].tolist(),
        **({"box2_pos": ents["box2_pos"].tolist(), "box2_dir": float(ents["box2_dir"]),
            "box2_color": ents["box2_color"].tolist()} if "box2_pos" in ents else {}),
    }


def capture_gl_top(cls, kwargs, seed, dr, pose=None):
    """render_top_view() (miniworld.py:1087-1158) under the call recorder: the room polygons of the display list (recorded
    when reset() compiled it), then this frame's glOrtho / glLoadMatrixf, the non-static entities and the agent's triangle -
    drawn with whatever normal the last box face left current (entity.py:494-514)."""
    env = construct(cls, kwargs, dr)
    env.seed(seed)
    deps.GL_LOG.clear()
    deps.GL_LOG_ENABLED[0] = True
    env.reset()
    deps.GL_LOG_ENABLED[0] = False
    reset_polys, lights, misc = parse_gl_log(list(deps.GL_LOG))
    deps.GL_LOG.clear()
    if pose is not None:
        env.agent.pos = np.array([pose[0], 0.0, pose[1]])
        env.agent.dir = pose[2]
    deps.GL_LOG_ENABLED[0] = True
    env.render_top_view()
    deps.GL_LOG_ENABLED[0] = False
    top_polys, _, top_misc = parse_gl_log(list(deps.GL_LOG))
    deps.GL_LOG.clear()
    ents = snapshot_entities(env)
    boxes = [e for e in env.entities if e is not env.agent]
    return {
        "task": cls.__name__, "kwargs": kwargs, "seed": seed, "domain_rand": int(dr),
        "lights": lights, "misc": {**misc, **top_misc},
        "polys": [p for p in reset_polys if p["tex_on"]] + top_polys,   # rooms (textured), then this frame's entities + agent
        "room_tex": [[tex_basename(r.wall_tex), tex_basename(r.floor_tex), tex_basename(r.ceil_tex)] for r in env.rooms],
        "extents": [float(env.min_x), float(env.max_x), float(env.min_z), float(env.max_z)],
        "agent_pos": ents["agent_pos"].tolist(), "agent_dir": float(ents["agent_dir"]),
        "boxes_pos": [np.array(b.pos, dtype=float).tolist() for b in boxes], "boxes_dir": [float(b.dir) for b in boxes],
    }


def math_kat():
    """Known-answer vectors for the two importable numpy-only reference modules (math.py)."""
    rng = np.random.default_rng(7)
    n = 256
    segs = rng.uniform(-5, 5, size=(n, 16, 2, 3))
    segs[:, :, :, 1] = 0
    pts = rng.uniform(-5, 5, size=(n, 3))
    rad = rng.uniform(0.05, 1.5, size=n)
    hit = np.array([bool(ref_math.intersect_circle_segs(pts[i], rad[i], segs[i])) for i in range(n)])
    ang = rng.uniform(-7, 7, size=n)
    axes = np.array([ref_math.Y_VEC, ref_math.Z_VEC, ref_math.X_VEC])
    rots = np.stack([[ref_math.gen_rot_matrix(ax, a) for ax in axes] for a in ang])
    return {"segs": segs, "pts": pts, "rad": rad, "hit": hit, "ang": ang, "rots": rots}


# render_top_view() streams: polygon rooms with the culled connectors, rectangles with portals + domain randomisation,
# six boxes; the agent moved off its spawn pose in two of them
TOP_VIEWS = [("YMaze", 0, None), ("FourRooms", 1, (2.5, -3.0, 0.7)), ("PutNext", 1, (6.0, 6.0, -2.0))]


def main():
    global HERE
    args = sys.argv[1:]
    if "--out" in args:   # write somewhere else (tests/test_fixtures_regenerate.py diffs that against the committed files)
        i = args.index("--out")
        HERE = args[i + 1]
        os.makedirs(HERE, exist_ok=True)
        args = args[:i] + args[i + 2:]
    only = args   # optional task-name prefixes: regenerate just those fixtures
    for task, (cls, kwargs) in TASKS.items():
        if only and not any(task.startswith(o) for o in only):
            continue
        blob = {}
        for (seed, dr, policy, n) in PLAN[task]:
            case = run_case(cls, kwargs, seed, dr, policy, n)
            tag = "s%d_dr%d_%s/" % (seed, dr, policy)
            for k, v in case.items():
                blob[tag + k] = v
            print(task, tag, "dones:", int(case["traj/done"].sum()),
                  "rewards>0:", int((case["traj/reward"] > 0).sum()))
        np.savez_compressed(os.path.join(HERE, "state_%s.npz" % task), **blob)
        for dr in (0, 1):
            if task.startswith("Maze") and (dr == 1) != (task == "MazeS3"):
                continue
            if task in ("MazeR2C4", "OneRoomS6", "Hallway6"):
                continue
            if task.startswith("TMaze") and not ((task == "TMaze" and dr == 0) or (task == "TMazeTwoBoxFeatures" and dr == 1)):
                continue
            if task.startswith("SimToReal") and dr == 0:
                continue
            if task in ("YMazeLeft", "YMazeRight"):
                continue
            # the two-box scene is captured from inside the bar of the T, with both boxes in view
            pose = (11.5, 7.6, 2.13) if task == "TMazeTwoBoxFeatures" else None
            if task == "YMaze" and dr == 1:
                pose = (0.4, 0.3, 1.0)   # from inside the triangular hub, looking up the left arm past a sliver-shaped connector
            g = capture_gl(cls, kwargs, 1, dr, pose)
            g["posed"] = pose is not None
            with open(os.path.join(HERE, "glstream_%s_dr%d.json" % (task, dr)), "w") as fh:
                json.dump(g, fh, separators=(",", ":"))
    for task, (cls, kwargs) in CARRY_TASKS.items():
        if only and not any(task.startswith(o) for o in only):
            continue
        blob = {}
        for (seed, dr, policy, n) in CARRY_PLAN[task]:
            case = run_carry_case(cls, kwargs, seed, dr, policy, n)
            tag = "s%d_dr%d_%s/" % (seed, dr, policy)
            for k, v in case.items():
                blob[tag + k] = v
            print(task, tag, "dones:", int(case["traj/done"].sum()), "rewards>0:", int((case["traj/reward"] > 0).sum()),
                  "pickups/drops:", case["meta/pickups_drops"].tolist())
        np.savez_compressed(os.path.join(HERE, "state_%s.npz" % task), **blob)
        for dr in (0, 1):
            g = capture_gl_carry(cls, kwargs, 1, dr)
            with open(os.path.join(HERE, "glstream_%s_dr%d.json" % (task, dr)), "w") as fh:
                json.dump(g, fh, separators=(",", ":"))
    for task, dr, pose in TOP_VIEWS:
        if only and not any(task.startswith(o) for o in only):
            continue
        cls, kwargs = {**TASKS, **CARRY_TASKS}[task]
        with open(os.path.join(HERE, "gltop_%s_dr%d.json" % (task, dr)), "w") as fh:
            json.dump(capture_gl_top(cls, kwargs, 1, dr, pose), fh, separators=(",", ":"))
    if only:
        return
    np.savez_compressed(os.path.join(HERE, "math_kat.npz"), **math_kat())
    # the assumed gym seed hashing, exposed as data so the product/oracle can be checked without gym
    seeds = [0, 1, 2, 3, 7, 42, 4097, 65535, 2**32 + 5, 2**63 + 11]
    with open(os.path.join(HERE, "seed_keys.json"), "w") as fh:
        json.dump({str(s): deps.seed_to_mt_key(s) for s in seeds}, fh)


if __name__ == "__main__":
    main()
