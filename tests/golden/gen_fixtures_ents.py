#!/usr/bin/env python3
"""Golden vectors for the tasks with a general entity list: mesh entities (Key, Ball, MeshEnt), image / text
frames, entities that leave the list (PickupObjs) or re-enter it at its end (CollectHealth), fixed-position placements.
Same method as gen_fixtures.py: the UNMODIFIED reference (/root/reference) runs under the inert `gym` / `pyglet` stand-ins
of _inert_deps.py; what is kept is data only - the reference's own state arithmetic and the exact inputs it hands to OpenGL.

Run:  python tests/golden/gen_fixtures_ents.py [--out DIR] [task prefixes ...]     (build container only)
Outputs: state_<task>.npz, glstream_<task>_dr<k>.json, meshes.json

NumPy note.  MeshEnt computes `scale = height / sy` and `radius = math.sqrt(sx*sx + sz*sz) * scale` from float32 mesh extents
(entity.py:118-127): under the NumPy of this container (>= 2, NEP 50 promotion) both are float32 scalars, and sums such as
`self.agent.radius + ent.radius` (Python float + float32) are evaluated in float32.  The fixtures record the values AND their
scalar types (`ents_radius_f32`), the restatements follow them.
"""
import hashlib
import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_fixtures as G  # noqa: E402  (installs the stand-ins, imports the reference)
from gen_fixtures import deps  # noqa: E402

from gym_miniworld.entity import Box, ImageFrame, MeshEnt, TextFrame  # noqa: E402
from gym_miniworld.envs import CollectHealth, PickupObjs, RoomObjs, Sidewalk, Sign, ThreeRooms, WallGap  # noqa: E402
from gym_miniworld.objmesh import ObjMesh  # noqa: E402

OUT = HERE
ENT_TASKS = {
    "PickupObjs": (PickupObjs, {}), "RoomObjs": (RoomObjs, {}), "CollectHealth": (CollectHealth, {}),
    "ThreeRooms": (ThreeRooms, {}), "Sign": (Sign, {}), "SignGreenKey": (Sign, {"color_index": 1, "goal": 1}),
    "Sidewalk": (Sidewalk, {}), "WallGap": (WallGap, {}),
    "PickupObjsS8N7": (PickupObjs, {"size": 8, "num_objs": 7}),
}
# (seed, domain_rand, policy, n_steps)
ENT_PLAN = {
    "PickupObjs": [(0, 0, "collect", 900), (1, 1, "collect", 900), (2, 0, "random", 700), (3, 1, "random", 500)],
    "PickupObjsS8N7": [(0, 0, "collect", 500), (1, 1, "random", 400)],
    "RoomObjs": [(0, 0, "collect", 500), (1, 1, "random", 500), (2, 0, "random", 400)],
    "CollectHealth": [(0, 0, "collect", 900), (1, 1, "collect", 700), (2, 0, "random", 400), (3, 1, "random", 300)],
    "ThreeRooms": [(0, 0, "random", 600), (1, 1, "wander", 700), (2, 0, "wander", 500)],
    "Sign": [(0, 0, "random", 300), (1, 0, "touch", 300), (2, 0, "wander", 300)],
    "SignGreenKey": [(0, 0, "touch", 300), (1, 0, "random", 200)],
    "Sidewalk": [(0, 0, "greedy", 500), (1, 1, "greedy", 500), (2, 0, "random", 500), (3, 1, "wander", 400)],
    "WallGap": [(0, 0, "greedy", 700), (1, 1, "greedy", 700), (2, 0, "random", 500)],
}
NO_DR_KWARG = {"Sign", "SignGreenKey"}   # Sign.__init__ takes no domain_rand (it forces False)


def construct(cls, kwargs, dr):
    return cls(domain_rand=True, **kwargs) if dr else cls(**kwargs)


def kind_of(e):
    return 0 if isinstance(e, Box) else 1 if isinstance(e, MeshEnt) else 2 if isinstance(e, ImageFrame) else 3 if isinstance(e, TextFrame) else -1


def mesh_name_of(e):
    for path, m in ObjMesh.cache.items():
        if m is e.mesh:
            return os.path.splitext(os.path.basename(path))[0]
    raise KeyError


def describe_entities(env):
    """the entity list but the agent, in list order: everything that is fixed for the episode"""
    ents = [e for e in env.entities if e is not env.agent]
    n = len(ents)
    d = {
        "ents_kind": np.array([kind_of(e) for e in ents], np.int32),
        "ents_mesh": np.array([mesh_name_of(e) if isinstance(e, MeshEnt) else "" for e in ents]),
        "ents_color": np.array([e.color if isinstance(e, Box) else "" for e in ents]),
        "ents_radius": np.array([float(e.radius) for e in ents]),
        "ents_radius_f32": np.array([isinstance(e.radius, np.float32) for e in ents]),
        "ents_height": np.array([float(e.height) for e in ents]),
        "ents_height_f32": np.array([isinstance(e.height, np.float32) for e in ents]),
        "ents_scale": np.array([float(e.scale) if isinstance(e, MeshEnt) else np.nan for e in ents]),
        "ents_scale_f32": np.array([isinstance(getattr(e, "scale", None), np.float32) for e in ents]),
        "ents_static": np.array([bool(e.is_static) for e in ents]),
        "ents_size": np.array([np.asarray(e.size, float) if isinstance(e, Box)
                               else [e.depth, e.height, e.width] if isinstance(e, (ImageFrame, TextFrame)) else [np.nan] * 3 for e in ents]).reshape(n, 3),
        "ents_color_vec": np.array([np.asarray(e.color_vec, float) if isinstance(e, Box) else [np.nan] * 3 for e in ents]).reshape(n, 3),
        "ents_tex": np.array([G.tex_basename(e.tex) if isinstance(e, ImageFrame) else "" for e in ents]),
        "ents_text": np.array([e.str if isinstance(e, TextFrame) else "" for e in ents]),
        "ents_text_tex": np.array(["|".join(G.tex_basename(t) if t else "" for t in e.texs) if isinstance(e, TextFrame) else "" for e in ents]),
        "ents_pos": np.array([np.asarray(e.pos, float) for e in ents]).reshape(n, 3),
        "ents_dir": np.array([float(e.dir) for e in ents]),
    }
    return ents, d


def snapshot(env):
    a = env.agent
    _, d = describe_entities(env)
    segs = np.asarray(env.wall_segs)
    d.update({
        "agent_pos": np.array(a.pos, float), "agent_dir": np.array(float(a.dir)), "agent_radius": np.array(float(a.radius)),
        "cam": np.array([a.cam_height, a.cam_fwd_disp, a.cam_pitch, a.cam_fov_y], float),
        "sky_color": np.array(env.sky_color, float), "light_pos": np.array(env.light_pos, float),
        "light_color": np.array(env.light_color, float), "light_ambient": np.array(env.light_ambient, float),
        "rng": G.rng_fingerprint(env),
        "wall_segs": np.concatenate([segs[:, 0, [0, 2]], segs[:, 1, [0, 2]]], axis=1),
        "room_probs": np.asarray(env.room_probs),
        "outline": np.stack([np.stack([r.outline[:, 0], r.outline[:, 2]], axis=1) for r in env.rooms]),
        "wall_height": np.array([r.wall_height for r in env.rooms]),
        "no_ceiling": np.array([bool(r.no_ceiling) for r in env.rooms]),
        "tex_names": np.array([[G.tex_basename(r.wall_tex), G.tex_basename(r.floor_tex), G.tex_basename(r.ceil_tex)] for r in env.rooms]),
        "tex_width": np.array([[r.wall_tex.width, r.floor_tex.width, r.ceil_tex.width] for r in env.rooms], np.int32),
        "tex_height": np.array([[r.wall_tex.height, r.floor_tex.height, r.ceil_tex.height] for r in env.rooms], np.int32),
        "quad_offsets": np.cumsum([0] + [r.wall_verts.shape[0] // 4 for r in env.rooms]).astype(np.int32),
        "wall_verts": np.concatenate([r.wall_verts for r in env.rooms]),
        "wall_norms": np.concatenate([r.wall_norms for r in env.rooms]),
        "wall_texcs": np.concatenate([r.wall_texcs for r in env.rooms]).astype(np.float32),
        "floor_texcs": np.stack([r.floor_texcs for r in env.rooms]), "ceil_texcs": np.stack([r.ceil_texcs for r in env.rooms]),
    })
    pmax = max(1, max(len(p) for r in env.rooms for p in r.portals))
    portals = np.full((len(env.rooms), 4, pmax, 4), np.nan)
    pcount = np.zeros((len(env.rooms), 4), np.int32)
    for i, r in enumerate(env.rooms):
        for e in range(r.num_walls):
            pcount[i, e] = len(r.portals[e])
            for k, p in enumerate(r.portals[e]):
                portals[i, e, k] = [p["start_pos"], p["end_pos"], p["min_y"], p["max_y"]]
    d["portals"], d["portal_count"] = portals, pcount
    return d


def steer(a, tgt, A, tol_deg=9):
    want = math.atan2(-(tgt[2] - a.pos[2]), tgt[0] - a.pos[0])
    diff = (want - a.dir + math.pi) % (2 * math.pi) - math.pi
    if abs(diff) > math.radians(tol_deg):
        return int(A.turn_left if diff > 0 else A.turn_right)
    return int(A.move_forward)


def choose(env, policy, arng, t):
    A, a = env.actions, env.agent
    n_act = env.action_space.n
    if policy == "random" or arng.random() < 0.05:
        return int(arng.integers(0, n_act))
    if policy == "wander":
        return int(A.move_forward) if arng.random() < 0.8 else int(arng.integers(0, 2))
    movable = [e for e in env.entities if e is not a and not e.is_static]
    if policy == "collect":   # walk up to the nearest object that can be picked up, pick it up
        if not movable:
            return int(arng.integers(0, n_act))
        b = min(movable, key=lambda e: np.linalg.norm(np.asarray(e.pos, float)[[0, 2]] - a.pos[[0, 2]]))
        d = math.hypot(b.pos[0] - a.pos[0], b.pos[2] - a.pos[2])
        act = steer(a, b.pos, A)
        if act == int(A.move_forward) and d < 1.5 * a.radius + 1.2 * a.radius + float(b.radius) - 0.05 and int(A.pickup) < n_act:
            return int(A.pickup)
        return act
    if policy == "touch":   # Sign: head for the object the sign and the goal name
        tgt = env._objects[env._goal][env._color_index]
        return steer(a, tgt.pos, A, tol_deg=23)
    if policy == "greedy":
        tgt = env.box.pos
        if type(env).__name__ == "WallGap" and a.pos[2] > -0.9:   # through the gap in the wall first
            tgt = np.array([0.0, 0.0, 1.2]) if (a.pos[2] > 1.4 or abs(a.pos[0]) > 0.9) and a.pos[2] > 1.0 else np.array([0.0, 0.0, -1.6])
        return steer(a, tgt, A)
    raise KeyError(policy)


def run_case(task, cls, kwargs, seed, dr, policy, n_steps):
    env = construct(cls, kwargs, dr)
    env.seed(seed)
    env.reset()
    out = {"reset0/" + k: v for k, v in snapshot(env).items()}
    ents0, _ = describe_entities(env)
    E = len(ents0)
    arng = np.random.default_rng(5000 + seed)
    rec = {"actions": np.zeros(n_steps, np.int32), "pos": np.zeros((n_steps, 3)), "dir": np.zeros(n_steps), "reward": np.zeros(n_steps),
           "done": np.zeros(n_steps, np.uint8), "step_count": np.zeros(n_steps, np.int32), "rng": np.zeros((n_steps, 5), np.int64),
           "ents_pos": np.full((n_steps, E, 3), np.nan), "ents_dir": np.full((n_steps, E), np.nan), "ents_alive": np.zeros((n_steps, E), np.uint8),
           "order": np.full((n_steps, E + 1), -1, np.int32),   # the entity list after the step as slots of THIS episode's first list; -2 = the agent
           "carrying": np.full(n_steps, -1, np.int32), "health": np.full(n_steps, np.nan), "picked": np.full(n_steps, -1, np.int32),
           "cam_pos": np.zeros((n_steps, 3)), "cam_dir": np.zeros((n_steps, 3))}
    post = []
    ep_ents = list(ents0)
    n_pick = 0
    for t in range(n_steps):
        act = choose(env, policy, arng, t)
        rec["actions"][t] = act
        _, r, d, info = env.step(act)
        a = env.agent
        rec["pos"][t] = a.pos; rec["dir"][t] = a.dir; rec["reward"][t] = r; rec["done"][t] = d; rec["step_count"][t] = env.step_count
        rec["rng"][t] = G.rng_fingerprint(env)
        rec["cam_pos"][t] = a.cam_pos; rec["cam_dir"][t] = a.cam_dir
        slot = {id(e): i for i, e in enumerate(ep_ents)}
        for k, e in enumerate(env.entities):
            rec["order"][t, k] = -2 if e is a else slot[id(e)]
            if e is not a:
                i = slot[id(e)]
                rec["ents_alive"][t, i] = 1; rec["ents_pos"][t, i] = e.pos; rec["ents_dir"][t, i] = e.dir
        rec["carrying"][t] = -1 if a.carrying is None else slot[id(a.carrying)]
        if "health" in info:
            rec["health"][t] = info["health"]
        if hasattr(env, "num_picked_up"):
            rec["picked"][t] = env.num_picked_up
            n_pick = max(n_pick, env.num_picked_up)
        if d:
            env.reset()
            ep_ents, _ = describe_entities(env)
            assert len(ep_ents) == E   # the tasks recorded here build the same number of entities every episode
            post.append((t, snapshot(env)))
    for k, v in rec.items():
        out["traj/" + k] = v
    out["post/step"] = np.array([t for t, _ in post], np.int64)
    for k in (post[0][1] if post else {}):
        vals = [s[k] for _, s in post]
        if all(np.asarray(v).shape == np.asarray(vals[0]).shape for v in vals):
            out["post/" + k] = np.array(vals)
    out["meta/max_episode_steps"] = np.array(float(env.max_episode_steps))
    out["meta/max_forward_step"] = np.array(env.max_forward_step)
    out["meta/n_actions"] = np.array(env.action_space.n)
    print(task, "s%d dr%d %s" % (seed, dr, policy), "dones", int(rec["done"].sum()), "reward sum", float(rec["reward"].sum()),
          "removed/respawned", int((rec["ents_alive"] == 0).any(axis=1).sum()), int((rec["order"][:, -1] >= 0).sum()))
    return out


def digest(a):
    a = np.ascontiguousarray(a)
    return hashlib.sha256(a.tobytes()).hexdigest()


def mesh_table():
    """every mesh the entity tasks load: extents, counts and SHA-256 digests of the float32 arrays objmesh.py hands to
    pyglet.graphics.vertex_list (chunk by chunk, in draw order) - the arrays themselves would be megabytes"""
    out = {}
    for path, m in sorted(ObjMesh.cache.items()):
        name = os.path.splitext(os.path.basename(path))[0]
        chunks = []
        for vl, tex in zip(m.vlists, m.textures):
            chunks.append({"n_verts": int(vl.count), "v3f": digest(vl.data["v3f"]), "t2f": digest(vl.data["t2f"]), "n3f": digest(vl.data["n3f"]),
                           "c3f": digest(vl.data["c3f"]), "v3f_sum": float(vl.data["v3f"].astype(np.float64).sum()),
                           "texture": os.path.basename(deps.TEX_PATHS[tex.id]) if tex is not None else None})
        out[name] = {"min_coords": [float(x) for x in m.min_coords], "max_coords": [float(x) for x in m.max_coords],
                     "coords_dtype": str(np.asarray(m.max_coords).dtype), "chunks": chunks}
    return out


def parse_ent_stream(log):
    """the GL calls of the non-room part of a frame, structured: glBegin polygons (boxes, frames) with their matrix stack, and mesh
    draws (vertex-list id -> mesh name, chunk) with theirs"""
    gl = sys.modules["pyglet.gl"]
    names = {getattr(gl, n): n for n in dir(gl) if n.startswith("GL_")}
    vl_owner = {}
    for path, m in ObjMesh.cache.items():
        for ci, vl in enumerate(m.vlists):
            vl_owner[vl.uid] = (os.path.splitext(os.path.basename(path))[0], ci)
    items, xform, stack = [], [], []
    cur = {"color": None, "normal": None, "texc": None, "tex_on": False, "tex": None}
    poly = None
    for name, args in log:
        if name == "glPushMatrix":
            stack.append(list(xform))
        elif name == "glPopMatrix":
            xform = stack.pop()
        elif name == "glTranslatef":
            xform.append(["translate"] + [float(x) for x in args])
        elif name == "glRotatef":
            xform.append(["rotate"] + [float(x) for x in args])
        elif name == "glScalef":
            xform.append(["scale"] + [float(x) for x in args])
        elif name == "glEnable" and names.get(args[0]) == "GL_TEXTURE_2D":
            cur["tex_on"] = True
        elif name == "glDisable" and names.get(args[0]) == "GL_TEXTURE_2D":
            cur["tex_on"] = False
        elif name == "glBindTexture":
            cur["tex"] = os.path.splitext(os.path.basename(deps.TEX_PATHS[args[1]]))[0] if args[1] in deps.TEX_PATHS and deps.TEX_PATHS[args[1]] else None
        elif name == "glColor3f":
            cur["color"] = [float(x) for x in args]
        elif name == "glNormal3f":
            cur["normal"] = [float(x) for x in args]
        elif name == "glTexCoord2f":
            cur["texc"] = [float(x) for x in args]
        elif name == "glBegin":
            poly = {"type": "poly", "mode": names[args[0]], "tex_on": cur["tex_on"], "tex": cur["tex"] if cur["tex_on"] else None,
                    "xform": list(xform), "verts": [], "texcs": [], "norms": [], "colors": []}
        elif name == "glVertex3f":
            poly["verts"].append([float(x) for x in args]); poly["texcs"].append(cur["texc"]); poly["norms"].append(cur["normal"]); poly["colors"].append(cur["color"])
        elif name == "glEnd":
            items.append(poly); poly = None
        elif name == "vlist_draw":
            mesh, chunk = vl_owner[args[0]]
            items.append({"type": "mesh", "mesh": mesh, "chunk": chunk, "mode": names[args[1]], "tex_on": cur["tex_on"],
                          "tex": cur["tex"] if cur["tex_on"] else None, "xform": list(xform)})
    return items


def capture_gl(task, cls, kwargs, seed, dr, prelude=0, policy="collect"):
    """reset() under the recorder (rooms, lights, STATIC entities: they go into the display list, miniworld.py:1047-1052), then
    `prelude` unrecorded policy steps, then one render_obs() under the recorder (camera + the non-static entities)."""
    env = construct(cls, kwargs, dr)
    env.seed(seed)
    deps.GL_LOG.clear(); deps.GL_LOG_ENABLED[0] = True
    env.reset()
    deps.GL_LOG_ENABLED[0] = False
    reset_log = list(deps.GL_LOG); deps.GL_LOG.clear()
    arng = np.random.default_rng(7000 + seed)
    actions = []
    for t in range(prelude):
        a = choose(env, policy, arng, t)
        _, _, d, _ = env.step(a)
        actions.append(int(a))
        if d or (policy == "collect" and env.agent.carrying is not None and t > 3):
            break   # RoomObjs: the frame shows a carried object
    deps.GL_LOG_ENABLED[0] = True
    env.render_obs()
    deps.GL_LOG_ENABLED[0] = False
    frame_log = list(deps.GL_LOG); deps.GL_LOG.clear()
    # the display list: everything between glNewList and glEndList of the reset's _render_static
    i0 = max(i for i, (n, _) in enumerate(reset_log) if n == "glNewList")
    i1 = max(i for i, (n, _) in enumerate(reset_log) if n == "glEndList")
    static_log = reset_log[i0:i1]
    room_polys, lights, _ = G.parse_gl_log(static_log)
    room_polys = [p for p in room_polys if not p["xform"] and p["tex_on"] and p["mode"] in ("GL_POLYGON", "GL_QUADS") and p["color"] == [1.0, 1.0, 1.0]]
    static_items = [it for it in parse_ent_stream(static_log) if it["xform"]]
    _, _, misc = G.parse_gl_log(frame_log)
    dyn_items = [it for it in parse_ent_stream(frame_log) if it["xform"]]
    st = snapshot(env)
    ents, desc = describe_entities(env)
    return {
        "task": task, "kwargs": kwargs, "seed": seed, "domain_rand": int(dr), "actions": actions,
        "lights": lights, "misc": misc, "room_polys": room_polys, "static_items": static_items, "dynamic_items": dyn_items,
        "room_tex": st["tex_names"].tolist(), "no_ceiling": st["no_ceiling"].tolist(),
        "agent_pos": st["agent_pos"].tolist(), "agent_dir": float(st["agent_dir"]), "cam": st["cam"].tolist(),
        "carrying": -1 if env.agent.carrying is None else next(i for i, e in enumerate(ents) if e is env.agent.carrying),
        "ents": {k[5:]: (v.tolist()) for k, v in desc.items()},
    }


def main():
    global OUT
    args = sys.argv[1:]
    if "--out" in args:
        i = args.index("--out")
        OUT = args[i + 1]
        os.makedirs(OUT, exist_ok=True)
        args = args[:i] + args[i + 2:]
    only = args
    for task, (cls, kwargs) in ENT_TASKS.items():
        if only and not any(task.startswith(o) for o in only):
            continue
        blob = {}
        for (seed, dr, policy, n) in ENT_PLAN[task]:
            case = run_case(task, cls, kwargs, seed, dr, policy, n)
            tag = "s%d_dr%d_%s/" % (seed, dr, policy)
            for k, v in case.items():
                blob[tag + k] = v
        np.savez_compressed(os.path.join(OUT, "state_%s.npz" % task), **blob)
        if task in ("SignGreenKey", "PickupObjsS8N7"):
            continue
        for dr in ((0,) if task in NO_DR_KWARG else (0, 1)):
            # PickupObjs: the first frame, all five objects on the floor (a picked object leaves the list before the next frame);
            # RoomObjs: until something is carried; CollectHealth: after a few respawns; the others after a short walk
            g = capture_gl(task, cls, kwargs, 1, dr, prelude={"PickupObjs": 0, "RoomObjs": 200, "CollectHealth": 120}.get(task, 25),
                           policy="collect" if task in ("PickupObjs", "CollectHealth", "RoomObjs") else "wander")
            with open(os.path.join(OUT, "glstream_%s_dr%d.json" % (task, dr)), "w") as fh:
                json.dump(g, fh, separators=(",", ":"))
    if not only:
        with open(os.path.join(OUT, "meshes.json"), "w") as fh:
            json.dump(mesh_table(), fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
