"""ctypes front-end of the CPU oracle (oracle/mw_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg - never by the product package.  See mw_oracle.h for what is pinned
against the reference and what is "parity unpinned".
"""
import ctypes
import hashlib
import os
import struct
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libmw_oracle.so")
TEX_DIR = os.path.join(os.path.dirname(HERE), "gym_miniworld_amd", "textures")

TASKS = {"Hallway": 0, "OneRoom": 1, "FourRooms": 2, "Maze": 3, "TMaze": 4, "TMazeTwoBox": 5,
         "SimToRealGoTo": 6, "SimToRealPush": 7, "PutNext": 8, "YMaze": 9,
         "PickupObjs": 10, "RoomObjs": 11, "CollectHealth": 12, "ThreeRooms": 13, "Sign": 14, "Sidewalk": 15, "WallGap": 16}
MAX_BOXES = 20
MESH_GEOMS = ["ball", "key", "medkit", "duckie", "building", "cone"]   # MWO_MESH_*
COLOR_NAMES = ["blue", "green", "grey", "purple", "red", "yellow"]
MESH_DIR = os.path.join(os.path.dirname(HERE), "gym_miniworld_amd", "meshes")
# (geometry, height) pairs the tasks build (entity.py:410-434 Key 0.35 / Ball size, sign.py:9-20 BigKey 0.6, the MeshEnt calls of the task files)
MESH_HEIGHTS = [("ball", 0.9), ("ball", 0.6), ("key", 0.35), ("key", 0.6), ("medkit", 0.40), ("duckie", 0.25), ("building", 30), ("cone", 0.75)]
# texture id table (family -> files), reference opengl.py:40-69 picks <name>_<i>.png
TEX_FILES = ["floor_tiles_bw_1", "concrete_1", "concrete_2", "concrete_3", "concrete_4",
             "concrete_tiles_1", "brick_wall_1",
             # the sim-to-real tasks' choices (envs/simtorealgoto.py:52-66)
             "cardboard_1", "cardboard_2", "cardboard_3", "cardboard_4", "wood_1", "wood_2", "wood_planks_1",
             "drywall_1", "stucco_1", "ceiling_tiles_1",
             # the entity tasks: room textures, the ImageFrame's picture, the images of the textured meshes (paths under meshes/), and
             # nine variants of each character of the Sign task's words (textures/chars/ch_0x<ord>_<i>.png, entity.py:268-278)
             "asphalt_1", "slime_1", "cinder_blocks_1", "logo_mila_1",
             "../meshes/medkit", "../meshes/duckie", "../meshes/building", "../meshes/cone"] + \
            ["chars/ch_0x%d_%d" % (ord(c), v) for c in "BLUERDGN" for v in range(1, 10)]
MAX_PORTALS = 2
NPARAM = 13


def build(force=False):
    src = os.path.join(HERE, "mw_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "libmw_oracle.so"], stdout=subprocess.DEVNULL)
    return LIB_PATH


class MwoState(ctypes.Structure):
    _fields_ = [
        ("agent_pos", ctypes.c_double * 3), ("agent_dir", ctypes.c_double),
        ("box_pos", ctypes.c_double * 3), ("box_dir", ctypes.c_double), ("box_color", ctypes.c_double * 3),
        ("cam_height", ctypes.c_double), ("cam_fwd_disp", ctypes.c_double), ("cam_pitch", ctypes.c_double),
        ("cam_fov_y", ctypes.c_double),
        ("sky_color", ctypes.c_double * 3), ("light_pos", ctypes.c_double * 3),
        ("light_color", ctypes.c_double * 3), ("light_ambient", ctypes.c_double * 3),
        ("cam_pos", ctypes.c_double * 3), ("cam_dir", ctypes.c_double * 3),
        ("step_count", ctypes.c_int), ("max_episode_steps", ctypes.c_int), ("n_rooms", ctypes.c_int),
        ("n_segs", ctypes.c_int), ("n_quads", ctypes.c_int), ("rng_pos", ctypes.c_int),
        ("rng_key0", ctypes.c_uint32), ("rng_key1", ctypes.c_uint32), ("rng_key623", ctypes.c_uint32),
        ("rng_keysum", ctypes.c_uint32),
        ("n_boxes", ctypes.c_int), ("goal_idx", ctypes.c_int),
        ("box2_pos", ctypes.c_double * 3), ("box2_dir", ctypes.c_double), ("box2_color", ctypes.c_double * 3),
        ("episode_count", ctypes.c_longlong), ("task_step_count", ctypes.c_longlong),
        ("feature", ctypes.c_double * 2),
        ("box_size", ctypes.c_double), ("box2_size", ctypes.c_double), ("agent_radius", ctypes.c_double),
        ("goal_dist", ctypes.c_double),
        ("boxes_pos", (ctypes.c_double * 3) * MAX_BOXES), ("boxes_dir", ctypes.c_double * MAX_BOXES),
        ("boxes_color", (ctypes.c_double * 3) * MAX_BOXES), ("boxes_size", ctypes.c_double * MAX_BOXES), ("carrying", ctypes.c_int),
        ("ents_kind", ctypes.c_int * MAX_BOXES), ("ents_mesh", ctypes.c_int * MAX_BOXES), ("ents_alive", ctypes.c_int * MAX_BOXES),
        ("ents_static", ctypes.c_int * MAX_BOXES), ("ents_rad_f32", ctypes.c_int * MAX_BOXES),
        ("ents_radius", ctypes.c_double * MAX_BOXES), ("ents_height", ctypes.c_double * MAX_BOXES), ("ents_scale", ctypes.c_double * MAX_BOXES),
        ("order", ctypes.c_int * (MAX_BOXES + 1)), ("n_order", ctypes.c_int), ("health", ctypes.c_double), ("num_picked", ctypes.c_int),
        ("ents_tex", (ctypes.c_int * 8) * MAX_BOXES),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(LIB_PATH)
        vp, dp, ip = ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
        L.mwo_create.restype = vp
        L.mwo_create.argtypes = [ctypes.c_int, dp, ctypes.c_int, ctypes.c_int, dp]
        L.mwo_destroy.argtypes = [vp]
        L.mwo_seed_key.argtypes = [vp, ctypes.POINTER(ctypes.c_uint32), ctypes.c_int]
        L.mwo_reset.argtypes = [vp]
        L.mwo_step.argtypes = [vp, ctypes.c_int, dp, ip]
        L.mwo_get_state.argtypes = [vp, ctypes.POINTER(MwoState)]
        L.mwo_set_agent.argtypes = [vp, ctypes.c_double, ctypes.c_double, ctypes.c_double]
        L.mwo_set_step_count.argtypes = [vp, ctypes.c_int]
        L.mwo_set_box.argtypes = [vp, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double]
        L.mwo_set_box_y.argtypes = [vp, ctypes.c_int, ctypes.c_double]
        L.mwo_set_carrying.argtypes = [vp, ctypes.c_int]
        L.mwo_render_step_frame.argtypes = [vp, ctypes.c_int]
        L.mwo_intersect_ent.argtypes = [vp, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double]
        L.mwo_set_counters.argtypes = [vp, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_int]
        L.mwo_get_geometry.argtypes = [vp] + [vp] * 13
        L.mwo_intersect.argtypes = [vp, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double]
        L.mwo_render.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp, vp]
        L.mwo_render_top.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp]
        L.mwo_visible_ents.argtypes = [vp, ctypes.c_int, ctypes.c_int]
        L.mwo_visible_ents.restype = ctypes.c_uint32
        L.mwo_set_texture.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
        L.mwo_set_mesh.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp, vp, ctypes.c_int, vp, vp]
        L.mwo_set_mesh_dims.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int]
        L.mwo_intersect_circle_segs.argtypes = [dp, ctypes.c_double, dp, ctypes.c_int]
        L.mwo_gen_rot_matrix.argtypes = [dp, ctypes.c_double, dp]
        L.mwo_bench_loop.restype = ctypes.c_double
        L.mwo_bench_loop.argtypes = [vp, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int,
                                     ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        _lib = L
    return _lib


# ----------------------------------------------------------------------------- seeding
def seed_to_mt_key(seed):
    """gym<=0.21 gym/utils/seeding.py np_random -> hash_seed -> _int_list_from_bigint, as recalled
    (un-vendored dependency, reference random.py:10).  PARITY UNPINNED: gym is absent here."""
    seed = int(seed) % 2 ** 64
    h = hashlib.sha512(str(seed).encode("utf8")).digest()[:8]
    lo, hi = struct.unpack("<2I", h)
    big = lo + (hi << 32)
    if big == 0:
        return [0]
    out = []
    while big > 0:
        big, mod = divmod(big, 2 ** 32)
        out.append(mod)
    return out


# ---------------------------------------------------------------------------- textures
def build_mip_chain(img_rgb):
    """img_rgb: HxWx3 uint8, row 0 = TOP of the image.  Returns list of RGBA8 levels with row 0 =
    BOTTOM (pyglet/GL upload order, opengl.py:85-96).  Levels: dims max(1, n//2), each texel the
    equal-weight box mean (round half up) of the source texels it covers (2x2 for even sizes)."""
    h, w, _ = img_rgb.shape
    lvl = np.concatenate([np.flipud(img_rgb), np.full((h, w, 1), 255, np.uint8)], axis=2).astype(np.uint8)
    levels = [lvl]
    while lvl.shape[0] > 1 or lvl.shape[1] > 1:
        sh, sw = lvl.shape[:2]
        dh, dw = max(1, sh // 2), max(1, sw // 2)
        out = np.zeros((dh, dw, 4), np.uint8)
        src = lvl.astype(np.int64)
        for j in range(dh):
            j0, j1 = (j * sh) // dh, -((-(j + 1) * sh) // dh)
            rows = src[j0:j1]
            if sw == 2 * dw:
                blk = rows[:, 0::2] + rows[:, 1::2]
                cnt = (j1 - j0) * 2
                out[j] = (blk.sum(axis=0) + cnt // 2) // cnt
            else:
                for i in range(dw):
                    i0, i1 = (i * sw) // dw, -((-(i + 1) * sw) // dw)
                    cnt = (j1 - j0) * (i1 - i0)
                    out[j, i] = (rows[:, i0:i1].sum(axis=(0, 1)) + cnt // 2) // cnt
        lvl = out
        levels.append(lvl)
    return levels


_tex_loaded = 0
_tex_cache = {}
N_TEX_BASE = 17   # the slots of the box-only tasks; the entity tasks load every slot


def load_textures(count=N_TEX_BASE):
    global _tex_loaded
    if _tex_loaded >= count:
        return _tex_cache
    from PIL import Image
    L = lib()
    for tid, name in enumerate(TEX_FILES[:count]):
        if tid < _tex_loaded:
            continue
        with Image.open(os.path.join(TEX_DIR, name + ".png")) as im:
            img = np.asarray(im.convert("RGB"))
        levels = build_mip_chain(img)
        flat = np.concatenate([lv.reshape(-1) for lv in levels]).astype(np.uint8)
        _tex_cache[tid] = (img.shape[1], img.shape[0], levels)
        rc = L.mwo_set_texture(tid, img.shape[1], img.shape[0], len(levels), flat.ctypes.data_as(ctypes.c_void_p))
        assert rc == 0
    _tex_loaded = count
    return _tex_cache


# ------------------------------------------------------------------------------ meshes
def load_obj(name):
    """OBJ / MTL -> the float32 arrays objmesh.py:33-216 hands to pyglet (the oracle's own small restatement: `v`, `vt`, `vn`,
    `usemtl`, `f` lines, faces stably sorted by material, re-centred with the reference's extents arithmetic - the smallest of
    the three per-slot maxima at objmesh.py:164).  Pinned by the digests of tests/golden/meshes.json."""
    geom = name.split("_")[0] if name.split("_")[0] in ("ball", "key") and "_" in name else name
    mats = {"": True}
    mtl = os.path.join(MESH_DIR, name + ".mtl")
    if os.path.exists(mtl):
        for ln in open(mtl):
            tk = ln.split()
            if tk and tk[0] == "newmtl":
                mats[tk[1]] = True
    V, T, N, F, cur = [], [], [], [], ""
    for ln in open(os.path.join(MESH_DIR, geom + ".obj")):
        tk = ln.split()
        if not tk or tk[0].startswith("#"):
            continue
        if tk[0] == "v":
            V.append([float(x) for x in tk[1:]])
        elif tk[0] == "vt":
            T.append([float(x) for x in tk[1:3]])
        elif tk[0] == "vn":
            N.append([float(x) for x in tk[1:]])
        elif tk[0] == "usemtl":
            cur = tk[1] if tk[1] in mats else ""
        elif tk[0] == "f":
            F.append((cur, [[int(i) for i in t.split("/") if i] for t in tk[1:]]))
    F = [f for _, f in sorted(enumerate(F), key=lambda kf: (kf[1][0], kf[0]))]
    n = len(F)
    verts, norms, texcs = np.zeros((n, 3, 3), np.float32), np.zeros((n, 3, 3), np.float32), np.zeros((n, 3, 2), np.float32)
    for i, (_, face) in enumerate(F):
        for k, idx in enumerate(face):
            verts[i, k] = V[idx[0] - 1]
            norms[i, k] = N[idx[-1] - 1]
            if len(idx) == 3:
                texcs[i, k] = T[idx[1] - 1]
    lo = verts.min(axis=0).min(axis=0)
    hi = verts.max(axis=0).min(axis=0)
    mid = (lo + hi) / 2
    verts[:, :, 1] -= lo[1]
    verts[:, :, 0] -= mid[0]
    verts[:, :, 2] -= mid[2]
    return verts, norms, texcs, verts.min(axis=0).min(axis=0), verts.max(axis=0).max(axis=0)


def mesh_dims(max_coords, height):
    """MeshEnt.__init__, entity.py:118-127, the reference's expressions with whatever scalar types the installed NumPy gives"""
    import math
    sx, sy, sz = max_coords
    scale = height / sy
    radius = math.sqrt(sx * sx + sz * sz) * scale
    return float(scale), float(radius), isinstance(radius, np.float32)


_mesh_loaded = False
_mesh_cache = {}


def load_meshes():
    global _mesh_loaded
    if _mesh_loaded:
        return _mesh_cache
    L = lib()
    for gi, g in enumerate(MESH_GEOMS):
        verts, norms, texcs, lo, hi = load_obj(g)
        tex = TEX_FILES.index("../meshes/" + g) if os.path.exists(os.path.join(MESH_DIR, g + ".png")) else -1
        _mesh_cache[g] = (verts, norms, texcs, lo, hi, tex)
        fp = lambda a: np.ascontiguousarray(a, np.float32).ctypes.data_as(ctypes.c_void_p)   # noqa: E731
        keep = [np.ascontiguousarray(a, np.float32) for a in (verts, norms, texcs, lo, hi)]
        assert L.mwo_set_mesh(gi, verts.shape[0], *[k.ctypes.data_as(ctypes.c_void_p) for k in keep[:3]], tex,
                              keep[3].ctypes.data_as(ctypes.c_void_p), keep[4].ctypes.data_as(ctypes.c_void_p)) == 0
    for g, h in MESH_HEIGHTS:
        sc, rad, f32 = mesh_dims(_mesh_cache[g][4], h)
        assert L.mwo_set_mesh_dims(MESH_GEOMS.index(g), float(h), sc, rad, int(f32)) == 0
    _mesh_loaded = True
    return _mesh_cache


# --------------------------------------------------------------------------------- env
class OracleEnv:
    """One environment of the CPU restatement; mirrors the slice of the Gym API the tests use."""

    def __init__(self, task="OneRoom", seed=None, domain_rand=False, max_episode_steps=0, task_args=None,
                 params=None, obs_width=80, obs_height=60, textures=True):
        L = lib()
        self.L = L
        self.task = task
        ta = None
        if task_args is not None:
            arr = (ctypes.c_double * 4)(*(list(task_args) + [0, 0, 0, 0])[:4])
            ta = arr
        pa = None
        if params is not None:
            p = np.ascontiguousarray(params, dtype=np.float64).reshape(NPARAM * 9)
            pa = p.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
            self._params_keep = p
        self.h = L.mwo_create(TASKS[task], ta, int(max_episode_steps), int(bool(domain_rand)), pa)
        self.W, self.H = obs_width, obs_height
        if TASKS[task] >= 10:
            load_meshes()   # before mwo_create's first reset: MeshEnt dimensions
        if textures:
            load_textures(len(TEX_FILES) if TASKS[task] >= 10 else N_TEX_BASE)
        if seed is not None:
            self.seed(seed)

    def __del__(self):
        try:
            self.L.mwo_destroy(self.h)
        except Exception:
            pass

    def seed(self, seed):
        key = seed_to_mt_key(seed)
        arr = (ctypes.c_uint32 * len(key))(*key)
        self.L.mwo_seed_key(self.h, arr, len(key))

    def reset(self, render=True):
        self.L.mwo_reset(self.h)
        return self.render_obs() if render else None

    def step(self, action, render=False):
        r = ctypes.c_double()
        d = ctypes.c_int()
        self.L.mwo_step(self.h, int(action), ctypes.byref(r), ctypes.byref(d))
        obs = self.render_obs() if render else None
        return obs, r.value, bool(d.value), {}

    def state(self):
        s = MwoState()
        self.L.mwo_get_state(self.h, ctypes.byref(s))
        return s

    def set_agent(self, x, z, d):
        self.L.mwo_set_agent(self.h, x, z, d)

    def set_box(self, b, x, z, d):
        self.L.mwo_set_box(self.h, int(b), x, z, d)

    def set_box_y(self, b, y):
        self.L.mwo_set_box_y(self.h, int(b), float(y))

    def set_carrying(self, b):
        self.L.mwo_set_carrying(self.h, int(b))

    def intersect_ent(self, ent_index, x, z, radius):
        return self.L.mwo_intersect_ent(self.h, int(ent_index), x, z, radius)

    def set_step_count(self, n):
        self.L.mwo_set_step_count(self.h, n)

    def set_counters(self, episode_count, task_step_count, goal_idx):
        self.L.mwo_set_counters(self.h, int(episode_count), int(task_step_count), int(goal_idx))

    def intersect_agent(self, x, z, radius=0.4):
        return self.L.mwo_intersect(self.h, 1, x, z, radius)

    def render_obs(self, depth=False, step_frame=False):
        """step_frame: the entities as the last step's own frame saw them (before PickupObjs removed / CollectHealth respawned one)"""
        self.L.mwo_render_step_frame(self.h, int(bool(step_frame)))
        rgb = np.zeros((self.H, self.W, 3), np.uint8)
        dep = np.zeros((self.H, self.W), np.float32) if depth else None
        self.L.mwo_render(self.h, self.W, self.H, rgb.ctypes.data_as(ctypes.c_void_p),
                          dep.ctypes.data_as(ctypes.c_void_p) if depth else None)
        return (rgb, dep) if depth else rgb

    def visible_ents(self):
        """get_visible_ents() (miniworld.py:1222-1315) as a bit mask over the boxes in entity-list order"""
        return int(self.L.mwo_visible_ents(self.h, self.W, self.H))

    def render_top(self, width=None, height=None):
        """render_top_view(frame_buffer) (miniworld.py:1087-1158) at width x height (default: the observation size)"""
        W, H = int(width or self.W), int(height or self.H)
        rgb = np.zeros((H, W, 3), np.uint8)
        self.L.mwo_render_top(self.h, W, H, rgb.ctypes.data_as(ctypes.c_void_p))
        return rgb

    def geometry(self):
        s = self.state()
        R, S, Q = s.n_rooms, s.n_segs, s.n_quads
        g = {
            "outline": np.zeros((R, 4, 2)), "wall_height": np.zeros(R),
            "portals": np.zeros((R, 4, MAX_PORTALS, 4)), "portal_count": np.zeros((R, 4), np.int32),
            "wall_segs": np.zeros((S, 4)), "room_probs": np.zeros(R),
            "wall_verts": np.zeros((Q * 4, 3)), "wall_norms": np.zeros((Q * 4, 3)),
            "wall_texcs": np.zeros((Q * 4, 2), np.float32), "quad_offsets": np.zeros(R + 1, np.int32),
            "floor_texcs": np.zeros((R, 4, 2)), "ceil_texcs": np.zeros((R, 4, 2)),
            "tex_ids": np.zeros((R, 3), np.int32),
        }
        order = ["outline", "wall_height", "portals", "portal_count", "wall_segs", "room_probs", "wall_verts",
                 "wall_norms", "wall_texcs", "quad_offsets", "floor_texcs", "ceil_texcs", "tex_ids"]
        self.L.mwo_get_geometry(self.h, *[g[k].ctypes.data_as(ctypes.c_void_p) for k in order])
        return g

    def bench_loop(self, n_steps, action_seed=0, env_index=0, want_depth=False, constant_action=-1, n_actions=3):
        return self.L.mwo_bench_loop(self.h, n_steps, action_seed, env_index, self.W, self.H,
                                     int(want_depth), constant_action, int(n_actions))


def intersect_circle_segs(pt, radius, segs):
    L = lib()
    pt = np.ascontiguousarray(pt, np.float64)
    segs = np.ascontiguousarray(segs, np.float64)
    dp = ctypes.POINTER(ctypes.c_double)
    return bool(L.mwo_intersect_circle_segs(pt.ctypes.data_as(dp), float(radius), segs.ctypes.data_as(dp),
                                            segs.shape[0]))


def gen_rot_matrix(axis, angle):
    L = lib()
    axis = np.ascontiguousarray(axis, np.float64)
    out = np.zeros(9)
    dp = ctypes.POINTER(ctypes.c_double)
    L.mwo_gen_rot_matrix(axis.ctypes.data_as(dp), float(angle), out.ctypes.data_as(dp))
    return out.reshape(3, 3)


def action_stream(action_seed, step, env_index):
    """Counter-based action in {0,1,2} shared by bench.py's GPU and CPU legs (splitmix64)."""
    M = (1 << 64) - 1

    def sm(x):
        x = (x + 0x9E3779B97F4A7C15) & M
        x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M
        x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M
        return x ^ (x >> 31)
    return (sm(action_seed ^ sm((step * 0x100000001B3 + env_index) & M)) >> 33) % 3
