"""BatchedMiniWorld: N MiniWorld environments stepped and rendered on one MI355X.

Thin host object over the C ABI (include/miniworld_batch.h).  PyTorch is used only as plumbing:
device tensors for actions / zero-copy views of the library's output buffers, and the current HIP
stream.  All simulation and rendering runs in the HIP kernels of csrc/.
"""
import ctypes
import os

import numpy as np

from . import _lib
from .params import DEFAULT_PARAMS, sim_to_real_params

TEX_FILES = ["floor_tiles_bw_1", "concrete_1", "concrete_2", "concrete_3", "concrete_4", "concrete_tiles_1",
             "brick_wall_1",
             # the sim-to-real rinks' choices (envs/simtorealgoto.py:52-66); uploaded only for those tasks
             "cardboard_1", "cardboard_2", "cardboard_3", "cardboard_4", "wood_1", "wood_2", "wood_planks_1",
             "drywall_1", "stucco_1", "ceiling_tiles_1",
             # the tasks with mesh entities / frames: room textures, the ImageFrame's picture, the images of the textured meshes, and
             # for Sign nine variants of each character of its words (Texture.get probes <name>_1 .. _9, opengl.py:50-58)
             "asphalt_1", "slime_1", "cinder_blocks_1", "logo_mila_1",
             "../meshes/medkit", "../meshes/duckie", "../meshes/building", "../meshes/cone"] + \
            ["chars/ch_0x%d_%d" % (ord(c), v) for c in "BLUERDGN" for v in range(1, 10)]
TEX_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "textures")
# (mesh geometry, height) pairs each entity task builds: Ball(size) / Key 0.35 / BigKey 0.6 (entity.py:410-434, sign.py:9-20), MeshEnt(...)
TASK_MESHES = {"PickupObjs": [("ball", 0.9), ("key", 0.35)], "RoomObjs": [("ball", 0.9), ("key", 0.35)], "CollectHealth": [("medkit", 0.40)],
               "ThreeRooms": [("duckie", 0.25), ("key", 0.35), ("ball", 0.6)], "Sign": [("key", 0.6)],
               "Sidewalk": [("building", 30), ("cone", 0.75)], "WallGap": [("building", 30)]}


def _sign_params():   # Sign.__init__, sign.py:62-64
    p = DEFAULT_PARAMS.no_random()
    p.set("forward_step", 0.7)
    p.set("turn_step", 45)
    return p

# registered ids of the reference (envs/__init__.py:43-49) that this package covers:
# id -> (task, task_args, max_episode_steps or 0 for the class default, params override)
def _fast_params(forward_step=0.7, turn_step=45):   # envs/oneroom.py:52-66, envs/maze.py:123-141
    p = DEFAULT_PARAMS.no_random()
    p.set("forward_step", forward_step)
    p.set("turn_step", turn_step)
    return p


ENV_SPECS = {
    "MiniWorld-Hallway-v0": ("Hallway", [12], 0, None, None),
    "MiniWorld-OneRoom-v0": ("OneRoom", [10], 0, None, None),
    "MiniWorld-OneRoomS6-v0": ("OneRoom", [6], 100, None, None),
    "MiniWorld-OneRoomS6Fast-v0": ("OneRoom", [6], 50, _fast_params, False),
    "MiniWorld-FourRooms-v0": ("FourRooms", [], 0, None, None),
    "MiniWorld-Maze-v0": ("Maze", [8, 8, 3], 0, None, None),
    "MiniWorld-MazeS2-v0": ("Maze", [2, 2, 3], 0, None, None),
    "MiniWorld-MazeS3-v0": ("Maze", [3, 3, 3], 0, None, None),
    "MiniWorld-MazeS3Fast-v0": ("Maze", [3, 3, 3], 300, _fast_params, False),
    # the T-maze family (envs/tmaze.py); task_args as documented at MWB_TASK_TMAZE / MWB_TASK_TMAZE_TWOBOX
    "MiniWorld-TMaze-v0": ("TMaze", [0, 0, 0, 0], 0, None, None),
    "MiniWorld-TMazeLeft-v0": ("TMaze", [1, 10, -6, 0], 0, None, None),      # tmaze.py:69-71
    "MiniWorld-TMazeRight-v0": ("TMaze", [1, 10, 6, 0], 0, None, None),      # tmaze.py:73-75
    "MiniWorld-TMazeDynamic-v0": ("TMaze", [1, 10, -6, 100], 0, None, None),  # tmaze.py:77-96
    "MiniWorld-TMazeTwoBoxDynamic-v0": ("TMazeTwoBox", [0, 0, 0, 100], 0, None, None),                  # tmaze.py:108-149
    "MiniWorld-TMazeTwoBoxDynamicFeatures100K-v0": ("TMazeTwoBox", [1, 0, 0, 100000], 0, None, None),  # tmaze.py:220-251
    "MiniWorld-TMazeTwoBoxDynamicFeatures1M-v0": ("TMazeTwoBox", [1, 0, 0, 1000000], 0, None, None),
    "MiniWorld-TMazeTwoBoxDynamicFeatures10M-v0": ("TMazeTwoBox", [1, 0, 0, 10000000], 0, None, None),
    "MiniWorld-TMazeTwoBoxDynamicFeaturesDebug-v0": ("TMazeTwoBox", [1, 0, 0, 9000000000000], 0, None, None),
    # the sim-to-real rinks: own parameter table, domain randomisation forced on (simtorealgoto.py:27-33)
    "MiniWorld-SimToRealGoTo-v0": ("SimToRealGoTo", [], 0, sim_to_real_params, True),
    "MiniWorld-SimToRealPush-v0": ("SimToRealPush", [], 0, lambda: sim_to_real_params(push=True), True),
    # several boxes + the carry actions, envs/putnext.py
    "MiniWorld-PutNext-v0": ("PutNext", [12], 0, None, None),
    # rooms that are general convex polygons, envs/ymaze.py; task_args {goal given, x, z}
    "MiniWorld-YMaze-v0": ("YMaze", [0, 0, 0], 0, None, None),
    "MiniWorld-YMazeLeft-v0": ("YMaze", [1, 3.9, -7.0], 0, None, None),    # ymaze.py:96-98
    "MiniWorld-YMazeRight-v0": ("YMaze", [1, 3.9, 7.0], 0, None, None),    # ymaze.py:100-102
    # a general entity list: mesh entities (Ball, Key, MeshEnt), frames, entities leaving / re-entering the list
    "MiniWorld-PickupObjs-v0": ("PickupObjs", [12, 5], 0, None, None),         # pickupobjs.py:13-22
    "MiniWorld-RoomObjs-v0": ("RoomObjs", [10], 0, None, None),                # roomobjs.py:14-21
    "MiniWorld-CollectHealth-v0": ("CollectHealth", [16], 0, None, None),      # collecthealth.py:19-26
    "MiniWorld-ThreeRooms-v0": ("ThreeRooms", [], 0, None, None),              # threerooms.py:12-20: two openings in one wall
    "MiniWorld-Sign-v0": ("Sign", [10, 0, 0], 0, _sign_params, False),         # sign.py:41-71: its own params, domain_rand forced off
    "MiniWorld-Sidewalk-v0": ("Sidewalk", [], 0, None, None),
    "MiniWorld-WallGap-v0": ("WallGap", [], 0, None, None),
}



_BVH_CACHE = {}


class _DevView:
    """Exposes a library-owned device buffer through __cuda_array_interface__ (zero-copy)."""

    def __init__(self, ptr, shape, typestr, owner):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}
        self._owner = owner   # keeps the handle alive while views exist


class BatchedMiniWorld:
    """N environments of one task on one GPU.

    reset()/step() mirror VecEnv semantics of the reference's SubprocVecEnv worker
    (vec_env/subproc_vec_env.py:5-33): env i is seeded `seed + first_env_index + i`
    (pytorch-a2c-ppo-acktr/envs.py:36), a finished env is reset inside step() and the returned
    observation is the first one of the new episode while reward/done are the terminal ones.
    """

    def __init__(self, env_id="MiniWorld-OneRoom-v0", num_envs=1, seed=None, domain_rand=False, obs_width=80,
                 obs_height=60, want_depth=False, layout="HWC", device=0, max_episode_steps=None, params=None,
                 task=None, task_args=None, first_env_index=0, auto_reset=True):
        import torch
        self.torch = torch
        self.L = _lib.load()
        if task is None:
            if env_id not in ENV_SPECS:
                raise KeyError("unknown or out-of-scope env id %r; supported: %s" % (env_id, sorted(ENV_SPECS)))
            task, spec_args, spec_steps, spec_params, spec_dr = ENV_SPECS[env_id]
            task_args = spec_args if task_args is None else task_args
            if max_episode_steps is None:
                max_episode_steps = spec_steps
            if params is None and spec_params is not None:
                params = spec_params()
            if spec_dr is not None:
                domain_rand = spec_dr
        self.env_id, self.task = env_id, task
        self.task_args = list(task_args or []) + [0, 0, 0, 0]
        self.num_envs = int(num_envs)
        self.W, self.H = int(obs_width), int(obs_height)
        self.want_depth = bool(want_depth)
        self.layout = layout
        self.domain_rand = bool(domain_rand)
        self.params = params if params is not None else DEFAULT_PARAMS
        self.first_env_index = int(first_env_index)
        if not torch.cuda.is_available():
            raise _lib.MwbError("BatchedMiniWorld needs a GPU (torch.cuda.is_available() is False); no CPU path exists")
        self.device_index = int(device)
        self.device = torch.device("cuda", self.device_index)

        cfg = _lib.MwbConfig()
        cfg.abi_version = _lib.ABI_VERSION
        cfg.task = _lib.TASK_IDS[task]
        cfg.num_envs = self.num_envs
        cfg.obs_width, cfg.obs_height = self.W, self.H
        cfg.want_depth = int(self.want_depth)
        cfg.layout = {"HWC": _lib.LAYOUT_HWC, "CWH": _lib.LAYOUT_CWH}[layout]
        cfg.domain_rand = int(self.domain_rand)
        cfg.max_episode_steps = int(max_episode_steps or 0)
        cfg.device = self.device_index
        ta = list(task_args or []) + [0, 0, 0, 0]
        for i in range(4):
            cfg.task_args[i] = float(ta[i])
        cfg.use_default_params = 0
        cfg.no_auto_reset = 0 if auto_reset else 1
        table = self.params.to_table()
        for i in range(_lib.NPARAM):
            for j in range(9):
                cfg.params[i][j] = float(table[i, j])
        h = ctypes.c_void_p()
        _lib.check(self.L.mwb_create(ctypes.byref(cfg), ctypes.byref(h)))
        self.h = h
        self._load_textures()
        if task in TASK_MESHES:
            self._load_meshes(task)
        out = _lib.MwbOutputs()
        _lib.check(self.L.mwb_get_outputs(self.h, ctypes.byref(out)))
        N, W, H = self.num_envs, self.W, self.H
        obs_shape = (N, H, W, 3) if layout == "HWC" else (N, 3, W, H)
        as_t = lambda ptr, shape, ts: torch.as_tensor(_DevView(ptr, shape, ts, self), device=self.device)  # noqa: E731
        self.obs = as_t(out.obs, obs_shape, "|u1")
        self.depth = as_t(out.depth, (N, H, W, 1), "<f4") if self.want_depth else None
        self.reward = as_t(out.reward, (N,), "<f4")
        self.reward64 = as_t(out.reward64, (N,), "<f8")
        self.done = as_t(out.done, (N,), "|u1")
        self.ep_steps = as_t(out.ep_steps, (N,), "<i4")
        self.feature = as_t(out.feature, (N, 2), "<f4")     # info['feature'] (tmaze.py:311-318); zeros elsewhere
        self.goal_pos = as_t(out.goal_pos, (N, 3), "<f8")   # info['goal_pos'] of the T-maze family
        # the six small outputs above are parts of one allocation: a host-side consumer copies `pack` once per step
        self.pack = as_t(out.pack, (out.pack_bytes,), "|u1")
        self.pack_offsets = {k: int(getattr(out, k)) - int(out.pack) for k in ("reward64", "goal_pos", "reward", "feature", "ep_steps", "done")}
        self.n_boxes = int(self.L.mwb_num_boxes(self.h))   # 1; 2 (TMazeTwoBox, SimToRealPush); 6 (PutNext)
        self.agent_radius = 0.11 if task.startswith("SimToReal") else 1.5 if task == "RoomObjs" else 0.4
        self.ent_task = _lib.TASK_IDS[task] >= _lib.ENT_TASK_FIRST
        # action_space: Discrete(move_forward + 1) in the navigation tasks (e.g. hallway.py:23), Discrete(move_back + 1)
        # in SimToRealPush (simtorealpush.py:37), the base class' Discrete(len(Actions)) = 8 where the task does not
        # narrow it (PutNext: miniworld.py:470)
        self.n_actions = {"SimToRealPush": 4, "PutNext": 8, "PickupObjs": 5, "RoomObjs": 8, "CollectHealth": 8, "Sign": 4}.get(task, 3)
        self.has_health = task == "CollectHealth"   # info['health'] arrives in feature[:, 0]
        self.has_features = task == "TMazeTwoBox" and ta[0] != 0
        self.has_goal_pos = task in ("TMaze", "TMazeTwoBox", "YMaze")   # tmaze.py:66,206, ymaze.py:92
        self.max_episode_steps = self._max_steps(task, ta, max_episode_steps)
        if seed is not None:
            self.seed(seed)

    @staticmethod
    def _max_steps(task, ta, mes):
        if mes:
            return int(mes)
        return {"Hallway": 250, "OneRoom": 180, "FourRooms": 250, "TMaze": 280, "TMazeTwoBox": 280,
                "SimToRealGoTo": 100, "SimToRealPush": 150, "PutNext": 250, "YMaze": 280, "PickupObjs": 400, "RoomObjs": 2 ** 31 - 1,
                "CollectHealth": 1000, "ThreeRooms": 400, "Sign": 20, "Sidewalk": 150, "WallGap": 300}.get(task) or int(ta[0] or 8) * int(ta[1] or 8) * 24

    def _load_textures(self):
        from PIL import Image
        for tid, name in enumerate(TEX_FILES[:self.L.mwb_num_textures(self.h)]):
            with Image.open(os.path.join(TEX_DIR, name + ".png")) as im:
                img = np.ascontiguousarray(np.asarray(im.convert("RGB"), dtype=np.uint8))
            _lib.check(self.L.mwb_set_texture(self.h, tid, img.shape[1], img.shape[0],
                                              img.ctypes.data_as(ctypes.c_void_p)))

    def _load_meshes(self, task):
        """ObjMesh.get for every mesh the task's entities use (objmesh.py:16-216) + the BVH the render kernel walks, and
        MeshEnt.__init__'s scale / radius (entity.py:118-127) evaluated here with the reference's expressions"""
        from . import meshes as M
        vp = ctypes.c_void_p
        done = set()
        for geom, height in TASK_MESHES[task]:
            name = geom + "_red" if geom in ("ball", "key") else geom   # the colour variants share their geometry
            m = M.get(name)
            if geom not in done:
                done.add(geom)
                octants = os.environ.get("MWB_BVH_OCTANTS", "1") != "0"   # 0: one threading for every ray direction (A/B timing)
                leaf = int(os.environ.get("MWB_BVH_LEAF", "8"))   # leaves of up to 8 triangles: 4 and 16 are 2-5 % slower, 2 and 1 10-35 %
                key = (name, octants, leaf)
                if key not in _BVH_CACHE:   # a second of NumPy per mesh: built once per process
                    _BVH_CACHE[key] = M.build_bvh(m.verts, leaf_size=leaf, octants=octants)
                nodes, perm = _BVH_CACHE[key]
                n_orders = 8 if octants else 1
                tex = TEX_FILES.index("../meshes/" + geom) if m.chunks[0][2] is not None else -1
                assert len(m.chunks) == 1, "one material per mesh"
                arrs = [np.ascontiguousarray(a, np.float32) for a in (m.verts, m.norms, m.texcs, m.min_coords, m.max_coords, nodes)]
                perm = np.ascontiguousarray(perm, np.int32)
                _lib.check(self.L.mwb_set_mesh(self.h, _lib.MESH_GEOMS.index(geom), m.n_tris, arrs[0].ctypes.data_as(vp), arrs[1].ctypes.data_as(vp),
                                               arrs[2].ctypes.data_as(vp), tex, arrs[3].ctypes.data_as(vp), arrs[4].ctypes.data_as(vp),
                                               nodes.shape[0] // n_orders, n_orders, arrs[5].ctypes.data_as(vp), perm.ctypes.data_as(vp)))
            scale, radius = M.mesh_ent_dims(name, height)
            _lib.check(self.L.mwb_set_mesh_dims(self.h, _lib.MESH_GEOMS.index(geom), float(height), float(scale), float(radius),
                                                int(isinstance(radius, np.float32))))

    # -------------------------------------------------------------------------------- lifecycle
    def close(self):
        if getattr(self, "h", None):
            self.torch.cuda.synchronize(self.device)
            self.L.mwb_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return ctypes.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    # ------------------------------------------------------------------------------- simulation
    def seed(self, seed):
        """int -> env i gets seed + first_env_index + i (envs.py:36); or an explicit array of N seeds."""
        if np.isscalar(seed):
            seeds = (int(seed) + self.first_env_index + np.arange(self.num_envs, dtype=np.uint64)).astype(np.uint64)
        else:
            seeds = np.asarray(seed, dtype=np.uint64)
            assert seeds.shape == (self.num_envs,)
        seeds = np.ascontiguousarray(seeds)
        _lib.check(self.L.mwb_seed(self.h, seeds.ctypes.data_as(ctypes.c_void_p)))
        return [seed]

    def reset(self, mask=None):
        m = None
        if mask is not None:
            mask = mask.to(device=self.device, dtype=self.torch.uint8).contiguous()
            m = ctypes.c_void_p(mask.data_ptr())
        _lib.check(self.L.mwb_reset(self.h, m, self._stream()))
        return self.obs

    def step(self, actions, skip_mask=None):
        """actions: int tensor [N] (any int dtype / device); returns views (obs, reward, done) of the
        library-owned buffers, valid until the next call."""
        torch = self.torch
        a = torch.as_tensor(actions)
        i64 = a.dtype == torch.int64 and a.device == self.device   # the policy's LongTensor: read in place (mwb_step_i64)
        a = (a if i64 else a.to(device=self.device, dtype=torch.int32)).reshape(-1).contiguous()
        assert a.numel() == self.num_envs
        m = None
        if skip_mask is not None:
            skip_mask = torch.as_tensor(skip_mask).to(device=self.device, dtype=torch.uint8).contiguous()
            m = ctypes.c_void_p(skip_mask.data_ptr())
        _lib.check((self.L.mwb_step_i64 if i64 else self.L.mwb_step)(self.h, ctypes.c_void_p(a.data_ptr()), m, self._stream()))
        self._keep = (a, skip_mask)   # keep inputs alive until the async kernels have consumed them
        return self.obs, self.reward, self.done

    def step_longtensor(self, actions):
        """step() for the one shape a trainer's loop produces every time - a contiguous int64 tensor [N] on this device - without
        the generic conversions (the host time before the first kernel launch is on the critical path of a VecEnv step)"""
        rc = self.L.mwb_step_i64(self.h, actions.data_ptr(), None, self._stream())
        if rc:
            _lib.check(rc)
        self._keep = (actions, None)

    def render(self):
        _lib.check(self.L.mwb_render(self.h, self._stream()))
        return self.obs

    def visible_ents(self):
        """MiniWorldEnv.get_visible_ents() for every env (miniworld.py:1222-1315): int32 [N] on the device, bit b = box b
        (entity-list order) passes its occlusion query"""
        out = self.torch.empty((self.num_envs,), dtype=self.torch.int32, device=self.device)
        _lib.check(self.L.mwb_visible_ents(self.h, out.data_ptr(), self._stream()))
        return out

    def render_top_view(self, width=None, height=None):
        """MiniWorldEnv.render_top_view(frame_buffer) for every env (miniworld.py:1087-1158): uint8 [N, height, width, 3] on
        the device, default the observation size (the reference's default frame buffer is obs_fb)."""
        W, H = int(width or self.W), int(height or self.H)
        out = self.torch.empty((self.num_envs, H, W, 3), dtype=self.torch.uint8, device=self.device)
        _lib.check(self.L.mwb_render_top_view(self.h, out.data_ptr(), W, H, self._stream()))
        return out

    def render_view(self, width, height, depth=False):
        """MiniWorldEnv.render_obs(frame_buffer) with another frame buffer than the observation's (miniworld.py:1160-1205), e.g.
        the 800 x 600 human view of render(mode='rgb_array'): uint8 [N, height, width, 3] (and float32 [N, height, width, 1]
        metres with depth=True) on the device, any size (rendered in tiles)"""
        W, H = int(width), int(height)
        out = self.torch.empty((self.num_envs, H, W, 3), dtype=self.torch.uint8, device=self.device)
        dep = self.torch.empty((self.num_envs, H, W, 1), dtype=self.torch.float32, device=self.device) if depth else None
        _lib.check(self.L.mwb_render_view(self.h, out.data_ptr(), dep.data_ptr() if depth else None, W, H, self._stream()))
        return (out, dep) if depth else out

    # ---------------------------------------------------------------------------- introspection
    def check(self):
        """Synchronous: raises if world generation ever flagged a failure (see mwb_check)."""
        _lib.check(self.L.mwb_check(self.h))

    _STATE_F64 = {"agent_pos": (3,), "agent_dir": (), "cam": (4,), "sky_color": (3,), "light_pos": (3,),
                  "light_color": (3,), "light_ambient": (3,), "goal_dist": ()}
    _STATE_BOX = {"box_pos": (3,), "box_dir": (), "box_color": (3,), "box_size": ()}   # [count][B] + shape
    _STATE_INT = {"step_count": np.int32, "rng_pos": np.int32, "rng_keysum": np.uint32, "n_rooms": np.int32,
                  "n_segs": np.int32, "goal_idx": np.int32, "episode_count": np.int64, "task_step_count": np.int64,
                  "carrying": np.int32}

    def get_state(self, first=0, count=None, rng_state=False):
        """Host snapshot of the env range (mwb_get_state).  Box fields come as "boxes_pos" [count, B, 3],
        "boxes_dir", "boxes_color", "boxes_size" and, for the one- and two-box tasks, also under the
        reference's attribute names: "box_*" = entity 0 (env.box / red_box), "box2_*" = entity 1."""
        self.check()
        count = self.num_envs - first if count is None else count
        B = self.n_boxes
        out = {k: np.zeros((count,) + s, np.float64) for k, s in self._STATE_F64.items()}
        boxes = {k: np.zeros((count, B) + s, np.float64) for k, s in self._STATE_BOX.items()}
        out.update({k: np.zeros(count, dt) for k, dt in self._STATE_INT.items()})
        if rng_state:
            out["rng_state"] = np.zeros((count, _lib.MT_WORDS), np.uint32)
        st = _lib.MwbState()
        for k, v in list(out.items()) + list(boxes.items()):
            setattr(st, k, v.ctypes.data_as(ctypes.c_void_p))
        ents = {}
        if self.ent_task:   # the general entity list: per slot kind / flags / dimensions, the list order, the task's counters
            ents = {"ent_meta": np.zeros((count, B), np.int32), "ent_radius": np.zeros((count, B)), "ent_height": np.zeros((count, B)),
                    "ent_scale": np.zeros((count, B)), "ent_order": np.zeros((count, B + 1), np.int32), "task_f": np.zeros(count),
                    "task_i": np.zeros(count, np.int32), "text_tex": np.zeros((count, 8), np.int32)}
            for k, v in ents.items():
                setattr(st, k, v.ctypes.data_as(ctypes.c_void_p))
        _lib.check(self.L.mwb_get_state(self.h, first, count, ctypes.byref(st)))
        if self.ent_task:
            m = ents["ent_meta"]
            ents.update({"ent_kind": m & 15, "ent_geom": (m >> 4) & 15, "ent_static": (m >> 8) & 1, "ent_alive": (m >> 9) & 1,
                         "ent_rad_f32": (m >> 10) & 1, "ent_color": ((m >> 12) & 15) - 1})
        out.update(ents)
        for k, v in boxes.items():
            out["boxes_" + k[4:]] = v
            out[k] = v[:, 0]
            out["box2_" + k[4:]] = v[:, 1] if B > 1 else np.zeros_like(v[:, 0])
        return out

    def set_state(self, first, **fields):
        """Overwrite simulator state of envs first .. first+count-1 (mwb_set_state).  Keys as returned by
        get_state(): agent_pos [count,3], agent_dir, boxes_pos [count,B,3], boxes_dir, boxes_color, boxes_size,
        cam, sky_color, light_*, step_count, goal_idx, episode_count, task_step_count, goal_dist, carrying,
        rng_state [count,625]."""
        st = _lib.MwbState()
        keep, count = [], None
        for k, v in fields.items():
            name = "box_" + k[6:] if k.startswith("boxes_") else k
            if name in self._STATE_BOX:
                shape, dt = (-1, self.n_boxes) + self._STATE_BOX[name], np.float64
            elif name in self._STATE_F64:
                shape, dt = (-1,) + self._STATE_F64[name], np.float64
            elif name == "rng_state":
                shape, dt = (-1, _lib.MT_WORDS), np.uint32
            elif name in ("step_count", "goal_idx", "episode_count", "task_step_count", "carrying"):
                shape, dt = (-1,), self._STATE_INT[name]
            elif name in ("ent_meta", "ent_order", "task_i"):
                shape, dt = {"ent_meta": (-1, self.n_boxes), "ent_order": (-1, self.n_boxes + 1), "task_i": (-1,)}[name], np.int32
            elif name == "task_f":
                shape, dt = (-1,), np.float64
            else:
                raise KeyError("set_state: %r is not a writable state field" % k)
            a = np.ascontiguousarray(np.asarray(v, dtype=dt).reshape(shape))
            if count is not None and a.shape[0] != count:
                raise ValueError("set_state: fields disagree on the number of envs")
            count = a.shape[0]
            keep.append(a)
            setattr(st, name, a.ctypes.data_as(ctypes.c_void_p))
        if count:
            _lib.check(self.L.mwb_set_state(self.h, int(first), int(count), ctypes.byref(st)))

    def set_agent(self, first, pos_xz=None, dir=None, step_count=None):
        arrs = [None if a is None else np.ascontiguousarray(a, dt) for a, dt in
                ((pos_xz, np.float64), (dir, np.float64), (step_count, np.int32))]
        count = next(len(a) if a.ndim else 1 for a in arrs if a is not None)
        ptrs = [None if a is None else a.ctypes.data_as(ctypes.c_void_p) for a in arrs]
        _lib.check(self.L.mwb_set_agent(self.h, first, count, *ptrs))

    def set_domain_rand(self, flag):
        """`env.domain_rand = flag` of the reference (run_tests.py:64-66); effective from the next reset / step."""
        _lib.check(self.L.mwb_set_domain_rand(self.h, int(bool(flag))))
        self.domain_rand = bool(flag)

    def set_task_state(self, first, episode_count=None, task_step_count=None, goal_idx=None):
        """Overwrite the goal-alternation state of the T-maze family (test hook, mwb_set_task_state)."""
        arrs = [None if a is None else np.ascontiguousarray(np.atleast_1d(a), dt) for a, dt in
                ((episode_count, np.int64), (task_step_count, np.int64), (goal_idx, np.int32))]
        count = next(len(a) for a in arrs if a is not None)
        ptrs = [None if a is None else a.ctypes.data_as(ctypes.c_void_p) for a in arrs]
        _lib.check(self.L.mwb_set_task_state(self.h, first, count, *ptrs))

    def intersect(self, env, x, z, radius=0.4, ent=None):
        """MiniWorldEnv.intersect(ent, pos, radius) of env `env` (miniworld.py:933-959).  ent = index of the
        querying entity in the entity list (boxes first, the agent last; default: the agent), -1 = nobody is
        skipped.  Returns 0 none, 1 wall, 2 + k = entity k."""
        r = ctypes.c_int()
        ent = self.n_boxes if ent is None else int(ent)
        _lib.check(self.L.mwb_intersect(self.h, env, ent, float(x), float(z), float(radius), ctypes.byref(r)))
        return r.value

    def get_geometry(self, env, max_rooms=512, max_segs=2048):
        """(rooms [n_rooms, mwb_room_words] f32, segs [n_segs, 4] f64) of one env as the kernels see them: the rectangle
        room table, or for YMaze the polygon one (layouts: include/miniworld_batch.h at mwb_get_geometry)."""
        rooms = np.zeros((max_rooms, int(self.L.mwb_room_words(self.h))), np.float32)
        segs = np.zeros((max_segs, 4), np.float64)
        nr, ns = ctypes.c_int(), ctypes.c_int()
        _lib.check(self.L.mwb_get_geometry(self.h, env, rooms.ctypes.data_as(ctypes.c_void_p), max_rooms,
                                           segs.ctypes.data_as(ctypes.c_void_p), max_segs, ctypes.byref(nr),
                                           ctypes.byref(ns)))
        return rooms[:nr.value], segs[:ns.value]

    # ------------------------------------------------------------------------------ frame stack
    def stack_enable(self, nstack=4, dtype="float32", sliding=True, fused=False):
        """Library-owned [N, nstack*3, W, H] stack kept by one fused HIP pass per step
        (VecPyTorchFrameStack + .float(), pytorch-a2c-ppo-acktr/envs.py:117-165). Needs layout='CWH'.
        sliding=True: the stack is a window that moves over a longer run of planes per env, so a step writes the new frame only
        (1/4 of the shifting stack's HBM traffic); stack_update() then returns a strided view [N, nstack*3, W, H] of it (same
        values; a different view object every step).  sliding=False: the shifting stack in one fixed contiguous tensor.
        fused=True (implies sliding): reset() / step() write each new frame into the window themselves, straight from the render
        kernel's LDS frame, and zero the history of the envs they regenerate - stack_update() then only returns the view."""
        torch = self.torch
        is_f = {"float32": 1, "uint8": 0}[dtype]
        sliding = sliding or fused
        _lib.check(self.L.mwb_stack_enable(self.h, int(nstack), is_f | (_lib.STACK_SLIDING if sliding else 0) | (_lib.STACK_FUSED if fused else 0)))
        out = _lib.MwbOutputs()
        _lib.check(self.L.mwb_get_outputs(self.h, ctypes.byref(out)))
        first, planes = ctypes.c_int32(), ctypes.c_int32()
        _lib.check(self.L.mwb_stack_window(self.h, ctypes.byref(first), ctypes.byref(planes)))
        self._stack_c, self._stack_sliding = nstack * 3, bool(sliding)
        shape = (self.num_envs, planes.value, self.W, self.H)
        self._stack_base = torch.as_tensor(_DevView(out.stack, shape, "<f4" if is_f else "|u1", self), device=self.device)
        self.stack = self._stack_base[:, first.value:first.value + self._stack_c]
        return self.stack

    def stack_update(self, after_reset=False):
        _lib.check(self.L.mwb_stack_update(self.h, int(bool(after_reset)), self._stream()))
        if self._stack_sliding:
            first = ctypes.c_int32()
            _lib.check(self.L.mwb_stack_window(self.h, ctypes.byref(first), None))
            self.stack = self._stack_base[:, first.value:first.value + self._stack_c]
        return self.stack

    def timing_enable(self, on=True):
        """on: False / True, or an int n > 1 to time every n-th pass only (the events cost ~20 us per pass)."""
        _lib.check(self.L.mwb_timing_enable(self.h, int(on)))

    def timing_read(self):
        v = [ctypes.c_double() for _ in range(4)]
        n = ctypes.c_int()
        _lib.check(self.L.mwb_timing_read(self.h, *[ctypes.byref(x) for x in v], ctypes.byref(n)))
        return {"step": v[0].value, "reset": v[1].value, "prep": v[2].value, "render": v[3].value, "n": n.value}
