"""ctypes binding of the C ABI in include/miniworld_batch.h (libmwbatch.so, built in-tree by
`python -m gym_miniworld_amd.build` / __graft_entry__.build()).

There is deliberately no fallback: if the HIP library is missing or no GPU is present, loading or
mwb_create fails loudly.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MWB_LIB") or os.path.join(HERE, "libmwbatch.so")   # MWB_LIB: kernel A/B experiments only

NPARAM = 13
STACK_SLIDING = 16   # MWB_STACK_SLIDING
STACK_FUSED = 32     # MWB_STACK_FUSED
STACK_SLACK_FRAMES = 8   # MWB_STACK_SLACK_FRAMES
ROOM_WORDS = 24
POLY_ROOM_WORDS = 52   # MWB_TASK_YMAZE (include/miniworld_batch.h)
ABI_VERSION = 5
MT_WORDS = 625

TASK_IDS = {"Hallway": 0, "OneRoom": 1, "FourRooms": 2, "Maze": 3, "TMaze": 4, "TMazeTwoBox": 5,
            "SimToRealGoTo": 6, "SimToRealPush": 7, "PutNext": 8, "YMaze": 9,
            "PickupObjs": 10, "RoomObjs": 11, "CollectHealth": 12, "ThreeRooms": 13, "Sign": 14, "Sidewalk": 15, "WallGap": 16}
ENT_TASK_FIRST = 10      # MWB_TASK_PICKUPOBJS: the tasks with a general entity list
MAX_ENTS = 20            # MWB_MAX_ENTS
MESH_GEOMS = ["ball", "key", "medkit", "duckie", "building", "cone"]   # MWB_MESH_*
TEX_MESH0, TEX_CHAR0 = 21, 25
LAYOUT_HWC, LAYOUT_CWH = 0, 1


class MwbConfig(ctypes.Structure):
    _fields_ = [
        ("abi_version", ctypes.c_int32), ("task", ctypes.c_int32), ("num_envs", ctypes.c_int32),
        ("obs_width", ctypes.c_int32), ("obs_height", ctypes.c_int32), ("want_depth", ctypes.c_int32),
        ("layout", ctypes.c_int32), ("domain_rand", ctypes.c_int32), ("max_episode_steps", ctypes.c_int32),
        ("device", ctypes.c_int32), ("task_args", ctypes.c_double * 4),
        ("use_default_params", ctypes.c_int32), ("no_auto_reset", ctypes.c_int32),
        ("params", (ctypes.c_double * 9) * NPARAM),
    ]


class MwbOutputs(ctypes.Structure):
    _fields_ = [
        ("obs", ctypes.c_void_p), ("depth", ctypes.c_void_p), ("reward", ctypes.c_void_p),
        ("reward64", ctypes.c_void_p), ("done", ctypes.c_void_p), ("ep_steps", ctypes.c_void_p),
        ("obs_bytes", ctypes.c_size_t), ("depth_bytes", ctypes.c_size_t),
        ("stack", ctypes.c_void_p), ("stack_bytes", ctypes.c_size_t),
        ("feature", ctypes.c_void_p), ("goal_pos", ctypes.c_void_p),
        ("pack", ctypes.c_void_p), ("pack_bytes", ctypes.c_size_t),
    ]


class MwbState(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in (
        "agent_pos", "agent_dir", "box_pos", "box_dir", "box_color", "box_size", "cam", "sky_color", "light_pos",
        "light_color", "light_ambient", "step_count", "rng_pos", "rng_keysum", "n_rooms", "n_segs",
        "goal_idx", "episode_count", "task_step_count", "goal_dist", "rng_state", "carrying",
        "ent_meta", "ent_radius", "ent_height", "ent_scale", "ent_order", "task_f", "task_i", "text_tex")]


EXPORTS = [
    "mwb_create", "mwb_destroy", "mwb_last_error", "mwb_abi_version", "mwb_set_texture", "mwb_seed", "mwb_reset",
    "mwb_step", "mwb_render", "mwb_get_outputs", "mwb_get_state", "mwb_set_agent", "mwb_intersect",
    "mwb_get_geometry", "mwb_timing_enable", "mwb_timing_read", "mwb_stack_enable", "mwb_stack_update", "mwb_stack_window", "mwb_check", "mwb_seed_key",
    "mwb_set_task_state", "mwb_set_domain_rand", "mwb_num_textures", "mwb_debug_wg_times", "mwb_set_state", "mwb_num_boxes", "mwb_room_words", "mwb_step_i64", "mwb_render_top_view", "mwb_visible_ents",
    "mwb_set_mesh", "mwb_set_mesh_dims", "mwb_render_view", "mwb_debug_counters",
]

_lib = None


class MwbError(RuntimeError):
    pass


def load():
    """Load libmwbatch.so; raises MwbError if it has not been built (no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MwbError("HIP extension %s is missing - run `python -m gym_miniworld_amd.build` "
                       "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    # PyTorch-ROCm wheels bundle their own libamdhip64; import torch first so that this library binds to
    # the SAME HIP runtime (two runtimes in one process do not see each other's devices or streams)
    import torch  # noqa: F401
    L = ctypes.CDLL(LIB_PATH)
    vp, i32 = ctypes.c_void_p, ctypes.c_int
    L.mwb_last_error.restype = ctypes.c_char_p
    L.mwb_create.argtypes = [ctypes.POINTER(MwbConfig), ctypes.POINTER(vp)]
    L.mwb_destroy.argtypes = [vp]
    L.mwb_set_texture.argtypes = [vp, i32, i32, i32, vp]
    L.mwb_set_mesh.argtypes = [vp, i32, i32, vp, vp, vp, i32, vp, vp, i32, i32, vp, vp]
    L.mwb_set_mesh_dims.argtypes = [vp, i32, ctypes.c_double, ctypes.c_double, ctypes.c_double, i32]
    L.mwb_seed.argtypes = [vp, vp]
    L.mwb_reset.argtypes = [vp, vp, vp]
    L.mwb_step.argtypes = [vp, vp, vp, vp]
    L.mwb_step_i64.argtypes = [vp, vp, vp, vp]
    L.mwb_render.argtypes = [vp, vp]
    L.mwb_render_top_view.argtypes = [vp, vp, i32, i32, vp]
    L.mwb_visible_ents.argtypes = [vp, vp, vp]
    L.mwb_render_view.argtypes = [vp, vp, vp, i32, i32, vp]
    L.mwb_get_outputs.argtypes = [vp, ctypes.POINTER(MwbOutputs)]
    L.mwb_get_state.argtypes = [vp, i32, i32, ctypes.POINTER(MwbState)]
    L.mwb_set_state.argtypes = [vp, i32, i32, ctypes.POINTER(MwbState)]
    L.mwb_num_boxes.argtypes = [vp]
    L.mwb_room_words.argtypes = [vp]
    L.mwb_set_agent.argtypes = [vp, i32, i32, vp, vp, vp]
    L.mwb_set_task_state.argtypes = [vp, i32, i32, vp, vp, vp]
    L.mwb_set_domain_rand.argtypes = [vp, i32]
    L.mwb_num_textures.argtypes = [vp]
    L.mwb_debug_wg_times.argtypes = [vp, vp, i32]
    L.mwb_debug_counters.argtypes = [vp, vp, i32]
    L.mwb_intersect.argtypes = [vp, i32, i32, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.POINTER(i32)]
    L.mwb_get_geometry.argtypes = [vp, i32, vp, i32, vp, i32, ctypes.POINTER(i32), ctypes.POINTER(i32)]
    L.mwb_timing_enable.argtypes = [vp, i32]
    L.mwb_stack_enable.argtypes = [vp, i32, i32]
    L.mwb_check.argtypes = [vp]
    L.mwb_seed_key.argtypes = [ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint32)]
    L.mwb_stack_update.argtypes = [vp, i32, vp]
    L.mwb_stack_window.argtypes = [vp, ctypes.POINTER(i32), ctypes.POINTER(i32)]
    L.mwb_timing_read.argtypes = [vp] + [ctypes.POINTER(ctypes.c_double)] * 4 + [ctypes.POINTER(i32)]
    if L.mwb_abi_version() != ABI_VERSION:
        raise MwbError("libmwbatch.so ABI version mismatch")
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise MwbError("libmwbatch error %d: %s" % (rc, load().mwb_last_error().decode("utf8", "replace")))
