"""Inert stand-ins for the two third-party packages the reference imports but this
container lacks (`gym`, `pyglet`).  BUILD-CONTAINER ONLY: used by gen_fixtures.py to run the
*unmodified* reference (read from /root/reference) so that its own state arithmetic
(geometry, placement, trajectories, rewards, dones, camera vectors) and the exact argument
stream it hands to OpenGL can be captured as golden vectors.  Nothing here restates
reference logic; every gl*/glu* entry point is a recorder that computes nothing, so the
observations the reference returns under these stand-ins are all-zero (there is no pixel
oracle).

The only arithmetic in this file is `np_random`, the seed -> MT19937 key mapping of the
un-vendored `gym` (<= 0.21) package as recalled (gym/utils/seeding.py: sha512 of the
decimal string, first 8 bytes, little-endian 32-bit limbs -> RandomState.seed(list)).
It cannot be checked against gym here ("parity unpinned" for that one mapping); the
product isolates the same mapping in one function.
"""
import ctypes
import hashlib
import re
import struct
import sys
import types

import numpy as np

GL_LOG = []          # (name, args) tuples, appended by every gl*/glu* call
TEX_PATHS = {}       # texture id handed out by the image stand-in -> path of the PNG it was loaded from
VLISTS = []          # every pyglet.graphics.vertex_list the reference created (mesh chunks), by id
GL_LOG_ENABLED = [False]


# ----------------------------------------------------------------------------- gym
def _bigint_from_bytes(b):
    sizeof_int = 4
    padding = sizeof_int - len(b) % sizeof_int
    b += b"\0" * padding
    n = len(b) // sizeof_int
    vals = struct.unpack("{}I".format(n), b)
    acc = 0
    for i, v in enumerate(vals):
        acc += 2 ** (sizeof_int * 8 * i) * v
    return acc


def _int_list_from_bigint(bigint):
    if bigint == 0:
        return [0]
    out = []
    while bigint > 0:
        bigint, mod = divmod(bigint, 2 ** 32)
        out.append(mod)
    return out


def seed_to_mt_key(seed):
    seed = int(seed) % 2 ** 64
    h = hashlib.sha512(str(seed).encode("utf8")).digest()
    return _int_list_from_bigint(_bigint_from_bytes(h[:8]))


def np_random(seed=None):
    if seed is None:
        seed = 0x5EED  # the reference constructor calls seed() once with None; value irrelevant
    rng = np.random.RandomState()
    rng.seed(seed_to_mt_key(seed))
    return rng, seed


class _Space:
    pass


class Discrete(_Space):
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.int64


class Box(_Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is None:
            shape = np.asarray(low).shape
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.low = np.full(self.shape, low, dtype=self.dtype) if np.isscalar(low) else np.asarray(low)
        self.high = np.full(self.shape, high, dtype=self.dtype) if np.isscalar(high) else np.asarray(high)


class Env:
    metadata = {}

    @property
    def unwrapped(self):
        return self


class Wrapper(Env):
    def __init__(self, env):
        self.env = env
        self.observation_space = getattr(env, "observation_space", None)
        self.action_space = getattr(env, "action_space", None)


class ObservationWrapper(Wrapper):
    pass


def _install_gym():
    gym = types.ModuleType("gym")
    gym.Env = Env
    gym.Wrapper = Wrapper
    gym.ObservationWrapper = ObservationWrapper
    core = types.ModuleType("gym.core")
    core.Env = Env
    spaces = types.ModuleType("gym.spaces")
    spaces.Discrete = Discrete
    spaces.Box = Box
    spaces.Dict = dict
    box_mod = types.ModuleType("gym.spaces.box")
    box_mod.Box = Box
    spaces.box = box_mod
    utils = types.ModuleType("gym.utils")
    seeding = types.ModuleType("gym.utils.seeding")
    seeding.np_random = np_random
    utils.seeding = seeding
    envs = types.ModuleType("gym.envs")
    registration = types.ModuleType("gym.envs.registration")
    registration.registry = {}

    def register(id, entry_point=None, **kw):
        registration.registry[id] = entry_point

    registration.register = register
    envs.registration = registration
    gym.core, gym.spaces, gym.utils, gym.envs = core, spaces, utils, envs
    for name, mod in [("gym", gym), ("gym.core", core), ("gym.spaces", spaces),
                      ("gym.spaces.box", box_mod), ("gym.utils", utils),
                      ("gym.utils.seeding", seeding), ("gym.envs", envs),
                      ("gym.envs.registration", registration)]:
        sys.modules[name] = mod


# -------------------------------------------------------------------------- pyglet
def _plain(v):
    """ctypes arrays / numpy scalars -> plain python numbers (for the call log)."""
    if isinstance(v, ctypes.Array):
        return [float(x) if isinstance(x, float) else x for x in v]
    if isinstance(v, (np.floating, np.integer)):
        return v.item()
    if isinstance(v, ctypes._SimpleCData):
        return v.value
    return v


def _make_gl_fn(name, ret=None):
    def fn(*args):
        if GL_LOG_ENABLED[0]:
            GL_LOG.append((name, tuple(_plain(a) for a in args)))
        return ret
    fn.__name__ = name
    return fn


def _install_pyglet(reference_root):
    import glob
    import os

    names = set()
    pat = re.compile(r"\b(glu?[A-Z][A-Za-z0-9]+|GL_[A-Z0-9_]+)\b")
    for f in glob.glob(os.path.join(reference_root, "gym_miniworld", "**", "*.py"), recursive=True):
        with open(f) as fh:
            names.update(pat.findall(fh.read()))

    gl = types.ModuleType("pyglet.gl")
    consts = sorted(n for n in names if n.startswith("GL_"))
    for i, n in enumerate(consts):
        setattr(gl, n, 0x1000 + i)
    for n in sorted(n for n in names if not n.startswith("GL_")):
        setattr(gl, n, _make_gl_fn(n))
    gl.glCheckFramebufferStatus = _make_gl_fn("glCheckFramebufferStatus", gl.GL_FRAMEBUFFER_COMPLETE)

    def glGetIntegerv(pname, out):
        try:
            out.value = 16
        except AttributeError:
            pass
    gl.glGetIntegerv = glGetIntegerv
    gl.GLfloat, gl.GLint, gl.GLuint = ctypes.c_float, ctypes.c_int, ctypes.c_uint
    gl.GLubyte, gl.GLushort = ctypes.c_ubyte, ctypes.c_ushort
    gl_info = types.ModuleType("pyglet.gl.gl_info")
    gl_info.have_extension = lambda name: True
    gl.gl_info = gl_info

    pyglet = types.ModuleType("pyglet")
    pyglet.options = {}
    pyglet.gl = gl

    window = types.ModuleType("pyglet.window")

    class Window:
        def __init__(self, *a, **k):
            pass

        def switch_to(self):
            pass

        def __getattr__(self, name):
            return lambda *a, **k: None
    window.Window = Window
    pyglet.window = window

    text = types.ModuleType("pyglet.text")

    class Label:
        def __init__(self, *a, **k):
            self.text = ""

        def draw(self):
            pass
    text.Label = Label
    pyglet.text = text

    image = types.ModuleType("pyglet.image")

    class _Tex:
        def __init__(self, w, h, path=None):
            self.width, self.height, self.target = w, h, gl.GL_TEXTURE_2D
            self.id = 1 + len(TEX_PATHS)      # a distinct id per loaded image: glBindTexture's argument names the file
            TEX_PATHS[self.id] = path

    class _ImgData:
        def get_data(self, fmt, pitch):
            return b""

    class _Img:
        def __init__(self, path):
            from PIL import Image
            with Image.open(path) as im:
                self.width, self.height = im.size
            self.path = path

        def get_texture(self):
            return _Tex(self.width, self.height, self.path)

        def get_image_data(self):
            return _ImgData()
    image.load = lambda path: _Img(path)
    pyglet.image = image

    graphics = types.ModuleType("pyglet.graphics")

    class _VertexList:
        """pyglet.graphics.vertex_list(count, ('v3f', data), ...): keeps the arrays it was given (objmesh.py:190-196); draw()
        records that it was drawn - the arrays themselves are looked up in VLISTS by the logged id"""
        def __init__(self, count, *data):
            self.count = count
            self.data = {fmt: np.array(arr, dtype=np.float32) for fmt, arr in data}
            self.uid = len(VLISTS)
            VLISTS.append(self)

        def draw(self, mode):
            if GL_LOG_ENABLED[0]:
                GL_LOG.append(("vlist_draw", (self.uid, mode)))
    graphics.vertex_list = lambda count, *data: _VertexList(count, *data)
    pyglet.graphics = graphics

    for name, mod in [("pyglet", pyglet), ("pyglet.gl", gl), ("pyglet.gl.gl_info", gl_info),
                      ("pyglet.window", window), ("pyglet.text", text),
                      ("pyglet.image", image), ("pyglet.graphics", graphics)]:
        sys.modules[name] = mod


def install(reference_root="/root/reference"):
    _install_gym()
    _install_pyglet(reference_root)
    if reference_root not in sys.path:
        sys.path.insert(0, reference_root)
