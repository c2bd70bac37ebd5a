"""Domain-randomisation parameter table with the interface of the reference's
gym_miniworld/params.py:10-123 (DomainParams.set / get_max / no_random / copy, DEFAULT_PARAMS),
so code that customises `params=` for a reference env (e.g. envs/oneroom.py:52-66) keeps working.
The table is handed to the HIP library as 13 x (default[3], min[3], max[3]) doubles; sampling
itself happens on the GPU from each env's MT19937 stream.
"""
from collections import namedtuple
from copy import deepcopy

import numpy as np

DomainParam = namedtuple("DomainParam", ["default", "min", "max", "type"])

# order of the C ABI table (include/miniworld_batch.h MWB_P_*), same as params.py:110-123
PARAM_ORDER = ["sky_color", "light_pos", "light_color", "light_ambient", "obj_color_bias", "forward_step",
               "forward_drift", "turn_step", "bot_radius", "cam_pitch", "cam_fov_y", "cam_height", "cam_fwd_disp"]


class DomainParams:
    def __init__(self):
        self.params = {}

    def copy(self):
        return deepcopy(self)

    def no_random(self):
        c = self.copy()
        for name, p in c.params.items():
            c.params[name] = DomainParam(p.default, p.default, p.default, p.type)
        return c

    def set(self, name, default, min=None, max=None, type="float"):
        conv = lambda v: np.array(v, dtype=float) if isinstance(v, (list, tuple, np.ndarray)) else v  # noqa: E731
        default, min, max = conv(default), conv(min), conv(max)
        if min is None:
            min = default
        if max is None:
            max = default
        assert np.all(np.greater_equal(max, default)) and np.all(np.greater_equal(default, min))
        if name in self.params:
            assert type == self.params[name].type
            assert np.shape(default) == np.shape(self.params[name].default)
        self.params[name] = DomainParam(default, min, max, type)

    def get_max(self, name):
        assert name in self.params, name
        return self.params[name].max

    def to_table(self):
        """-> float64 [13, 9] (default[3], min[3], max[3]) for mwb_config.params"""
        t = np.zeros((len(PARAM_ORDER), 9))
        for i, name in enumerate(PARAM_ORDER):
            p = self.params[name]
            if p.type != "float":
                raise NotImplementedError("only float parameters exist in the reference table")
            for j, v in enumerate((p.default, p.min, p.max)):
                v = np.atleast_1d(np.asarray(v, dtype=float))
                t[i, j * 3:j * 3 + len(v)] = v
        return t


# name -> (default, min, max): the values of the reference table (params.py:110-123), vectors as tuples
_DEFAULT_TABLE = {
    "sky_color": ((0.25, 0.82, 1), (0.1, 0.1, 0.1), (1.0, 1.0, 1.0)),
    "light_pos": ((0, 2.5, 0), (-40, 2.5, -40), (40, 5, 40)),
    "light_color": ((0.7,) * 3, (0.45,) * 3, (0.8,) * 3),
    "light_ambient": ((0.45,) * 3, (0.35,) * 3, (0.55,) * 3),
    "obj_color_bias": ((0,) * 3, (-0.2,) * 3, (0.2,) * 3),
    "forward_step": (0.15, 0.12, 0.17),
    "forward_drift": (0, -0.05, 0.05),
    "turn_step": (15, 10, 20),
    "bot_radius": (0.4, 0.38, 0.42),
    "cam_pitch": (0, -5, 5),
    "cam_fov_y": (60, 55, 65),
    "cam_height": (1.5, 1.45, 1.55),
    "cam_fwd_disp": (0, -0.05, 0.10),
}


def _from_table(base, table):
    p = base.copy() if base is not None else DomainParams()
    for name, (dflt, lo, hi) in table.items():
        as_list = lambda v: list(v) if isinstance(v, tuple) else v  # noqa: E731
        p.set(name, as_list(dflt), as_list(lo), as_list(hi))
    return p


DEFAULT_PARAMS = _from_table(None, _DEFAULT_TABLE)


def sim_to_real_params(push=False):
    """envs/simtorealgoto.py:8-18 / simtorealpush.py:8-18: a robot about 15 cm tall with a Pi camera.  The two
    files differ only in bot_radius (0.4 +- 0.02, unused, vs 0.11, which SimToRealPush uses to keep boxes off the walls)."""
    return _from_table(DEFAULT_PARAMS, {
        "forward_step": (0.035, 0.028, 0.042),
        "forward_drift": (0, -0.005, 0.005),
        "turn_step": (17, 13, 21),
        "bot_radius": (0.11, 0.11, 0.11) if push else (0.4, 0.38, 0.42),
        "cam_pitch": (-10, -15, -3),
        "cam_fov_y": (49, 45, 55),
        "cam_height": (0.18, 0.17, 0.19),
        "cam_fwd_disp": (0, -0.02, 0.02),
    })
