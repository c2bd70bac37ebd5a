"""gym_miniworld_amd - MI355X-native batched MiniWorld stepper + renderer.

Drop-in for the hot path of mjsargent/gym-miniworld (MiniWorldEnv.step / render_obs behind the
Gym and VecEnv protocols) for the Hallway / OneRoom / FourRooms / Maze tasks and the T-maze
family (TMaze, TMazeLeft / Right, TMazeDynamic, the two-box variants).  See DESIGN.md.
"""
from .params import DEFAULT_PARAMS, DomainParams  # noqa: F401

__all__ = ["BatchedMiniWorld", "MiniWorldVecEnv", "MiniWorldEnv", "make", "make_vec_envs", "DEFAULT_PARAMS",
           "DomainParams", "ENV_SPECS"]


def __getattr__(name):   # lazy: importing the package must not need torch / the GPU
    if name in ("BatchedMiniWorld", "ENV_SPECS"):
        from . import batch
        return getattr(batch, name)
    if name in ("MiniWorldVecEnv", "make_vec_envs", "VecEnv"):
        from . import vec_env
        return getattr(vec_env, name)
    if name in ("MiniWorldEnv", "make"):
        from . import env
        return getattr(env, name)
    raise AttributeError(name)
