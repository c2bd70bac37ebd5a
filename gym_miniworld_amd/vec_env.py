"""VecEnv front-end with the contract of the reference's RL harness.

Mirrors pytorch-a2c-ppo-acktr/vec_env/__init__.py:3-63 (VecEnv), vec_env/subproc_vec_env.py:36-97
(SubprocVecEnv incl. the fork's `mask` / 'dummy' command) and the wrapper stack of
pytorch-a2c-ppo-acktr/envs.py:57-165 (make_vec_envs -> TransposeImage -> VecPyTorch ->
VecPyTorchFrameStack) - but the N environments live in one HIP handle instead of N processes, and
observations never leave the GPU.
"""
from abc import ABC, abstractmethod

import numpy as np


class Discrete:
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.int64

    def __repr__(self):
        return "Discrete(%d)" % self.n


class Box:
    def __init__(self, low, high, shape, dtype):
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.low = np.full(self.shape, low, dtype=self.dtype)
        self.high = np.full(self.shape, high, dtype=self.dtype)

    def __repr__(self):
        return "Box%s" % (self.shape,)


class VecEnv(ABC):
    """vec_env/__init__.py:3-63"""

    def __init__(self, num_envs, observation_space, action_space):
        self.num_envs = num_envs
        self.observation_space = observation_space
        self.action_space = action_space

    @abstractmethod
    def reset(self):
        pass

    @abstractmethod
    def step_async(self, actions, mask=None):
        pass

    @abstractmethod
    def step_wait(self):
        pass

    @abstractmethod
    def close(self):
        pass

    def step(self, actions, mask=None):
        self.step_async(actions, mask)
        return self.step_wait()

    def render(self):
        pass


_EMPTY_INFO = {}
_DUMMY_INFO = {"feature": np.array([0, 0])}   # subproc_vec_env.py:30


class MiniWorldVecEnv(VecEnv):
    """N MiniWorld envs on one GPU behind the VecEnv protocol.

    transpose=True  -> observations are (3, W, H) like TransposeImage (envs.py:96-107)
    to_float=True   -> float32 0..255 device tensor like VecPyTorch (envs.py:117-130); False keeps uint8
    frame_stack=k   -> [N, k*3, W, H] ring like VecPyTorchFrameStack (envs.py:135-165), zeroed on done
    torch_api=True  -> actions arrive as LongTensor [N,1], rewards leave as CPU FloatTensor [N,1]
                       (VecPyTorch); False -> numpy in / numpy out like SubprocVecEnv
    feature_info    -> every info dict carries "feature" (length-2 zeros) as the fork's PPO loop
                       requires (pytorch-a2c-ppo-acktr/main.py:614-619)
    """

    def __init__(self, env_id, num_envs, seed=1, device=0, domain_rand=False, transpose=True, to_float=True,
                 frame_stack=0, torch_api=True, feature_info=False, first_env_index=0, **kwargs):
        import torch
        from .batch import BatchedMiniWorld
        self.torch = torch
        self.batch = BatchedMiniWorld(env_id, num_envs=num_envs, seed=seed, domain_rand=domain_rand, device=device,
                                      layout="CWH" if transpose else "HWC", first_env_index=first_env_index, **kwargs)
        b = self.batch
        shape = (3, b.W, b.H) if transpose else (b.H, b.W, 3)
        self.to_float, self.torch_api, self.nstack = to_float, torch_api, int(frame_stack)
        self.shape_dim0 = shape[0]
        if self.nstack:
            assert transpose, "frame stacking follows VecPyTorchFrameStack: channel-first observations"
            shape = (shape[0] * self.nstack,) + shape[1:]
            # fused in the library: shift + zero-on-done + append + uint8->float in one HBM pass
            self.stackedobs = b.stack_enable(self.nstack, "float32" if to_float else "uint8")
        VecEnv.__init__(self, num_envs, Box(0, 255, shape, np.float32 if to_float else np.uint8), Discrete(b.n_actions))
        self.device = b.device
        self.feature_info = feature_info
        self._infos_plain = tuple((_DUMMY_INFO if feature_info else _EMPTY_INFO) for _ in range(num_envs))
        self._pending = False

    # ---------------------------------------------------------------------------------- helpers
    def _obs_out(self, done=None):
        if self.nstack:   # VecPyTorchFrameStack.reset / step_wait, envs.py:149-162
            return self.batch.stack_update(after_reset=done is None)
        obs = self.batch.obs
        return obs.float() if self.to_float else obs   # torch.from_numpy(obs).float().to(device), envs.py:119,128

    # ----------------------------------------------------------------------------------- VecEnv
    def reset(self):
        self.batch.reset()
        return self._obs_out()

    def step_async(self, actions, mask=None):
        torch = self.torch
        if torch.is_tensor(actions):
            a = actions.reshape(-1)   # VecPyTorch.step_async: actions.squeeze(1), envs.py:123
        else:
            a = torch.as_tensor(np.asarray(actions).reshape(-1))
        skip = None
        if isinstance(mask, np.ndarray):   # subproc_vec_env.py:59: only ndarray masks are honoured
            skip = torch.as_tensor(mask != 0)
        self._skip = skip
        self.batch.step(a, skip_mask=skip)
        self._pending = True

    def step_wait(self):
        assert self._pending, "step_wait() without step_async()"
        self._pending = False
        b = self.batch
        obs = self._obs_out(done=b.done)
        dones = b.done.cpu().numpy().astype(bool)   # host sync: the VecEnv contract returns numpy dones
        if self.torch_api:
            rews = b.reward.cpu().unsqueeze(1)       # CPU FloatTensor [N,1], envs.py:129
        else:
            rews = b.reward64.cpu().numpy()
        sk = self._skip.cpu().numpy() if self._skip is not None else None
        if b.has_goal_pos:   # the T-maze family: info['goal_pos'] (tmaze.py:66,206) and, where produced, info['feature']
            gp = b.goal_pos.cpu().numpy()
            ft = b.feature.cpu().numpy().astype(np.float64) if b.has_features else None
            infos = []
            for i in range(self.num_envs):
                if sk is not None and sk[i]:
                    infos.append(_DUMMY_INFO)
                    continue
                info = {"goal_pos": gp[i].copy()}
                if ft is not None:
                    info["feature"] = ft[i].copy()
                elif self.feature_info:
                    info["feature"] = _DUMMY_INFO["feature"]
                infos.append(info)
            infos = tuple(infos)
        elif sk is not None:
            infos = tuple(_DUMMY_INFO if sk[i] else self._infos_plain[i] for i in range(self.num_envs))
        else:
            infos = self._infos_plain
        return obs, rews, dones, infos

    def close(self):
        self.batch.close()


def make_vec_envs(env_name, seed, num_processes, gamma=None, log_dir=None, add_timestep=False, device=None,
                  allow_early_resets=False, **kwargs):
    """Same signature as pytorch-a2c-ppo-acktr/envs.py:57 make_vec_envs; returns the VecPyTorch +
    4-frame-stack view the fork's main.py trains on ([N,12,80,60] float32 on `device`)."""
    import torch
    dev = torch.device(device if device is not None else "cuda:0")
    return MiniWorldVecEnv(env_name, num_processes, seed=seed, device=dev.index or 0, transpose=True, to_float=True,
                           frame_stack=4, torch_api=True, feature_info=True, **kwargs)
