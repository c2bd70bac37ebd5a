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


class LazyInfos:
    """The `infos` of one step: behaves like the tuple of N dicts SubprocVecEnv returns (len, indexing, iteration,
    each item a dict), but a dict is only built when it is asked for - at thousands of envs per GPU building them
    all costs milliseconds per step, several times the device step.  The batched arrays behind the dicts are
    exposed too, so that a trainer can replace its per-env loop
        for info in infos: _feature.append(info["feature"])          (pytorch-a2c-ppo-acktr/main.py:612-617)
    by `infos.feature`:  goal_pos [N,3] float64 or None, feature [N,2] float64 or None, skipped bool [N] or None."""

    def __init__(self, n, goal_pos=None, feature=None, skipped=None, default_feature=False, health=None):
        self.n, self.goal_pos, self.skipped = n, goal_pos, skipped
        self.health = health   # CollectHealth: info['health'] (collecthealth.py:75), float64 [N] or None
        self._task_feature = feature is not None
        self.feature = feature if feature is not None else (np.zeros((n, 2)) if default_feature else None)

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return tuple(self[j] for j in range(*i.indices(self.n)))
        if i < 0:
            i += self.n
        if not 0 <= i < self.n:
            raise IndexError(i)
        if self.skipped is not None and self.skipped[i]:
            return _DUMMY_INFO
        info = {}
        if self.goal_pos is not None:
            info["goal_pos"] = self.goal_pos[i].copy()
        if self.health is not None:
            info["health"] = float(self.health[i])
        if self._task_feature:
            info["feature"] = self.feature[i].copy()
        elif self.feature is not None:
            info["feature"] = _DUMMY_INFO["feature"]
        return info

    def __iter__(self):
        return (self[i] for i in range(self.n))


class MiniWorldVecEnv(VecEnv):
    """N MiniWorld envs on one GPU behind the VecEnv protocol.

    transpose=True  -> observations are (3, W, H) like TransposeImage (envs.py:96-107)
    to_float=True   -> float32 0..255 device tensor like VecPyTorch (envs.py:117-130); False keeps uint8
    frame_stack=k   -> [N, k*3, W, H] ring like VecPyTorchFrameStack (envs.py:135-165), zeroed on done
    torch_api=True  -> actions arrive as LongTensor [N,1], rewards leave as CPU FloatTensor [N,1]
                       (VecPyTorch); False -> numpy in / numpy out like SubprocVecEnv
    feature_info    -> every info dict carries "feature" (length-2 zeros) as the fork's PPO loop
                       requires (pytorch-a2c-ppo-acktr/main.py:614-619)
    graph=True      -> after two eager steps the whole step (mwb_step's kernels on both streams, the frame-stack /
                       float pass, the copies into the pinned host mirrors) is captured into one HIP graph and
                       replayed (steps with a `mask` run eagerly).  Off by default: on ROCm 7.2 / MI355X a replayed graph
                       puts the main chain's bulk render on an auxiliary hardware queue and the side chain on the
                       capturing stream's, so the side chain loses its priority and head start (~4 % slower than the
                       ~10 eager launches); and the SECOND graph instantiated in a process lands on a hardware queue
                       where every kernel runs 2-5x longer (+40 % per step) unless GPU_MAX_HW_QUEUES=2 is exported
                       (profiles/r03_graph_replay_queues.txt, scripts/ab_graph_replay.py, graph_branch_overlap.py)
    """

    def __init__(self, env_id, num_envs, seed=1, device=0, domain_rand=False, transpose=True, to_float=True,
                 frame_stack=0, torch_api=True, feature_info=False, first_env_index=0, graph=False, **kwargs):
        import torch
        from .batch import BatchedMiniWorld
        from . import _lib
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
            # fused in the library: the render kernels write each new frame (uint8 -> float on the way out of LDS) into a
            # sliding window of planes and zero the history of the envs they regenerate: no stack pass at all (a replayed
            # graph would freeze the host-side window position: the shifting stack there)
            fuse = not graph and (to_float or (b.W * b.H) % 16 == 0)
            try:
                self.stackedobs = b.stack_enable(self.nstack, "float32" if to_float else "uint8", sliding=not graph, fused=fuse)
            except _lib.MwbError:
                if not fuse:
                    raise
                # observations rendered in tiles (frames too large for one workgroup's LDS, MWB_TILE): the sliding window, one pass
                self.stackedobs = b.stack_enable(self.nstack, "float32" if to_float else "uint8", sliding=not graph, fused=False)
        VecEnv.__init__(self, num_envs, Box(0, 255, shape, np.float32 if to_float else np.uint8), Discrete(b.n_actions))
        self.device = b.device
        self.feature_info = feature_info
        self._infos_plain = tuple((_DUMMY_INFO if feature_info else _EMPTY_INFO) for _ in range(num_envs))
        self._pending = False
        # host mirrors of the small per-step outputs: pinned, filled by asynchronous copies, ONE wait per step
        pin = lambda shape, dt: torch.empty(shape, dtype=dt, pin_memory=True)   # noqa: E731
        # done | reward | feature | goal_pos | ep_steps | reward64 in one allocation: only the prefix this view returns is copied
        off_ = b.pack_offsets
        need = off_["feature"] if torch_api else int(b.pack.numel())   # done + float32 reward ...
        if torch_api and (b.has_features or b.has_health):
            need = off_["goal_pos"]
        if torch_api and b.has_goal_pos:
            need = off_["ep_steps"]
        self._pack_need = int(need)
        self._h_pack = pin((self._pack_need,), torch.uint8)
        hp, off, n = self._h_pack.numpy(), b.pack_offsets, num_envs
        part = lambda key, dt, cnt: np.frombuffer(hp, dtype=dt, count=cnt, offset=off[key])   # noqa: E731
        self._h_done = part("done", np.uint8, n)
        self._h_rew = part("reward", np.float32, n) if torch_api else part("reward64", np.float64, n)
        self._h_goal = part("goal_pos", np.float64, 3 * n).reshape(n, 3) if b.has_goal_pos else None
        self._h_feat = part("feature", np.float32, 2 * n).reshape(n, 2) if (b.has_features or b.has_health) else None
        self._h_skip = pin((num_envs,), torch.uint8)
        self._ev = torch.cuda.Event()
        self._skip_host = None
        self._use_graph, self._graph, self._eager_steps = bool(graph), None, 0
        self._a_static = torch.zeros(num_envs, dtype=torch.int32, device=self.device)
        self._obs_static = None

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

    def _device_step(self, a, skip):
        """everything of a step that runs on the device: the stepper + renderer, the learner-side layout pass and the
        asynchronous copies of the small outputs into the pinned host mirrors"""
        b = self.batch
        if skip is None and a.dtype == self.torch.int64:
            b.step_longtensor(a)
        else:
            b.step(a, skip_mask=skip)
        obs = self._obs_out(done=b.done)
        self._h_pack.copy_(b.pack[:self._pack_need], non_blocking=True)   # one copy: the small outputs share an allocation
        return obs

    def step_async(self, actions, mask=None):
        torch = self.torch
        if torch.is_tensor(actions):
            a = actions.reshape(-1)   # VecPyTorch.step_async: actions.squeeze(1), envs.py:123
        else:
            a = torch.as_tensor(np.asarray(actions).reshape(-1))
        assert a.numel() == self.num_envs
        # the policy's LongTensor on this device is read by the step kernel in place (mwb_step_i64): no conversion pass;
        # anything else (and the captured-graph path, whose kernels read one fixed buffer) goes through the int32 buffer
        direct = (not self._use_graph) and a.dtype == torch.int64 and a.device == self.device and a.is_contiguous()
        if not direct:
            self._a_static.copy_(a, non_blocking=True)
        skip = None
        self._skip_host = None
        if isinstance(mask, np.ndarray):   # subproc_vec_env.py:59: only ndarray masks are honoured
            self._skip_host = mask.reshape(-1) != 0
            self._h_skip.copy_(torch.from_numpy(self._skip_host.astype(np.uint8)))
            skip = self._h_skip.to(self.device, non_blocking=True)
        if skip is None and self._use_graph and self._graph is None and self._eager_steps >= 2:
            self._capture()
        if skip is None and self._graph is not None:
            self._graph.replay()
            self._obs_now = self._obs_static
        else:
            self._obs_now = self._device_step(a if direct else self._a_static, skip)
            self._eager_steps += 1
        self._pending = True

    def _capture(self):
        torch = self.torch
        try:
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):   # records only; the step itself happens at the first replay
                self._obs_static = self._device_step(self._a_static, None)
            self._graph = g
        except Exception:   # capture is an optimisation: any failure leaves the eager path in charge
            self._graph, self._use_graph = None, False
            torch.cuda.synchronize(self.device)

    def step_wait(self):
        assert self._pending, "step_wait() without step_async()"
        self._pending = False
        b = self.batch
        obs = self._obs_now
        # the VecEnv contract returns numpy dones / CPU rewards / info dicts: the asynchronous copies into the pinned
        # host mirrors were enqueued with the step; ONE wait on the stream (the reference pays a pipe round trip per env)
        self._ev.record(self.torch.cuda.current_stream(self.device))
        self._ev.synchronize()
        dones = self._h_done.astype(bool)
        if self.torch_api:
            rews = self.torch.from_numpy(self._h_rew.copy()).unsqueeze(1)       # CPU FloatTensor [N,1], envs.py:129
        else:
            rews = self._h_rew.copy()
        sk = self._skip_host
        if b.has_health:   # CollectHealth: info['health'] (delivered in feature[:, 0])
            infos = LazyInfos(self.num_envs, health=self._h_feat[:, 0].astype(np.float64), skipped=sk, default_feature=self.feature_info)
        elif b.has_goal_pos:   # the T-maze family: info['goal_pos'] (tmaze.py:66,206) and, where produced, info['feature']
            infos = LazyInfos(self.num_envs, goal_pos=self._h_goal.copy(),
                              feature=self._h_feat.astype(np.float64) if self._h_feat is not None else None,
                              skipped=sk, default_feature=self.feature_info)
        elif sk is not None and sk.any():
            infos = LazyInfos(self.num_envs, skipped=sk, default_feature=self.feature_info)
        else:
            infos = self._infos_plain
        return obs, rews, dones, infos

    def close(self):
        self._graph = None
        self.batch.close()


def make_vec_envs(env_name, seed, num_processes, gamma=None, log_dir=None, add_timestep=False, device=None,
                  allow_early_resets=False, **kwargs):
    """Same signature as pytorch-a2c-ppo-acktr/envs.py:57 make_vec_envs; returns the VecPyTorch +
    4-frame-stack view the fork's main.py trains on ([N,12,80,60] float32 on `device`)."""
    import torch
    dev = torch.device(device if device is not None else "cuda:0")
    return MiniWorldVecEnv(env_name, num_processes, seed=seed, device=dev.index or 0, transpose=True, to_float=True,
                           frame_stack=4, torch_api=True, feature_info=True, **kwargs)
