"""Multi-GPU sharding of the env batch: one process per GPU, env-index ranges, no data-path
collective except the exchange that concatenates every rank's step outputs for the learner and the
scatter that brings the learner's actions back.  Backend-agnostic (RCCL on MI355X via
backend "nccl"; gloo in the CPU tests).

The reference's counterpart is one OS process per env over multiprocessing.Pipe: every step each worker
sends (obs, reward, done, info) up and receives its action (vec_env/subproc_vec_env.py:5-33, 58-75);
env i keeps seed `base + i` whatever the number of GPUs (pytorch-a2c-ppo-acktr/envs.py:36), so results
do not depend on the sharding.
"""
import os
import time


def dist_env():
    """(rank, local_rank, world_size) from the torchrun environment (1-process default)."""
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)),
            int(os.environ.get("WORLD_SIZE", 1)))


def shard_range(total_envs, rank, world_size):
    """Contiguous env-index range [first, first+count) of `rank`; remainders go to the low ranks."""
    base, rem = divmod(int(total_envs), int(world_size))
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


AUX_WORDS = 8   # float64 words per env in the small pack: reward, done, ep_steps, feature[2], goal_pos[3]


class ShardExchange:
    """Double-buffered exchange of a rank's per-step outputs (equal shard sizes): the uint8 observation shard and
    an `aux` pack [n, 8] float64 = reward (the reference's Python float), done, ep_steps, info['feature'][2],
    info['goal_pos'][3] - what a SubprocVecEnv worker sends up its pipe every step (subproc_vec_env.py:10-14).

    push(obs, aux) snapshots both into staging buffers (the library overwrites its outputs on the next step) and
    starts the exchange asynchronously - both tensors travel in ONE round (one group of transfers); the caller keeps
    stepping.  latest() waits (stream-side on GPU) for the newest round and returns ([world*n, ...] obs,
    [world*n, 8] aux); previous() does the same for the round before, which is what a pipelined learner consumes
    while the current step renders.  scatter_actions() is the way back.

    Two exchange methods, same result:
      "ring"   `all_gather_into_tensor` (RCCL ring / tree over the xGMI links);
      "direct" every rank sends its shard straight to each of its world-1 peers and receives theirs (one batched
               group of isend/irecv): xGMI is a full point-to-point mesh (7 links x ~153 GB/s per GPU), so the
               direct form moves shard/link_bw instead of the ring's 7 x shard/link_bw per-link bound.
      "learner" only rank `learner` receives (every other rank sends its shard straight to it): a single learner process needs
               1 / world of the bytes an all-gather moves; latest() / previous() return the full batch on the learner and the
               rank's own shard elsewhere.
    "auto" times ring and direct on the first push (tune()) and keeps the faster; every rank takes the same decision (max
    over ranks).  verify() checks, outside any timed region, that the last round delivered every rank's shard (per-shard
    checksums exchanged separately).  Whether the backend offers batched point-to-point is agreed on by all ranks BEFORE any transfer
    is posted, so a rank can never be left waiting for peers that fell back.  Gather uint8, convert on the learner side.
    """

    def __init__(self, shard_shape, dtype, device, world_size, rank=0, group=None, method="auto", aux_words=AUX_WORDS, learner=0):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.world, self.rank = int(world_size), int(rank)
        self.n = shard_shape[0]
        self.device = torch.device(device)
        self.is_cuda = self.device.type == "cuda"
        mk = lambda shape, dt: [torch.empty(shape, dtype=dt, device=device) for _ in range(2)]   # noqa: E731
        self.staging = mk(shard_shape, dtype)
        self.gathered = mk((self.n * self.world,) + tuple(shard_shape[1:]), dtype)
        self.aux_words = int(aux_words)
        self.aux_staging = mk((self.n, self.aux_words), torch.float64)
        self.aux_gathered = mk((self.n * self.world, self.aux_words), torch.float64)
        self.work = [None, None]
        self.cur = 0
        if method not in ("auto", "ring", "direct", "learner"):
            raise ValueError("method must be auto, ring, direct or learner")
        self.learner = int(learner)
        self.method = method if self.world > 1 else "ring"
        self.tuned = {}
        self._direct_ok = None

    # ------------------------------------------------------------------ the two exchange methods
    def direct_supported(self):
        """All ranks agree (MIN over ranks of a local probe that posts nothing) on batched point-to-point."""
        if self._direct_ok is None:
            torch, dist = self.torch, self.dist
            local = 1 if (hasattr(dist, "batch_isend_irecv") and dist.get_backend(self.group) in ("nccl", "gloo")) else 0
            t = torch.tensor([local], dtype=torch.int32, device=self.device if self.is_cuda else "cpu")
            if self.world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
            self._direct_ok = bool(int(t.item()))
        return self._direct_ok

    def _start(self, i, method):
        dist = self.dist
        pairs = ((self.staging[i], self.gathered[i]), (self.aux_staging[i], self.aux_gathered[i]))
        if self.world == 1:
            for src, dst in pairs:
                dst.copy_(src, non_blocking=True)
            return None
        if method == "ring":
            return [dist.all_gather_into_tensor(dst, src, group=self.group, async_op=True) for src, dst in pairs]
        ops = []
        if method == "learner":   # gather to one rank: the others only send
            if self.rank == self.learner:
                for frm in range(self.world):
                    if frm != self.rank:
                        for _, dst in pairs:
                            ops.append(dist.P2POp(dist.irecv, dst[frm * self.n:(frm + 1) * self.n], frm, group=self.group))
                for src, dst in pairs:
                    dst[self.rank * self.n:(self.rank + 1) * self.n].copy_(src, non_blocking=True)
            else:
                for src, _ in pairs:
                    ops.append(dist.P2POp(dist.isend, src, self.learner, group=self.group))
            return dist.batch_isend_irecv(ops) if ops else None
        for k in range(1, self.world):   # stagger the peers so that every link carries one transfer at a time
            to, frm = (self.rank + k) % self.world, (self.rank - k) % self.world
            for src, dst in pairs:
                ops.append(dist.P2POp(dist.isend, src, to, group=self.group))
                ops.append(dist.P2POp(dist.irecv, dst[frm * self.n:(frm + 1) * self.n], frm, group=self.group))
        for src, dst in pairs:
            dst[self.rank * self.n:(self.rank + 1) * self.n].copy_(src, non_blocking=True)
        return dist.batch_isend_irecv(ops)

    @staticmethod
    def _wait(works):
        if works:
            for w in works:
                w.wait()

    def tune(self, iters=4):
        """Pick the faster exchange method (all ranks agree). Uses buffer 0 with whatever it holds."""
        if self.world == 1 or self.method != "auto":
            return self.method
        torch, dist = self.torch, self.dist
        methods = ("ring", "direct") if self.direct_supported() else ("ring",)
        times = {}
        for m in methods:   # no per-rank error handling here: a failure must surface on every rank, not strand the others
            self._wait(self._start(0, m))   # warm up (connection set-up)
            if self.is_cuda:
                torch.cuda.synchronize()
            dist.barrier(group=self.group)
            t0 = time.perf_counter()
            for _ in range(iters):
                self._wait(self._start(0, m))
            if self.is_cuda:
                torch.cuda.synchronize()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=self.device if self.is_cuda else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            times[m] = float(t.item()) / iters
        self.tuned = times
        self.method = min(times, key=times.get)
        return self.method

    # ----------------------------------------------------------------------------------- stream
    def pack_aux(self, env, out):
        """[n, 8] float64 <- the small per-step outputs of a BatchedMiniWorld (one fused device copy per field)."""
        out[:, 0].copy_(env.reward64, non_blocking=True)
        out[:, 1].copy_(env.done, non_blocking=True)
        out[:, 2].copy_(env.ep_steps, non_blocking=True)
        out[:, 3:5].copy_(env.feature, non_blocking=True)
        out[:, 5:8].copy_(env.goal_pos, non_blocking=True)
        return out

    def push(self, obs, aux=None, env=None):
        """aux: [n, aux_words] float64 tensor, or pass env= (a BatchedMiniWorld) to have it packed from its outputs."""
        i = self.cur
        self._wait(self.work[i])
        self.staging[i].copy_(obs, non_blocking=True)
        if env is not None:
            self.pack_aux(env, self.aux_staging[i])
        elif aux is not None:
            self.aux_staging[i].copy_(aux, non_blocking=True)
        if self.method == "auto":
            self.tune()
        self.work[i] = self._start(i, self.method)
        self.cur ^= 1
        return i

    def _take(self, i):
        self._wait(self.work[i])
        self.work[i] = None
        if self.method == "learner" and self.world > 1 and self.rank != self.learner:
            return self.staging[i], self.aux_staging[i]   # only the learner holds the whole batch
        return self.gathered[i], self.aux_gathered[i]

    def measure(self, iters=4):
        """seconds per exchange of the method in use, unoverlapped (max over ranks); buffer 0 with whatever it holds"""
        if self.world == 1:
            return 0.0
        torch, dist = self.torch, self.dist
        self._wait(self._start(0, self.method))
        if self.is_cuda:
            torch.cuda.synchronize()
        dist.barrier(group=self.group)
        t0 = time.perf_counter()
        for _ in range(iters):
            self._wait(self._start(0, self.method))
        if self.is_cuda:
            torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=self.device if self.is_cuda else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return float(t.item()) / iters

    def verify(self, which="latest"):
        """Did the round deliver EVERY rank's shard?  Each rank checksums what it staged (observation bytes and aux words); the
        checksums travel by a separate tiny all-gather; every rank that holds the batch compares each shard of it with its
        owner's checksum.  Returns (ok on all ranks, ranks seen).  Not for timed regions (host synchronisation)."""
        torch, dist = self.torch, self.dist
        i = self.cur ^ 1 if which == "latest" else self.cur
        full, faux = self._take(i)
        dev = self.device if self.is_cuda else "cpu"
        cs = lambda o, a: torch.stack([o.reshape(-1).to(torch.int64).sum().to(torch.float64), a.reshape(-1).sum()])   # noqa: E731
        mine = cs(self.staging[i], self.aux_staging[i]).to(dev)
        if self.world == 1:
            return bool(torch.equal(cs(full, faux).to(dev), mine)), 1
        everyone = torch.empty((self.world * 2,), dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(everyone, mine, group=self.group)
        everyone = everyone.view(self.world, 2)
        ok = 1
        if not (self.method == "learner" and self.rank != self.learner):
            for k in range(self.world):
                got = cs(full[k * self.n:(k + 1) * self.n], faux[k * self.n:(k + 1) * self.n]).to(dev)
                ok &= int(torch.equal(got, everyone[k]))
        t = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return bool(int(t.item())), int(everyone.shape[0])

    def latest(self):
        return self._take(self.cur ^ 1)

    def previous(self):
        """the round before the newest one: complete (or nearly) by the time the newest step has been enqueued"""
        return self._take(self.cur)

    def drain(self):
        for i in range(2):
            self._wait(self.work[i])
            self.work[i] = None

    # ------------------------------------------------------------------------------------ the way back
    def scatter_actions(self, actions_all, src=0):
        """The learner on rank `src` holds actions for all world*n envs ([world*n] or [world*n, 1], any int dtype);
        every rank gets its own [n] int32 slice (4 B/env: one small broadcast, then a local slice - cheaper to
        reason about than a scatter and supported by every backend).  On other ranks `actions_all` may be None."""
        torch, dist = self.torch, self.dist
        buf = torch.empty(self.n * self.world, dtype=torch.int32, device=self.device if self.is_cuda else "cpu")
        if self.rank == src:
            buf.copy_(torch.as_tensor(actions_all).reshape(-1).to(torch.int32))
        if self.world > 1:
            dist.broadcast(buf, src=src, group=self.group)
        return buf[self.rank * self.n:(self.rank + 1) * self.n]




def unpack_aux(aux):
    """[N, 8] float64 -> dict of views: reward [N] f64, done [N] bool, ep_steps [N] i32, feature [N,2], goal_pos [N,3]"""
    import torch
    return {"reward": aux[:, 0], "done": aux[:, 1] != 0, "ep_steps": aux[:, 2].to(torch.int32),
            "feature": aux[:, 3:5], "goal_pos": aux[:, 5:8]}
