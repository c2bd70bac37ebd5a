"""Multi-GPU sharding of the env batch: one process per GPU, env-index ranges, no data-path
collective except the all-gather that concatenates the observation batch for the learner
(SURVEY.md 8e).  Backend-agnostic (RCCL on MI355X via backend "nccl"; gloo in the CPU tests).

The reference has no counterpart (its DP is one OS process per env over multiprocessing.Pipe,
vec_env/subproc_vec_env.py:36-56); env i keeps seed `base + i` whatever the number of GPUs
(pytorch-a2c-ppo-acktr/envs.py:36), so results do not depend on the sharding.
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


class ObsGatherer:
    """Double-buffered all-gather of a per-rank observation shard (equal shard sizes).

    push(obs) snapshots the shard into a staging buffer (the library overwrites `obs` on the next
    step) and starts the exchange asynchronously; the caller keeps stepping.  latest() waits
    (stream-side on GPU) for the newest exchange and returns the [world * n, ...] batch.

    Two exchange methods, same result:
      "ring"   one `all_gather_into_tensor` (RCCL ring / tree over the xGMI links);
      "direct" every rank sends its shard straight to each of its world-1 peers and receives theirs
               (one batched group of isend/irecv): xGMI is a full point-to-point mesh (7 links x
               ~153 GB/s per GPU), so the direct form moves shard/link_bw instead of the ring's
               7 x shard/link_bw per-link bound.
    "auto" times both on the first pushes (tune()) and keeps the faster; every rank takes the same
    decision (max over ranks).  Gather uint8, convert to float on the learner side.
    """

    def __init__(self, shard_shape, dtype, device, world_size, rank=0, group=None, method="auto"):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.world, self.rank = int(world_size), int(rank)
        self.n = shard_shape[0]
        self.staging = [torch.empty(shard_shape, dtype=dtype, device=device) for _ in range(2)]
        full = (shard_shape[0] * self.world,) + tuple(shard_shape[1:])
        self.gathered = [torch.empty(full, dtype=dtype, device=device) for _ in range(2)]
        self.work = [None, None]
        self.cur = 0
        self.is_cuda = torch.device(device).type == "cuda"
        self.method = method if self.world > 1 else "ring"
        self.tuned = {}

    # ------------------------------------------------------------------ the two exchange methods
    def _start(self, i, method):
        dist, torch = self.dist, self.torch
        if self.world == 1:
            self.gathered[i].copy_(self.staging[i], non_blocking=True)
            return None
        if method == "ring":
            return [dist.all_gather_into_tensor(self.gathered[i], self.staging[i], group=self.group, async_op=True)]
        ops = []
        g = self.gathered[i]
        for k in range(1, self.world):   # stagger the peers so that every link carries one transfer at a time
            dst = (self.rank + k) % self.world
            src = (self.rank - k) % self.world
            ops.append(dist.P2POp(dist.isend, self.staging[i], dst, group=self.group))
            ops.append(dist.P2POp(dist.irecv, g[src * self.n:(src + 1) * self.n], src, group=self.group))
        g[self.rank * self.n:(self.rank + 1) * self.n].copy_(self.staging[i], non_blocking=True)
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
        times = {}
        for m in ("ring", "direct"):
            try:
                self._wait(self._start(0, m))   # warm up (connection set-up)
                if self.is_cuda:
                    torch.cuda.synchronize()
                dist.barrier(group=self.group)
                t0 = time.perf_counter()
                for _ in range(iters):
                    self._wait(self._start(0, m))
                if self.is_cuda:
                    torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            except RuntimeError:   # a backend without batched point-to-point support: keep the collective
                if m == "ring":
                    raise
                dt = float("inf")
            t = torch.tensor([dt], dtype=torch.float64, device=self.staging[0].device if self.is_cuda else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)   # inf on any rank -> every rank keeps the ring
            times[m] = float(t.item()) / iters
        self.tuned = times
        self.method = min(times, key=times.get)
        return self.method

    # ----------------------------------------------------------------------------------- stream
    def push(self, obs):
        i = self.cur
        self._wait(self.work[i])
        self.staging[i].copy_(obs, non_blocking=True)
        if self.method == "auto":
            self.tune()
        self.work[i] = self._start(i, self.method)
        self.cur ^= 1
        return i

    def latest(self):
        i = self.cur ^ 1
        self._wait(self.work[i])
        self.work[i] = None
        return self.gathered[i]

    def drain(self):
        for i in range(2):
            self._wait(self.work[i])
            self.work[i] = None
