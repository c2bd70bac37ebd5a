"""Multi-GPU sharding of the env batch: one process per GPU, env-index ranges, no data-path
collective except the all-gather that concatenates the observation batch for the learner
(SURVEY.md 8e).  Backend-agnostic (RCCL on MI355X via backend "nccl"; gloo in the CPU tests).

The reference has no counterpart (its DP is one OS process per env over multiprocessing.Pipe,
vec_env/subproc_vec_env.py:36-56); env i keeps seed `base + i` whatever the number of GPUs
(pytorch-a2c-ppo-acktr/envs.py:36), so results do not depend on the sharding.
"""
import os


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
    """Double-buffered all-gather of a per-rank observation shard.

    push(obs) snapshots the shard into a staging buffer (the library overwrites `obs` on the next
    step) and starts the collective on the communication stream; the caller keeps stepping.
    latest() waits (stream-side on GPU) for the newest gather and returns the [world*n, ...] batch.
    xGMI is point-to-point, so RCCL's ring all-gather is per-link bound (~7x shard / 153 GB/s for
    8 GPUs): gather uint8, convert to float on the learner side.
    """

    def __init__(self, shard_shape, dtype, device, world_size, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.world = world_size
        self.staging = [torch.empty(shard_shape, dtype=dtype, device=device) for _ in range(2)]
        full = (shard_shape[0] * world_size,) + tuple(shard_shape[1:])
        self.gathered = [torch.empty(full, dtype=dtype, device=device) for _ in range(2)]
        self.work = [None, None]
        self.cur = 0
        self.is_cuda = torch.device(device).type == "cuda"

    def push(self, obs):
        i = self.cur
        if self.work[i] is not None:
            self.work[i].wait()
        self.staging[i].copy_(obs, non_blocking=True)
        if self.world > 1:
            self.work[i] = self.dist.all_gather_into_tensor(self.gathered[i], self.staging[i], group=self.group,
                                                            async_op=True)
        else:
            self.gathered[i].copy_(self.staging[i], non_blocking=True)
        self.cur ^= 1
        return i

    def latest(self):
        i = self.cur ^ 1
        if self.work[i] is not None:
            self.work[i].wait()
            self.work[i] = None
        return self.gathered[i]

    def drain(self):
        for i in range(2):
            if self.work[i] is not None:
                self.work[i].wait()
                self.work[i] = None
