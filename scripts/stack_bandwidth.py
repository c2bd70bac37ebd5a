import sys, time
sys.path.insert(0, '.')
import torch
from gym_miniworld_amd.batch import BatchedMiniWorld
n = 8192
for dt in ("float32", "uint8"):
    b = BatchedMiniWorld("MiniWorld-Maze-v0", num_envs=n, seed=1, layout="CWH")
    b.reset(); st = b.stack_enable(4, dt); b.stack_update(True)
    a = torch.randint(0, 3, (n,), dtype=torch.int32, device="cuda")
    b.step(a)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3): b.stack_update(False)
    ev0.record()
    K = 20
    for _ in range(K): b.stack_update(False)
    ev1.record(); torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / K
    es = 4 if dt == "float32" else 1
    by = n * 4800 * (9 * es + 3 + 12 * es)
    print(dt, "stack update %.3f ms  bytes %.1f MB  %.0f GB/s  (%.1f%% of 8 TB/s)" % (ms, by / 1e6, by / ms / 1e6, by / ms / 1e6 / 80))
    b.close()
