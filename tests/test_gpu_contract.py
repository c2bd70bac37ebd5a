"""GPU: error behaviour of the C ABI and the bench.py / __graft_entry__ contracts the driver relies on."""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_refuses_misuse_loudly():
    import torch
    from gym_miniworld_amd import _lib
    from gym_miniworld_amd.batch import BatchedMiniWorld
    b = BatchedMiniWorld("MiniWorld-OneRoom-v0", num_envs=4)      # not seeded
    with pytest.raises(_lib.MwbError, match="mwb_seed"):
        b.reset()                                                    # the reference would seed from entropy; we refuse
    b.seed(3)
    b.reset()
    with pytest.raises(_lib.MwbError, match="stack"):
        b.stack_enable(4)                                            # HWC handle: the stack is channel-first
    with pytest.raises(_lib.MwbError):
        b.intersect(99, 0.0, 0.0)
    L = _lib.load()
    assert L.mwb_step(b.h, None, None, None) == -1 and b"null actions" in L.mwb_last_error()
    b.close()
    with pytest.raises(KeyError):
        BatchedMiniWorld("MiniWorld-PickupObjs-v0", num_envs=1)      # out of scope ids are named as such
    h = ctypes.c_void_p()
    cfg = _lib.MwbConfig()
    cfg.abi_version, cfg.task, cfg.num_envs, cfg.obs_width, cfg.obs_height, cfg.use_default_params = 1, 3, 4, 80, 60, 1
    cfg.task_args[0], cfg.task_args[1] = 30, 30                      # 900 cells: beyond the 160 KB LDS staging limit
    assert L.mwb_create(ctypes.byref(cfg), ctypes.byref(h)) == -1


def test_bench_emits_the_contract_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2",
                          "--envs-per-gpu", "256", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["unit"] == "env-steps/s" and d["value"] > 0 and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12


def test_graph_capture_of_a_step():
    """mwb_step (incl. the fork/join onto the side stream) can be captured into a HIP graph and replayed."""
    import torch
    from gym_miniworld_amd.batch import BatchedMiniWorld
    n = 256
    a = BatchedMiniWorld("MiniWorld-FourRooms-v0", num_envs=n, seed=4)
    b = BatchedMiniWorld("MiniWorld-FourRooms-v0", num_envs=n, seed=4)
    a.reset(); b.reset()
    acts = torch.zeros(n, dtype=torch.int32, device="cuda")
    a.step(acts); b.step(acts)            # eager warm-up, both handles in lock-step
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):             # capture only records; nothing executes
        b.step(acts)
    gen = torch.Generator().manual_seed(1)
    for t in range(60):
        acts.copy_(torch.randint(0, 3, (n,), generator=gen, dtype=torch.int32))
        a.step(acts)
        g.replay()
        assert torch.equal(a.obs, b.obs) and torch.equal(a.reward64, b.reward64) and torch.equal(a.done, b.done), t
    a.close(); b.close()


def test_cost_ordered_dispatch_renders_every_env_exactly_once():
    """mwb_step dispatches the bulk render's workgroups by decreasing measured frame cost (a map rebuilt every step
    on the side stream, the cheapest envs as half-frame workgroups).  Whatever the map, every env must be rendered
    exactly once: the observations a step leaves behind equal a plain mwb_render of the same state, for every env."""
    import torch
    from gym_miniworld_amd.batch import BatchedMiniWorld
    for env_id, n, layout in (("MiniWorld-Maze-v0", 3000, "HWC"), ("MiniWorld-FourRooms-v0", 1500, "CWH")):
        b = BatchedMiniWorld(env_id, num_envs=n, seed=11, domain_rand=True, want_depth=True, layout=layout)
        b.reset()
        g = torch.Generator().manual_seed(0)
        for t in range(40):
            b.step(torch.randint(0, 3, (n,), generator=g, dtype=torch.int32))
            if t % 8 == 7:
                obs_step, dep_step = b.obs.clone(), b.depth.clone()
                b.obs.zero_()   # a workgroup that never ran would leave zeros behind
                b.render()
                assert torch.equal(b.obs, obs_step) and torch.equal(b.depth, dep_step), (env_id, t)
                b.obs.zero_()
                b.step(torch.randint(0, 3, (n,), generator=g, dtype=torch.int32))
                assert int((b.obs.reshape(n, -1).max(dim=1).values == 0).sum()) == 0, (env_id, t, "an env was not rendered")
        b.close()
