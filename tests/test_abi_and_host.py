"""CPU: the C-ABI library loads and exports every symbol include/miniworld_batch.h declares (no
compute without a GPU), refuses loudly to run without a device, and the host-side mirror of the
reference interface (params table, env registry, action stream, sharding) behaves."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    from gym_miniworld_amd import build
    build.build()
    from gym_miniworld_amd import _lib
    return _lib


def header_functions():
    src = open(os.path.join(ROOT, "include", "miniworld_batch.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mwb_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(built):
    names = header_functions()
    assert len(names) >= 16 and set(names) == set(built.EXPORTS)
    L = built.load()
    for n in names:
        assert getattr(L, n) is not None, n
    assert L.mwb_abi_version() == built.ABI_VERSION
    assert isinstance(L.mwb_last_error(), bytes)


def test_struct_layout_matches_header(built):
    # mwb_config: 10 int32, 4 doubles, 2 int32, 13*9 doubles
    assert ctypes.sizeof(built.MwbConfig) == 10 * 4 + 4 * 8 + 2 * 4 + 13 * 9 * 8
    assert ctypes.sizeof(built.MwbOutputs) == 14 * 8 and ctypes.sizeof(built.MwbState) == 30 * 8


def test_create_fails_loudly_without_gpu_or_with_bad_args(built):
    import torch
    L = built.load()
    h = ctypes.c_void_p()
    cfg = built.MwbConfig()
    cfg.abi_version = built.ABI_VERSION + 7
    assert L.mwb_create(ctypes.byref(cfg), ctypes.byref(h)) == -1 and b"abi_version" in L.mwb_last_error()
    cfg.abi_version = built.ABI_VERSION
    cfg.num_envs = 0
    assert L.mwb_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    cfg.num_envs, cfg.task, cfg.obs_width, cfg.obs_height, cfg.use_default_params = 4, 1, 80, 60, 1
    rc = L.mwb_create(ctypes.byref(cfg), ctypes.byref(h))
    if not torch.cuda.is_available():
        assert rc == -2 and b"no HIP device" in L.mwb_last_error()   # no CPU fallback exists
        from gym_miniworld_amd.batch import BatchedMiniWorld
        with pytest.raises(built.MwbError):
            BatchedMiniWorld("MiniWorld-OneRoom-v0", num_envs=2, seed=0)
    else:
        assert rc == 0
        L.mwb_destroy(h)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "gym_miniworld_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("no oracle code", ""), (f, "product must not reference oracle/")


def test_params_table_mirrors_reference_defaults():
    from gym_miniworld_amd.params import DEFAULT_PARAMS, PARAM_ORDER
    t = DEFAULT_PARAMS.to_table()
    assert t.shape == (13, 9)
    assert list(t[PARAM_ORDER.index("forward_step")][[0, 3, 6]]) == [0.15, 0.12, 0.17]   # params.py:116
    assert list(t[PARAM_ORDER.index("light_pos")]) == [0, 2.5, 0, -40, 2.5, -40, 40, 5, 40]   # params.py:112
    assert DEFAULT_PARAMS.get_max("forward_step") == 0.17
    nr = DEFAULT_PARAMS.no_random()
    nr.set("forward_step", 0.7)
    tt = nr.to_table()
    assert list(tt[PARAM_ORDER.index("forward_step")][[0, 3, 6]]) == [0.7, 0.7, 0.7]
    assert list(tt[PARAM_ORDER.index("turn_step")][[0, 3, 6]]) == [15, 15, 15]
    with pytest.raises(AssertionError):
        DEFAULT_PARAMS.copy().set("turn_step", 30, 10, 20)   # default outside [min, max], params.py:66-67


def test_env_registry_covers_the_configured_tasks():
    from gym_miniworld_amd.batch import ENV_SPECS
    for k in ("MiniWorld-Hallway-v0", "MiniWorld-OneRoom-v0", "MiniWorld-FourRooms-v0", "MiniWorld-Maze-v0",
              "MiniWorld-MazeS2-v0", "MiniWorld-MazeS3-v0", "MiniWorld-OneRoomS6-v0",
              # round 3: the tasks with mesh entities / frames
              "MiniWorld-PickupObjs-v0", "MiniWorld-RoomObjs-v0", "MiniWorld-CollectHealth-v0", "MiniWorld-Sign-v0",
              "MiniWorld-Sidewalk-v0", "MiniWorld-WallGap-v0", "MiniWorld-ThreeRooms-v0"):
        assert k in ENV_SPECS


def test_bench_action_stream_matches_oracle(oracle_mod):
    import torch
    import bench
    a = bench.make_actions(7, 5, 11, torch.device("cpu")).numpy()
    for t in range(7):
        for i in range(11):
            assert a[t, i] == oracle_mod.action_stream(bench.ACTION_SEED, t, 5 + i)
    big = bench.make_actions(64, 0, 4096, torch.device("cpu")).numpy()
    frac = np.bincount(big.ravel(), minlength=3) / big.size
    assert np.all(np.abs(frac - 1 / 3) < 0.01)


def test_shard_range_partitions_exactly():
    from gym_miniworld_amd.distributed import shard_range
    for total, world in ((65536, 8), (10, 3), (7, 8), (4096, 1)):
        spans = [shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == total
        for (f0, c0), (f1, _) in zip(spans, spans[1:]):
            assert f0 + c0 == f1


def test_oracle_is_marked_as_test_infrastructure():
    for f in ("mw_oracle.h", "mw_oracle.c", "oracle.py"):
        assert "TEST INFRASTRUCTURE ONLY" in open(os.path.join(ROOT, "oracle", f)).read()


def test_native_seed_hash_matches_hashlib_and_fixture(built, oracle_mod):
    """The library's own SHA-512 based seed -> MT19937 key (mwb_api.hip) against hashlib (oracle side) and
    the committed seed_keys.json, without a GPU."""
    import json
    L = built.load()
    keys = json.load(open(os.path.join(ROOT, "tests", "golden", "seed_keys.json")))
    seeds = [int(s) for s in keys] + [5, 99, 123456789, 2 ** 64 - 1, 18446744073709551557]
    for s in seeds:
        k = (ctypes.c_uint32 * 2)()
        n = L.mwb_seed_key(ctypes.c_uint64(s), k)
        assert list(k)[:n] == oracle_mod.seed_to_mt_key(s), s
        if str(s) in keys:
            assert list(k)[:n] == keys[str(s)]
