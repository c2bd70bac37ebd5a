#!/usr/bin/env python3
"""bench.py - env-steps/s of the batched MiniWorld hot path on N MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (MiniWorldEnv.step + auto-reset + render_obs for every env of
the rank's shard) on synthetic uniformly random actions that are resident in HBM before the timed
region.  For N > 1 every rank owns a contiguous env range (weak scaling: envs per GPU fixed); per step
rank 0 (the "learner") scatters the actions, and every rank's observation shard + reward / done / info
pack is exchanged in one round (RCCL), overlapped with the next step and consumed one round behind
inside the timed loop.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# name -> (env id, envs per GPU, depth, domain_rand, SURVEY.md 8(d) algorithmic bytes per env-step[, actions drawn from range(n)])
WORKLOADS = {
    "maze8192": ("MiniWorld-Maze-v0", 8192, False, False, 25600),             # north_star target / configs[4] per-GPU shape
    "oneroom4096": ("MiniWorld-OneRoom-v0", 4096, False, False, 14700),       # configs[1]
    "maze8192_depth": ("MiniWorld-Maze-v0", 8192, True, False, 44800),        # configs[2]
    "fourrooms16384_dr": ("MiniWorld-FourRooms-v0", 16384, False, True, 16100),   # configs[3]
    # the two-box T-maze with info['feature'] the fork's trainer uses;
    # 14 400 B obs + ~0.2 KB state + 2 rooms x 96 B + 416 B frame constants
    "tmaze_features8192": ("MiniWorld-TMazeTwoBoxDynamicFeatures100K-v0", 8192, False, False, 15200),
    # the sim-to-real rink with box pushing (no ceiling, own params, always randomised, Discrete(4) incl. move_back
    # is available but the shared action stream stays on {0, 1, 2}); 14 400 B obs + ~0.8 KB state incl. the RNG words
    "sim2real_push8192": ("MiniWorld-SimToRealPush-v0", 8192, False, True, 15200),
    # six boxes + the carry actions; 14 400 B obs + ~0.6 KB state (six box poses / sizes / colours)
    # + 96 B room + 960 B frame constants.  The shared action stream stays on {0, 1, 2} (nothing gets picked up).
    "putnext8192": ("MiniWorld-PutNext-v0", 8192, False, False, 16000),
    # polygon rooms; 14 400 B obs + ~0.2 KB state + 6 rooms x 208 B + 288 B frame constants
    "ymaze8192": ("MiniWorld-YMaze-v0", 8192, False, False, 16200),
    # SURVEY.md 8f.2 (round 3): mesh entities.  14 400 B obs + 2 864 B frame constants (20 entity slots x 34 words + 36) + entity
    # state read / written (~0.1 KB per entity) + 96 B per room.  Actions over the task's whole action space, so that objects
    # ARE picked up (PickupObjs: Discrete(5); CollectHealth: the base class' Discrete(8))
    "pickupobjs8192": ("MiniWorld-PickupObjs-v0", 8192, False, False, 17900, 5),
    "collecthealth8192": ("MiniWorld-CollectHealth-v0", 8192, False, False, 19200, 8),
    "sidewalk8192": ("MiniWorld-Sidewalk-v0", 8192, False, True, 18300, 3),
}
ORACLE_TASK = {"MiniWorld-Maze-v0": ("Maze", None), "MiniWorld-OneRoom-v0": ("OneRoom", None),
               "MiniWorld-FourRooms-v0": ("FourRooms", None), "MiniWorld-Hallway-v0": ("Hallway", None),
               "MiniWorld-TMazeTwoBoxDynamicFeatures100K-v0": ("TMazeTwoBox", [1, 0, 0, 100000]),
               "MiniWorld-SimToRealPush-v0": ("SimToRealPush", None), "MiniWorld-PutNext-v0": ("PutNext", None),
               "MiniWorld-YMaze-v0": ("YMaze", [0, 0, 0, 0]), "MiniWorld-PickupObjs-v0": ("PickupObjs", [12, 5, 0, 0]),
               "MiniWorld-CollectHealth-v0": ("CollectHealth", [16, 0, 0, 0]), "MiniWorld-Sidewalk-v0": ("Sidewalk", None)}
HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
ACTION_SEED = 12345


def splitmix64(x):
    """torch int64 splitmix64 (logical shifts emulated); matches oracle.action_stream / mwo_bench_loop."""
    import torch
    def lsr(v, n):
        return (v >> n) & ((1 << (64 - n)) - 1)
    x = x + torch.tensor(-7046029254386353131, dtype=torch.int64, device=x.device)   # 0x9E3779B97F4A7C15
    x = (x ^ lsr(x, 30)) * torch.tensor(-4658895280553007687, dtype=torch.int64, device=x.device)   # 0xBF58476D1CE4E5B9
    x = (x ^ lsr(x, 27)) * torch.tensor(-7723592293110705685, dtype=torch.int64, device=x.device)   # 0x94D049BB133111EB
    return x ^ lsr(x, 31)


def make_actions(steps, first_env, n_envs, device, n_actions=3):
    """[steps, n_envs] int32 actions in range(n_actions) (default {0,1,2}): counter-based in (step, global env index)."""
    import torch
    t = torch.arange(steps, dtype=torch.int64, device=device)[:, None]
    e = torch.arange(first_env, first_env + n_envs, dtype=torch.int64, device=device)[None, :]
    inner = splitmix64(t * 0x100000001B3 + e)
    v = splitmix64(inner ^ ACTION_SEED)
    return (((v >> 33) & ((1 << 31) - 1)) % n_actions).to(torch.int32).contiguous()


def host_cores(cap=64):
    """CPU cores this process may actually use: the affinity mask, clipped by the cgroup CPU quota
    (a GPU box hands each job a share of the host, e.g. 16 of 256 cores) and by `cap`."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as fh:
                txt = fh.read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
                        n = min(n, max(1, int(q / int(fh.read()) + 0.5)))
            break
        except Exception:
            continue
    return max(1, min(n, cap))


def cpu_baseline(env_id, depth, dr, budget_s=10.0, n_actions=3):
    """Oracle (CPU restatement, scalar C) timed on this host on a bounded sample of the same workload:
    one env per thread on every host core (the reference's own parallelism is one process per env,
    vec_env/subproc_vec_env.py:36-56), plus the single-thread rate."""
    import threading
    from oracle import oracle as O
    task, args = ORACLE_TASK[env_id]

    params = None
    if task.startswith("SimToReal"):   # the classes' own parameter table (simtorealpush.py:8-18)
        from gym_miniworld_amd.params import sim_to_real_params
        params = sim_to_real_params(push=task.endswith("Push")).to_table()

    def make(i):
        e = O.OracleEnv(task, seed=1 + i, domain_rand=dr, task_args=args, params=params)
        e.reset(render=False)
        return e
    env0 = make(0)
    dt = env0.bench_loop(200, ACTION_SEED, 0, want_depth=depth, n_actions=n_actions)   # calibrate
    n1 = max(200, min(6000, int(200 * 4.0 / max(dt, 1e-6))))
    dt1 = env0.bench_loop(n1, ACTION_SEED, 0, want_depth=depth, n_actions=n_actions)
    single = n1 / dt1
    cores = host_cores()
    envs = [make(i) for i in range(cores)]
    n = max(100, min(6000, int(single * budget_s)))
    times = [0.0] * cores

    def run(i):   # ctypes releases the GIL inside the C loop
        times[i] = envs[i].bench_loop(n, ACTION_SEED, i, want_depth=depth, n_actions=n_actions)
    th = [threading.Thread(target=run, args=(i,)) for i in range(cores)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    wall = time.perf_counter() - t0
    return {"value": cores * n / wall, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "single_thread_value": single,
            "sample": "%d envs x %d steps (%s, seeds 1.., same counter-based random actions), oracle/mw_oracle.c "
                      "step+auto-reset+render, one env per thread on %d host cores" % (cores, n, env_id, cores)}


def vecenv_rates(env_id, per_gpu, dr, device, K, Wm, n_actions=3):
    """The same workload through the boundary the reference's trainer calls (pytorch-a2c-ppo-acktr/main.py:610:
    `obs, reward, done, infos = envs.step(action, env_mask)`): actions as LongTensor [N,1] on the device, numpy dones,
    CPU float rewards, info dicts.  Two views: the bare VecEnv (uint8 [N,3,80,60] observations: what the Python layer
    itself costs on top of the C ABI) and make_vec_envs' view (float32 4-frame stack [N,12,80,60], envs.py:57-165,
    whose fused stack pass moves 3.4 GB per step at 8192 envs and is HBM-bound)."""
    import torch
    from gym_miniworld_amd.vec_env import MiniWorldVecEnv, make_vec_envs
    out = {}
    for name in ("vecenv_u8", "make_vec_envs_f32_stack4"):
        if name == "vecenv_u8":
            v = MiniWorldVecEnv(env_id, per_gpu, seed=1, device=device.index, domain_rand=dr, to_float=False, feature_info=True)
        else:
            v = make_vec_envs(env_id, 1, per_gpu, device=str(device), domain_rand=dr)
        acts = make_actions(K + Wm, 0, per_gpu, device, n_actions).to(torch.int64).unsqueeze(2)   # [T, N, 1] LongTensor
        v.reset()
        n_done = 0
        for t in range(Wm):
            v.step(acts[t])
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for t in range(Wm, Wm + K):
            _, _, done, infos = v.step(acts[t])
            n_done += int(done.sum())
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        out[name] = {"value": per_gpu * K / dt, "unit": "env-steps/s", "ms_per_step": dt / K * 1e3, "episodes_ended": n_done,
                     "infos": type(infos).__name__}
        v.close()
    return out


def load_traffic(workload):
    """HBM bytes per render launch from the newest committed rocprofv3 --pmc summary that has this
    workload (profiles/r*_pmc_traffic.json, written by scripts/summarize_prof.py), else None."""
    import glob
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            with open(p) as fh:
                v = json.load(fh).get(workload, {}).get("hbm_bytes_per_render_launch")
            if v:
                return v
        except Exception:
            pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="maze8192", choices=sorted(WORKLOADS))
    ap.add_argument("--envs-per-gpu", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-vecenv", action="store_true", help="skip the (untimed-for-`value`) leg through the VecEnv boundary")
    ap.add_argument("--no-gather", action="store_true", help="skip the obs all-gather for N > 1")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend; gloo (+ --no-gather) only to rehearse the multi-rank flow on one GPU")
    ap.add_argument("--force-gather", action="store_true", help="run the staging + exchange path even at N = 1 (self-test)")
    ap.add_argument("--gather", default="auto", choices=["auto", "ring", "direct", "learner"],
                    help="method of the per-step shard exchange (auto: time ring and direct in the warm-up, keep the faster; "
                         "learner: only rank 0 receives the batch - 1/N of the bytes an all-gather moves)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from gym_miniworld_amd.batch import BatchedMiniWorld
    from gym_miniworld_amd.distributed import ShardExchange, dist_env, shard_range

    rank, local_rank, world = dist_env()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    env_id, per_gpu, depth, dr, alg_bytes = WORKLOADS[args.workload][:5]
    n_actions = WORKLOADS[args.workload][5] if len(WORKLOADS[args.workload]) > 5 else 3
    if args.envs_per_gpu:
        per_gpu = args.envs_per_gpu
    total = per_gpu * world
    first, count = shard_range(total, rank, world)
    dev_index = local_rank % max(1, torch.cuda.device_count())   # one rank per GPU; wraps only in gloo rehearsals
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")
            args.no_gather = True

    env = BatchedMiniWorld(env_id, num_envs=count, seed=1, domain_rand=dr, want_depth=depth, device=dev_index,
                           first_env_index=first)
    K, Wm = args.steps, args.warmup
    actions = make_actions(K + Wm, first, count, device, n_actions)
    gather = None
    actions_all = None
    if (world > 1 and not args.no_gather) or args.force_gather:
        gather = ShardExchange(tuple(env.obs.shape), env.obs.dtype, device, world, rank=rank, method=args.gather)
        if rank == 0:   # the learner's action table for ALL envs (identical values to every rank's own `actions`)
            actions_all = make_actions(K + Wm, 0, total, device, n_actions)
    env.reset()
    consumed = torch.zeros((), dtype=torch.float64, device=device)

    def run(t0, t1):
        for t in range(t0, t1):
            if gather is not None:
                a = gather.scatter_actions(actions_all[t] if rank == 0 else None, src=0)   # 4 B/env back to the shards
            else:
                a = actions[t]
            env.step(a)
            if gather is not None:
                gather.push(env.obs, env=env)            # obs + reward / done / ep_steps / feature / goal_pos, one round
                g_obs, g_aux = gather.previous()         # the learner consumes one round behind (overlap with this step)
                consumed.add_(g_obs.view(-1)[::8191].sum(dtype=torch.float64) + g_aux[:, 0].sum())   # device-side read, no host sync

    def fence():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(device)

    run(0, Wm)
    multi = None
    if gather is not None and Wm > 0:
        # self-check, outside the timed region: every rank's shard of the newest round arrived where the method delivers it
        # (per-shard checksums, all-gathered separately), on distinct devices, and what one exchange costs unoverlapped
        gather.latest()
        ok, seen = gather.verify()
        gather.drain()
        props = torch.cuda.get_device_properties(dev_index)
        me = {"rank": rank, "device": dev_index, "uuid": str(getattr(props, "uuid", "")), "name": props.name, "envs": [first, count]}
        ranks = [me]
        if world > 1:
            ranks = [None] * world
            dist.all_gather_object(ranks, me)
        multi = {"ranks_seen": seen, "world_size": dist.get_world_size() if world > 1 else 1, "ranks": ranks,
                 "distinct_devices": len({r["uuid"] or (r["rank"], r["device"]) for r in ranks}),
                 "gather": {"method": gather.method, "verified": bool(ok), "exchange_ms": gather.measure(4) * 1e3,
                            "bytes_per_rank_per_step": int(env.obs[0].numel()) * count + 64 * count}}
        if not ok or seen != world:
            sys.exit("bench.py: the shard exchange did not deliver every rank's shard (%r)" % (multi,))
    fence()
    # HIP events on the launch stream, every 4th step of the timed region whatever its length (seven events per step
    # cost ~2-5 % of a step's time); read back only after the timed region
    env.timing_enable(4 if K >= 4 else True)
    t_start = time.perf_counter()
    run(Wm, Wm + K)
    if gather is not None:
        gather.drain()
    fence()
    elapsed = time.perf_counter() - t_start
    kt = env.timing_read()
    env.timing_enable(False)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        value = total * K / elapsed
        render_ms = kt["render"]
        achieved = alg_bytes * count / (render_ms * 1e-3) / 1e9 if render_ms > 0 else 0.0
        out = {
            "metric": "env-steps/s (80x60 RGB obs)", "value": value, "unit": "env-steps/s", "n_gpus": world,
            "steps": K, "warmup": Wm, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64 sim / f32 render / u8 obs", "data": "synthetic",
            "config": {"workload": args.workload, "env_id": env_id, "envs_per_gpu": per_gpu, "global_envs": total,
                       "obs": "80x60 RGB" + (" + f32 depth" if depth else ""), "domain_rand": bool(dr),
                       "actions": "uniform random over %s, counter-based" % ("{turn_left, turn_right, move_forward}" if n_actions == 3 else "the %d actions of the task" % n_actions),
                       "auto_reset": True, "parallelism": "env-sharded x%d%s" % (world, "" if gather is None else " + per-step exchange(obs u8 + aux f64[8], %s) + action scatter" % gather.method),
                       "gather_tuning_s": None if gather is None else gather.tuned},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": load_traffic(args.workload),
                         "traffic_source": "committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload (profiles/r*_pmc_traffic.json, scripts/profile_workload.sh); not re-measured in this run",
                         "kernel": "render_kernel", "kernel_ms": render_ms,
                         "algorithmic_bytes_per_launch": alg_bytes * count,
                         "note": "nominally HBM-bound path; the practical limiter is VALU issue (DESIGN.md 4: issue-slot accounting from the committed SQ counter passes)"},
            "kernel_ms": {k: kt[k] for k in ("step", "reset", "prep", "render")},
        }
        if multi is not None:
            out["multi_gpu"] = multi
        if world == 1 and not args.no_vecenv:
            env.close()
            try:
                out["vecenv"] = vecenv_rates(env_id, per_gpu, dr, device, K, Wm, n_actions)
                out["vecenv"]["vs_c_abi"] = out["vecenv"]["vecenv_u8"]["value"] / value
            except Exception as ex:
                out["vecenv"] = {"error": repr(ex)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(env_id, depth, dr, n_actions=n_actions)
            except Exception as ex:   # the oracle is only a reported baseline; never fail the GPU number on it
                out["cpu_baseline"] = {"value": None, "unit": "env-steps/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (ex,)}
        print(json.dumps(out))
    env.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
