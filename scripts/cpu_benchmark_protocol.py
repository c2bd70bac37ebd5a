#!/usr/bin/env python3
"""The reference's benchmark.py protocol (benchmark.py:8-47) run on the CPU oracle: load time, mean of
100 resets, >= 5 s of stepping with reset on done.  Variant (a) is exactly benchmark.py (Maze-v0,
constant action 0); variant (b) is BASELINE.json configs[0] (Hallway-v0, uniform random actions).
The reference's own Pyglet/OpenGL path cannot run in this pipeline (no gym / pyglet / GL), so these
are numbers of the C restatement (oracle/mw_oracle.c, 1 thread) - a reported baseline only."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402


def run(task, constant_action, seconds=5.0):
    st = time.time()
    env = O.OracleEnv(task, seed=0)
    env.reset()
    load_ms = 1000 * (time.time() - st)
    st = time.time()
    for _ in range(100):
        env.reset()
    reset_ms = 1000 * (time.time() - st) / 100
    n, dt = 0, 0.0
    while dt < seconds:
        dt += env.bench_loop(500, 12345, 0, want_depth=False, constant_action=constant_action)
        n += 500
    return {"task": task, "action": "constant 0" if constant_action >= 0 else "uniform random",
            "load_ms": round(load_ms, 2), "reset_ms": round(reset_ms, 3), "frame_ms": round(1000 * dt / n, 4),
            "fps": round(n / dt, 1), "threads": 1, "host_cores": os.cpu_count()}


if __name__ == "__main__":
    print(json.dumps([run("Maze", 0), run("Hallway", -1)]))
