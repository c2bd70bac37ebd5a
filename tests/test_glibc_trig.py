"""The device sin / cos (gym_miniworld_amd/csrc/mwb_glibc_trig.h) must equal what math.sin / math.cos return
in the reference's process - glibc 2.35's __sin_fma / __cos_fma - bit for bit.  Here the same header is built
for the host and compared with this image's libm on more than 10^7 arguments, incl. the walk of an agent heading
(turn_agent, miniworld.py:635-656) and the half angles gen_rot_matrix takes (math.py:16-17).  The GPU leg
(tests/test_gpu_parity.py) then requires bit-equal poses after long rollouts."""
import ctypes
import math
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
DP, LP = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_long)


@pytest.fixture(scope="module")
def trig():
    out = os.path.join(HERE, "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libglibc_trig_host.so")
    srcs = [os.path.join(HERE, "glibc_trig_host.cpp"), os.path.join(ROOT, "gym_miniworld_amd", "csrc", "mwb_glibc_trig.h"),
            os.path.join(ROOT, "gym_miniworld_amd", "csrc", "mwb_sincos_table.inc")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        # -fno-builtin: keep sin() and cos() two libm calls (gcc would merge them into sincos(), whose glibc build is
        # NOT bit-identical to sin / cos); -ffp-contract=off: only the explicit fma() calls fuse
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-mfma", "-ffp-contract=off", "-fno-builtin",
                               "-o", so, srcs[0], "-lm"])
    return ctypes.CDLL(so)


def compare(L, x):
    x = np.ascontiguousarray(x, np.float64)
    out = (ctypes.c_long * 3)()
    L.trig_compare(x.ctypes.data_as(DP), ctypes.c_long(len(x)), out)
    return out[0], out[1], (x[out[2]].hex() if out[2] >= 0 else None)


def test_matches_libm_on_ten_million_arguments(trig):
    rng = np.random.default_rng(2024)
    n = 2_000_000
    sets = {
        "initial headings, uniform(-pi, pi)": rng.uniform(-math.pi, math.pi, 2 * n),
        "accumulated headings": rng.uniform(-700, 700, 2 * n),
        "table path |x| < 0.8555": rng.uniform(-0.8555, 0.8555, n),
        "hp0 path 0.8554 < |x| < 2.4263": rng.uniform(0.8554, 2.4263, n) * rng.choice([-1.0, 1.0], n),
        "log-uniform 1e-9 .. 1.05e8": np.exp(rng.uniform(math.log(1e-9), math.log(1.05e8), n)) * rng.choice([-1.0, 1.0], n),
        "near multiples of pi/2": rng.integers(-3000, 3000, n) * (math.pi / 2) + rng.normal(0, 1e-5, n),
        "a few ulps around multiples of pi/2": rng.integers(-3000, 3000, n) * (math.pi / 2) * (1 + rng.integers(-9, 9, n) * 2.0 ** -52),
        "around the branch thresholds": np.concatenate([t + rng.integers(-500, 500, n // 8) * (t * 2.0 ** -52) for t in
                                                        (0.126, 0.85546875, 2.426265, 2.0 ** -26, 2.0 ** -27, 1.0 / 256, 0.5)]),
        "multiples of 1/128 and their neighbours": (rng.integers(-110, 110, n) / 128.0) * (1 + rng.integers(-3, 3, n) * 2.0 ** -53),
    }
    total = 0
    for name, x in sets.items():
        bs, bc, first = compare(trig, x)
        assert bs == 0 and bc == 0, (name, bs, bc, first)
        total += len(x)
    assert total >= 10_000_000


def test_matches_python_math_module(trig):
    """the reference calls math.cos / math.sin (CPython -> libm `cos` / `sin`): spot-check against them directly"""
    rng = np.random.default_rng(7)
    x = np.concatenate([rng.uniform(-math.pi, math.pi, 100_000), rng.uniform(-500, 500, 100_000)])
    s, c = np.zeros_like(x), np.zeros_like(x)
    trig.trig_eval(x.ctypes.data_as(DP), ctypes.c_long(len(x)), s.ctypes.data_as(DP), c.ctypes.data_as(DP))
    assert s.tolist() == [math.sin(v) for v in x] and c.tolist() == [math.cos(v) for v in x]


def test_heading_walks(trig):
    """dir += turn_step * pi / 180 with the default 15 degrees, with domain-randomised steps (10..20), and the
    45 degrees of the *Fast tasks: sin / cos of the heading and of its half at every step of 200 k-step walks"""
    rng = np.random.default_rng(3)
    for name, steps in (("15", np.full(200_000, 15.0)), ("dr", rng.uniform(10, 20, 200_000)), ("45", np.full(200_000, 45.0))):
        for d0 in rng.uniform(-math.pi, math.pi, 4):
            turn = steps * rng.choice([-1.0, 1.0], len(steps), p=[0.3, 0.7])   # drifts: headings reach thousands of radians
            out = (ctypes.c_long * 1)()
            trig.trig_compare_walk(ctypes.c_double(d0), np.ascontiguousarray(turn).ctypes.data_as(DP), ctypes.c_long(len(turn)), out)
            assert out[0] == 0, (name, d0, out[0])


def test_table_is_the_correctly_rounded_one_but_for_glibcs_own_deviations():
    """the committed table regenerates from exact rational arithmetic (scripts/gen_sincos_table.py)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_sincos_table", os.path.join(ROOT, "scripts", "gen_sincos_table.py"))
    G = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(G)
    t = G.table()
    txt = open(os.path.join(ROOT, "gym_miniworld_amd", "csrc", "mwb_sincos_table.inc")).read()
    vals = [float.fromhex(v) for line in txt.splitlines() if not line.startswith("//") for v in line.strip().rstrip(",").split(", ")]
    assert len(vals) == 440 and [v.hex() for v in vals] == [v.hex() for v in t]
    exact = G.table(exact=True)
    assert sum(a != b for a, b in zip(t, exact)) == len(G.GLIBC_DEVIATIONS) == 18
    assert all(t[4 * k] == math.sin(k / 128) and t[4 * k + 2] == math.cos(k / 128) for k in range(110))
