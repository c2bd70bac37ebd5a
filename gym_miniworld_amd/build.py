"""Builds libmwbatch.so (the HIP kernels + C ABI) in-tree for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the .so then travels to
the GPU box with the repo snapshot.  `python -m gym_miniworld_amd.build [--force]`.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["mwb_api.hip", "mwb_kernels.hip"]
HEADERS = ["mwb_internal.h", "mwb_glibc_trig.h", "mwb_sincos_table.inc", os.path.join("..", "..", "include", "miniworld_batch.h")]
OUT = os.path.join(HERE, "libmwbatch.so")
# hipcc's per-kernel resource remarks of the build that produced OUT, headed by a digest of the sources: tests/test_kernel_budget.py
# reads them instead of compiling the kernels a second time (five minutes) when the digest still matches
REMARKS = os.path.join(HERE, "libmwbatch.resources.txt")
# -ffp-contract=off: world generation / step must reproduce the reference's float64 arithmetic bit
# for bit, and the render spec states every fused multiply-add explicitly
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wall",
         "-Wno-unused-value"]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def source_digest():
    import hashlib
    h = hashlib.sha1()
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    cmd = [hipcc] + FLAGS + ["-Rpass-analysis=kernel-resource-usage", "-o", OUT] + [os.path.join(CSRC, f) for f in SOURCES]
    if verbose:
        print(" ".join(cmd))
    digest = source_digest()
    r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
    remarks = [ln for ln in r.stderr.split("\n") if "-Rpass-analysis" in ln]
    rest = [ln for ln in r.stderr.split("\n") if "-Rpass-analysis" not in ln and ln.strip()]
    if rest:
        sys.stderr.write("\n".join(rest) + "\n")
    if r.returncode != 0:
        raise subprocess.CalledProcessError(r.returncode, cmd)
    with open(REMARKS, "w") as fh:
        fh.write("digest %s\n" % digest)
        fh.write("\n".join(remarks) + "\n")
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print("built", OUT)
