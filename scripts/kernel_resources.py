"""Register / scratch / LDS table of every kernel in csrc/mwb_kernels.hip, from hipcc's
-Rpass-analysis=kernel-resource-usage remarks (no GPU needed: hipcc cross-compiles gfx950).

    python scripts/kernel_resources.py [--json out.json] [extra hipcc flags ...]

Also used by tests/test_kernel_budget.py, which asserts the render kernels' VGPR and scratch budgets."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gym_miniworld_amd", "csrc", "mwb_kernels.hip")
FIELDS = {"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch",
          "Occupancy [waves/SIMD]": "occupancy", "LDS Size [bytes/block]": "lds", "SGPRs Spill": "sgpr_spill",
          "VGPRs Spill": "vgpr_spill"}


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"] + names, capture_output=True, text=True, check=True).stdout
        return out.strip().split("\n")
    except Exception:
        return names


def remarks_of_the_build():
    """the remarks gym_miniworld_amd/build.py kept from the compile that produced libmwbatch.so, if the sources (and flags) are
    still the ones it compiled; None otherwise"""
    sys.path.insert(0, ROOT)
    try:
        from gym_miniworld_amd import build as B
        with open(B.REMARKS) as fh:
            head, _, body = fh.read().partition("\n")
        if head.strip() == "digest " + B.source_digest():
            return body
    except Exception:
        pass
    return None


def kernel_resources(extra_flags=(), src=SRC):
    """-> {demangled kernel name: {vgprs, agprs, sgprs, scratch, occupancy, lds, ...}}"""
    err = remarks_of_the_build() if not extra_flags and src == SRC else None
    if err is None:
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        with tempfile.TemporaryDirectory() as td:
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-c", src,
                   "-o", os.path.join(td, "k.o"), "-Rpass-analysis=kernel-resource-usage"] + list(extra_flags)
            err = subprocess.run(cmd, capture_output=True, text=True, check=True).stderr
    res, cur = {}, None
    for line in err.split("\n"):
        m = re.search(r"remark: [^:]*:\d+:\d+: (.*?): (.*?) \[-Rpass-analysis", line) or \
            re.search(r"remark: (.*?): (.*?) \[-Rpass-analysis", line)
        if not m:
            m2 = re.search(r":\d+:\d+: (Function Name|Name): (\S+)", line)
            if m2:
                cur = m2.group(2)
                res[cur] = {}
            continue
        key, val = m.group(1).strip(), m.group(2).strip()
        if key in ("Function Name", "Name"):
            cur = val
            res[cur] = {}
        elif cur is not None and key in FIELDS:
            res[cur][FIELDS[key]] = int(val)
    names = list(res)
    return {d: res[n] for n, d in zip(names, demangle(names))}


if __name__ == "__main__":
    args = sys.argv[1:]
    out_json = None
    if "--json" in args:
        i = args.index("--json")
        out_json = args[i + 1]
        del args[i:i + 2]
    table = kernel_resources(args)
    print("%-64s %5s %5s %5s %8s %4s %7s" % ("kernel", "VGPR", "AGPR", "SGPR", "scratch", "occ", "LDS"))
    for k, v in sorted(table.items()):
        short = re.sub(r"\(MwbDev.*", "", k)
        print("%-64s %5d %5d %5d %8d %4d %7d" % (short[:64], v.get("vgprs", -1), v.get("agprs", -1), v.get("sgprs", -1),
                                                  v.get("scratch", -1), v.get("occupancy", -1), v.get("lds", -1)))
    if out_json:
        with open(out_json, "w") as f:
            json.dump(table, f, indent=1, sort_keys=True)
