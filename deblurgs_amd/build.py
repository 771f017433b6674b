"""Builds deblurgs_amd/libdgs_hip.so from deblurgs_amd/csrc/*.hip with hipcc for gfx950 (in-tree, so the
library travels with the repository snapshot).  `python -m deblurgs_amd.build` or `build()`.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "obj")
LIB = os.path.join(HERE, "libdgs_hip.so")
ARCH = "gfx950"
# per-file extra flags: the forward preprocess must not contract mul+add into FMA (bit-exact tile keys)
SOURCES = {
    "preprocess.hip": ["-ffp-contract=off"],
    # tile_cull: the count and emit instantiations of the per-slot test must round identically
    "binning.hip": ["-ffp-contract=off"],
    # SLP packing into v_pk_*_f32 costs more v_mov shuffles than it saves here (920 vs 715 VALU instructions)
    "composite.hip": ["-fno-slp-vectorize"],
    "geometry_bwd.hip": [],
    "pose.hip": [],
    "knn.hip": [],
    # Adam / densification restate torch elementwise ops one rounding per statement
    "optim.hip": ["-ffp-contract=off"],
    "api.hip": [],
}
# (no float atomics anywhere in the library: every reduction has a fixed order or is an integer sum)
COMMON = ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-fhip-fp32-correctly-rounded-divide-sqrt",
          "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, "dgs_common.h"), os.path.join(HERE, "..", "include", "dgs_hip.h")]
    hipcc = _hipcc()
    jobs = []
    objs = []
    for src, extra in SOURCES.items():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([hipcc] + COMMON + extra + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
