"""Builds deblurgs_amd/libdgs_hip.so from deblurgs_amd/csrc/*.hip with hipcc for gfx950 (in-tree, so the
library travels with the repository snapshot).  `python -m deblurgs_amd.build` or `build()`.
"""
import glob
import hashlib
import json
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


HEADER = os.path.join(HERE, "..", "include", "dgs_hip.h")


def _sha(*chunks):
    h = hashlib.sha256()
    for c in chunks:
        h.update(c if isinstance(c, bytes) else c.encode())
        h.update(b"\0")
    return h.hexdigest()


def _read(path):
    with open(path, "rb") as f:
        return f.read()


def flag_table():
    return json.dumps({"common": COMMON, "sources": SOURCES, "arch": ARCH}, sort_keys=True)


def build_id(csrc=CSRC, header=HEADER):
    """SHA-256 over what the library is built from: every csrc/*.hip and csrc/*.h (by name), include/dgs_hip.h and the
    flag table above.  The build compiles it into the library (dgs_build_id()); _lib.lib() recomputes it from the sources
    it finds next to itself and refuses a binary that does not match (a stale libdgs_hip.so travels to the GPU box like a
    fresh one: *.so is git-ignored, not gpurun-ignored)."""
    parts = []
    for f in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h"))):
        parts += [os.path.basename(f), _read(f)]
    parts += ["dgs_hip.h", _read(header), flag_table()]
    return _sha(*parts)


def _object_stamp(src, extra, headers, bid):
    """What an object file depends on: its source, the shared headers, its flags -- and, for api.hip, the build id it
    carries.  Staleness is decided by content, not by modification time (a checkout or a copy resets mtimes)."""
    return _sha(_read(src), *[_read(h) for h in headers], json.dumps(COMMON + extra), bid if src.endswith("api.hip") else "")


def _stamp_matches(path, want):
    try:
        with open(path) as f:
            return f.read().strip() == want
    except OSError:
        return False


SAN_LIB = os.path.join(HERE, "libdgs_hip_san.so")
SAN_OBJ = os.path.join(CSRC, "obj_san")
# Host-side AddressSanitizer + UndefinedBehaviorSanitizer build of the WHOLE library: the same sources, the
# host code of every translation unit -- argument checks, blob carving, size arithmetic, launch sequencing --
# instrumented, the device code left as it is (-fno-gpu-sanitize; GPU-side ASan needs xnack+ code objects, which the pool
# does not run).  For the CPU box only: tests/test_sanitize.py runs tests/test_abi.py and cfg5-size size queries against
# it under the sanitizer runtime.  Never loaded by the product (DGS_LIB_PATH selects it explicitly).
SAN_FLAGS = ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-gpu-sanitize", "-shared-libsan",
             "-fno-sanitize-recover=undefined"]


def asan_runtime():
    """The shared ASan runtime of hipcc's clang (to LD_PRELOAD into a python that loads libdgs_hip_san.so)."""
    r = subprocess.run([_hipcc(), "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True)
    path = r.stdout.strip()
    if os.path.isabs(path) and os.path.exists(path):
        return path
    import glob
    hits = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    return hits[0] if hits else None


def build_sanitized(force=False, verbose=False):
    os.makedirs(SAN_OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, "dgs_common.h"), os.path.join(HERE, "..", "include", "dgs_hip.h")]
    hipcc = _hipcc()
    common = [f for f in COMMON if f != "-O3"] + SAN_FLAGS
    jobs, objs = [], []
    for src, extra in SOURCES.items():
        s = os.path.join(CSRC, src)
        o = os.path.join(SAN_OBJ, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            define = [f'-DDGS_BUILD_ID="{build_id()}"'] if src == "api.hip" else []
            jobs.append([hipcc] + common + extra + define + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(SAN_LIB, objs):
        run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-fsanitize=address,undefined", "-shared-libsan",
             "-o", SAN_LIB] + objs)
    return SAN_LIB


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, "dgs_common.h"), HEADER]
    hipcc = _hipcc()
    bid = build_id()
    jobs = []
    objs = []
    stamps = []
    for src, extra in SOURCES.items():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(o)
        want = _object_stamp(s, extra, headers, bid)
        if force or not os.path.exists(o) or not _stamp_matches(o + ".stamp", want):
            define = [f'-DDGS_BUILD_ID="{bid}"'] if src == "api.hip" else []
            jobs.append([hipcc] + COMMON + extra + define + ["-c", s, "-o", o])
            stamps.append((o + ".stamp", want))

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    for path, want in stamps:
        with open(path, "w") as f:
            f.write(want + "\n")
    if force or jobs or not os.path.exists(LIB) or not _stamp_matches(LIB + ".stamp", bid):
        run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs)
        with open(LIB + ".stamp", "w") as f:
            f.write(bid + "\n")
    return LIB


if __name__ == "__main__":
    if "--sanitize" in sys.argv:
        print(build_sanitized(force="--force" in sys.argv, verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
