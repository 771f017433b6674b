"""AddressSanitizer + UndefinedBehaviorSanitizer over the host side of the C ABI and over the CPU oracle (CPU
box only; GPU-side ASan is not available on the pool).

* deblurgs_amd/libdgs_hip_san.so = the whole library with the HOST code of every translation unit instrumented (argument
  checks, blob carving, 64-bit size arithmetic, launch sequencing; `python -m deblurgs_amd.build --sanitize`, `make
  sanitize`).  tests/test_abi.py runs against it in a child python with the sanitizer runtime preloaded, followed by size /
  layout queries at cfg5's sizes and beyond (R = 4e8 duplicates and the 32-bit limits: every offset is 64-bit arithmetic).
* oracle/libdgs_oracle_san.so = the single-thread oracle; tests/test_oracle_golden.py runs against it the same way.

A sanitizer report aborts the child (halt_on_error / -fno-sanitize-recover), so rc == 0 and "passed" is the whole check."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SIZE_QUERIES = r'''
import ctypes, os, sys
sys.path.insert(0, os.environ["DGS_ROOT"])
from deblurgs_amd import _lib
L = _lib.lib()
assert os.path.basename(_lib.LIB_PATH) == "libdgs_hip_san.so"
cases = [(5_000_000, 3840, 2160, 31, 400_000_000),           # cfg5 (380 M surviving duplicates measured)
         (5_000_000, 3840, 2160, 31, (1 << 32) - 1),         # the largest duplicate count the ABI takes
         (33_000_000, 65520, 65520, 128, (1 << 32) - 1),     # K*P just below 2^32, the largest image, DGS_MAX_K
         (1, 1, 1, 1, 0), (0, 16, 16, 1, 0)]
for P, W, H, K, R in cases:
    for wide in (0, 1):
        lay = _lib.layout(P, W, H, K, R, wide_records=wide)
        assert lay.geom_total == L.dgs_geom_state_bytes(P, K)
        assert lay.image_total == L.dgs_image_state_bytes(W, H, K)
        assert lay.binning_total == L.dgs_binning_state_bytes(R, W, H, K)
        offs = [getattr(lay, n) for n, _ in _lib.DgsLayout._fields_ if n not in
                ("sort_bits", "sort_passes", "pack_g_shift", "pack_tile_shift")]
        assert all(o % 256 == 0 for o in offs)
        assert lay.binning_total >= 24 * R and lay.geom_total >= 48 * K * P
        assert lay.pack_tile_shift == 0 or lay.pack_tile_shift + (lay.sort_bits - 32) <= 64
    so, po = _lib.backward_scratch_layout(R, P, K)
    assert so >= 48 * R and po >= so + 64 * K * P and L.dgs_backward_scratch_bytes(R, P, K) > po
    assert L.dgs_sort_tmp_bytes(R) > 0 and L.dgs_scan_tmp_bytes(K * P) > 0 and L.dgs_depth_order_tmp_bytes(K, P) >= 256
    assert L.dgs_knn_tmp_bytes(P) > 0 and L.dgs_densify_tmp_bytes(P) >= 256 and L.dgs_pose_scratch_bytes(K) > 0
# the checks in front of every launch: blobs one byte short at cfg5's sizes are refused before any HIP call
P, W, H, K, R = 5_000_000, 3840, 2160, 31, 400_000_000
p, out, io = _lib.DgsProblem(), _lib.DgsForwardOut(), _lib.DgsBackwardIO()
p.P, p.W, p.H, p.K, p.D, p.M, p.tile_cull = P, W, H, K, 2, 9, 1
for name in ("means3D", "opacities", "shs", "scales", "rotations", "viewmatrix", "projmatrix", "campos", "bg"):
    setattr(p, name, 4096)
host = (ctypes.c_uint32 * 8)()
out.num_rendered_host = ctypes.cast(host, ctypes.c_void_p)
out.radii = out.out_color = 4096
lay = _lib.layout(P, W, H, K, R)
p.geom_state, p.geom_bytes = 4096, lay.geom_total - 1
assert L.dgs_forward_geometry(ctypes.byref(p), ctypes.byref(out), None) == -2 and b"geom_state" in L.dgs_last_error()
p.geom_bytes = lay.geom_total
p.image_state, p.image_bytes = 4096, lay.image_total - 1
assert L.dgs_forward_render(ctypes.byref(p), ctypes.byref(out), R, None) == -2 and b"image_state" in L.dgs_last_error()
p.image_bytes = lay.image_total
p.binning_state, p.binning_bytes = 4096, lay.binning_total - 1
assert L.dgs_forward_render(ctypes.byref(p), ctypes.byref(out), R, None) == -2 and b"binning_state" in L.dgs_last_error()
p.binning_bytes = lay.binning_total
io.num_rendered = R
for name in ("radii", "dL_dout_color", "dL_dmeans3D", "dL_dmeans2D", "dL_dsh", "dL_dcolors", "dL_dopacity", "dL_dscales",
             "dL_drotations", "dL_dcov3D", "dL_dviewmatrix", "dL_dprojmatrix"):
    setattr(io, name, 4096)
io.scratch, io.scratch_bytes = 4096, L.dgs_backward_scratch_bytes(R, P, K) - 1
assert L.dgs_backward(ctypes.byref(p), ctypes.byref(io), None) == -2 and b"scratch" in L.dgs_last_error()
assert L.dgs_backward_geometry(ctypes.byref(p), ctypes.byref(io), 128, 256, None) == -1      # g_begin not a multiple of 256
p.K = 129
assert L.dgs_forward_geometry(ctypes.byref(p), ctypes.byref(out), None) == -1
p.K, p.P = 128, 40_000_000
assert L.dgs_forward_geometry(ctypes.byref(p), ctypes.byref(out), None) == -1 and b"2^32" in L.dgs_last_error()
print("sanitized size queries ok")
'''


def _child_env(preload, **extra):
    env = dict(os.environ, LD_PRELOAD=preload, DGS_ROOT=ROOT, PYTHONPATH=ROOT,
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0:verify_asan_link_order=0",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    env.update(extra)
    return env


def _run(cmd, env, timeout=1500):
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    tail = f"{' '.join(cmd)}\n{r.stdout[-4000:]}\n{r.stderr[-4000:]}"
    assert "AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, tail
    assert r.returncode == 0, tail
    return r.stdout


def test_c_abi_host_side_under_asan_and_ubsan():
    from deblurgs_amd import build
    rt = build.asan_runtime()
    if rt is None:
        pytest.skip("hipcc's ASan runtime is not installed")
    lib = build.build_sanitized()
    env = _child_env(rt, DGS_LIB_PATH=lib)
    out = _run([sys.executable, "-m", "pytest", os.path.join("tests", "test_abi.py"), "-x", "-q", "-p", "no:cacheprovider"], env)
    assert " passed" in out and "failed" not in out, out
    out = _run([sys.executable, "-c", SIZE_QUERIES], env)
    assert "sanitized size queries ok" in out


def test_oracle_under_asan_and_ubsan():
    r = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True)
    rt = r.stdout.strip()
    if not (os.path.isabs(rt) and os.path.exists(rt)):
        pytest.skip("gcc's ASan runtime is not installed")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "sanitize"])
    env = _child_env(rt, DGS_ORACLE_LIB=os.path.join(ROOT, "oracle", "libdgs_oracle_san.so"))
    out = _run([sys.executable, "-m", "pytest", os.path.join("tests", "test_oracle_golden.py"), "-x", "-q",
                "-p", "no:cacheprovider"], env)
    assert " passed" in out and "failed" not in out, out
