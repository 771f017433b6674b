"""ctypes front-end of the CPU oracle (oracle/dgs_oracle.cpp).

TEST INFRASTRUCTURE ONLY.  Imported by tests/, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg
of ``bench.py`` -- never by the product package ``deblurgs_amd``.

One call = one subframe, with the reference's semantics
(/root/reference/submodules/diff-gaussian-rasterization/cuda_rasterizer/rasterizer_impl.cu:198-463).
Every intermediate of the reference's Geometry/Binning/Image state is returned so the HIP kernels can be
checked stage by stage.  Parity status: see the header of dgs_oracle.cpp ("parity unpinned" for the L0
kernels; pinned pieces listed there).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# DGS_ORACLE_LIB: another build of the single-thread oracle -- tests/test_sanitize.py points it at the ASan + UBSan build
_LIB_PATH = os.environ.get("DGS_ORACLE_LIB") or os.path.join(_HERE, "libdgs_oracle.so")
_LIB_OMP_PATH = os.path.join(_HERE, "libdgs_oracle_omp.so")   # bench.py cpu_baseline, full-size parity tests
_LIB_FMA_PATH = os.path.join(_HERE, "libdgs_oracle_fma.so")   # the OpenMP build with FMA contraction (as nvcc --fmad=true)
_lib = None
_lib_omp = None
_lib_fma = None
_use_omp = False
_use_fma = False

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)
_u32p = ctypes.POINTER(ctypes.c_uint32)
_u64p = ctypes.POINTER(ctypes.c_uint64)
_u8p = ctypes.POINTER(ctypes.c_uint8)


def build(force=False):
    """Compile oracle/libdgs_oracle.so with the committed Makefile (g++, -ffp-contract=off)."""
    src = os.path.join(_HERE, "dgs_oracle.cpp")
    for path in (_LIB_PATH, _LIB_OMP_PATH, _LIB_FMA_PATH):
        if force or not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "-s", os.path.basename(path)])
    return _LIB_PATH


def _load(path):
    L = ctypes.CDLL(path)
    L.dgs_oracle_preprocess.restype = ctypes.c_int
    L.dgs_oracle_threads.restype = ctypes.c_int
    L.dgs_oracle_higher_msb.restype = ctypes.c_uint32
    L.dgs_oracle_higher_msb.argtypes = [ctypes.c_uint32]
    return L


# OpenMP threads per oracle call.  The oracle's loops stop scaling -- and then lose -- beyond a few dozen threads: on the GPU
# box's host (2 x EPYC 9575F, 256 hardware threads) eight metric subframes take, forward / masks / backward,
# 15.5 / 10.2 / 14.3 s with 256 threads and 5.1 / 5.9 / 6.9 s with 32; running subframes SIDE BY SIDE on host threads of
# their own (8 x 32, 16 x 16, active or passive waiting) is slower than one after the other on 32 (8.4 / 8.9 / 9.0 s):
# profiles/oracle_threads_r06.txt.  DGS_ORACLE_THREADS overrides.
MAX_THREADS_PER_CALL = int(os.environ.get("DGS_ORACLE_THREADS", "32"))


def threads_per_call():
    return max(1, min(MAX_THREADS_PER_CALL, os.cpu_count() or 1))


def set_threads(n=None):
    """OpenMP threads for the oracle calls of the CALLING host thread (the setting is per thread)."""
    L = lib()
    L.dgs_oracle_set_threads.restype = ctypes.c_int
    return int(L.dgs_oracle_set_threads(int(n if n is not None else threads_per_call())))


def map_subframes(fn, items):
    """fn over the items (one oracle call each), one after the other, with the thread count that serves a call best."""
    if _use_omp:
        set_threads()
    return [fn(x) for x in items]


def use_openmp(on=True):
    """Route forward()/backward() to the OpenMP build (cpu_baseline timing only).  Returns the thread count."""
    global _use_omp
    _use_omp = bool(on)
    if _use_omp:
        return set_threads()
    return lib().dgs_oracle_threads()


def use_fma(on=True):
    """Route backward() / chain_backward() to the FMA-contracted OpenMP build (the arithmetic nvcc's default --fmad=true
    gives the reference).  Forward states must come from the uncontracted build: tile lists are compared bit for bit."""
    global _use_fma
    assert _use_omp or not on, "the FMA build is an OpenMP (deterministic, double-accumulating) build"
    _use_fma = bool(on)


def set_accum_f32(on):
    """OpenMP build only: accumulate the compositing backward in emulated fp32 (same deterministic order as the default
    double accumulation).  |double - f32| is the rounding-noise floor the parity tests put next to their 1e-4 bar."""
    assert _use_omp, "the fp32-emulation mode belongs to the OpenMP (deterministic, double-accumulating) build"
    lib().dgs_oracle_set_accum_f32(int(bool(on)))


def lib():
    global _lib, _lib_omp, _lib_fma
    if not os.path.exists(_LIB_PATH) or not os.path.exists(_LIB_OMP_PATH) or not os.path.exists(_LIB_FMA_PATH):
        build()
    if _use_omp and _use_fma:
        if _lib_fma is None:
            _lib_fma = _load(_LIB_FMA_PATH)
        return _lib_fma
    if _use_omp:
        if _lib_omp is None:
            _lib_omp = _load(_LIB_OMP_PATH)
        return _lib_omp
    if _lib is None:
        _lib = _load(_LIB_PATH)
    return _lib


def _p(a, ty):
    if a is None:
        return ctypes.cast(None, ty)
    assert a.flags["C_CONTIGUOUS"], "oracle arrays must be contiguous"
    return a.ctypes.data_as(ty)


def _f(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def higher_msb(n):
    return int(lib().dgs_oracle_higher_msb(int(n)))


def forward(means3D, opacities, viewmatrix, projmatrix, campos, bg, W, H, tanfovx, tanfovy, z_far=100.0,
            sh=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None, sh_degree=0,
            scale_modifier=1.0, use_sigmoid=False, render=True):
    """Reference forward for one subframe.  Returns a dict with outputs and all saved state."""
    L = lib()
    means3D = _f(means3D).reshape(-1, 3)
    P = means3D.shape[0]
    sh = _f(sh)
    M = 0 if sh is None else sh.shape[1]
    colors_precomp, scales, rotations, cov3D_precomp = map(_f, (colors_precomp, scales, rotations, cov3D_precomp))
    opacities = _f(opacities).reshape(-1)
    viewmatrix, projmatrix = _f(viewmatrix).reshape(16), _f(projmatrix).reshape(16)
    campos, bg = _f(campos).reshape(3), _f(bg).reshape(3)
    assert (sh is None) != (colors_precomp is None)
    assert (scales is None) == (rotations is None) and (scales is None) != (cov3D_precomp is None)
    N = W * H
    T = ((W + 15) // 16) * ((H + 15) // 16)
    st = dict(
        P=P, M=M, D=int(sh_degree), W=W, H=H,
        radii=np.zeros(P, np.int32), depths=np.zeros(P, np.float32), pre_sigmoid=np.zeros((P, 3), np.float32),
        means2D=np.zeros((P, 2), np.float32), cov3D=np.zeros((P, 6), np.float32),
        conic_opacity=np.zeros((P, 4), np.float32), rgb=np.zeros((P, 3), np.float32),
        tiles_touched=np.zeros(P, np.uint32), point_offsets=np.zeros(P, np.uint32),
    )
    R = L.dgs_oracle_preprocess(
        P, int(sh_degree), M, W, H, _p(means3D, _f32p), _p(sh, _f32p), _p(colors_precomp, _f32p),
        _p(opacities, _f32p), _p(scales, _f32p), ctypes.c_float(scale_modifier), _p(rotations, _f32p),
        _p(cov3D_precomp, _f32p), _p(viewmatrix, _f32p), _p(projmatrix, _f32p), _p(campos, _f32p),
        ctypes.c_float(tanfovx), ctypes.c_float(tanfovy), int(bool(use_sigmoid)),
        _p(st["radii"], _i32p), _p(st["depths"], _f32p), _p(st["pre_sigmoid"], _f32p), _p(st["means2D"], _f32p),
        _p(st["cov3D"], _f32p), _p(st["conic_opacity"], _f32p), _p(st["rgb"], _f32p),
        _p(st["tiles_touched"], _u32p), _p(st["point_offsets"], _u32p))
    st["num_rendered"] = int(R)
    st["keys_unsorted"] = np.zeros(R, np.uint64)
    st["vals_unsorted"] = np.zeros(R, np.uint32)
    st["keys"] = np.zeros(R, np.uint64)
    st["point_list"] = np.zeros(R, np.uint32)
    st["ranges"] = np.zeros((T, 2), np.uint32)
    L.dgs_oracle_bin(P, W, H, R, _p(st["radii"], _i32p), _p(st["means2D"], _f32p), _p(st["depths"], _f32p),
                     _p(st["point_offsets"], _u32p), _p(st["keys_unsorted"], _u64p), _p(st["vals_unsorted"], _u32p),
                     _p(st["keys"], _u64p), _p(st["point_list"], _u32p), _p(st["ranges"], _u32p))
    st["inputs"] = dict(means3D=means3D, sh=sh, colors_precomp=colors_precomp, opacities=opacities, scales=scales,
                        rotations=rotations, cov3D_precomp=cov3D_precomp, viewmatrix=viewmatrix,
                        projmatrix=projmatrix, campos=campos, bg=bg, tanfovx=float(tanfovx), tanfovy=float(tanfovy),
                        z_far=float(z_far), scale_modifier=float(scale_modifier), use_sigmoid=bool(use_sigmoid))
    if render:
        st["final_T"] = np.zeros(N, np.float32)
        st["n_contrib"] = np.zeros(N, np.uint32)
        st["color"] = np.zeros((3, H, W), np.float32)
        st["depth"] = np.zeros((1, H, W), np.float32)
        st["contrib_checksum"] = np.zeros(N, np.uint32)     # which pairs contributed to each pixel (see dgs_oracle_render)
        feats = colors_precomp if colors_precomp is not None else st["rgb"]
        L.dgs_oracle_render(W, H, _p(st["ranges"], _u32p), _p(st["point_list"], _u32p), _p(st["means2D"], _f32p),
                            _p(feats, _f32p), _p(st["depths"], _f32p), _p(st["conic_opacity"], _f32p), _p(bg, _f32p),
                            ctypes.c_float(z_far), _p(st["final_T"], _f32p), _p(st["n_contrib"], _u32p),
                            _p(st["color"], _f32p), _p(st["depth"], _f32p), _p(st["contrib_checksum"], _u32p))
    return st


def backward(st, dL_dcolor, dL_ddepth=None):
    """Reference backward for one subframe given the dict returned by :func:`forward`."""
    L = lib()
    inp = st["inputs"]
    P, M, D, W, H = st["P"], st["M"], st["D"], st["W"], st["H"]
    dL_dcolor = _f(dL_dcolor).reshape(3, H, W)
    dL_ddepth = np.zeros((1, H, W), np.float32) if dL_ddepth is None else _f(dL_ddepth).reshape(1, H, W)
    g = dict(
        dL_dmeans2D=np.zeros((P, 3), np.float32), dL_dconic=np.zeros((P, 4), np.float32),
        dL_dopacity=np.zeros((P, 1), np.float32), dL_dcolors=np.zeros((P, 3), np.float32),
        dL_ddepths=np.zeros((P, 1), np.float32), dL_dmeans3D=np.zeros((P, 3), np.float32),
        dL_dcov3D=np.zeros((P, 6), np.float32), dL_dsh=np.zeros((P, M, 3), np.float32),
        dL_dscales=np.zeros((P, 3), np.float32), dL_drotations=np.zeros((P, 4), np.float32),
        dL_dviewmatrix=np.zeros(16, np.float32), dL_dprojmatrix=np.zeros(16, np.float32),
    )
    feats = inp["colors_precomp"] if inp["colors_precomp"] is not None else st["rgb"]
    L.dgs_oracle_render_backward(
        W, H, _p(st["ranges"], _u32p), _p(st["point_list"], _u32p), _p(inp["bg"], _f32p), _p(st["means2D"], _f32p),
        _p(st["conic_opacity"], _f32p), _p(feats, _f32p), _p(st["depths"], _f32p), _p(st["final_T"], _f32p),
        _p(st["n_contrib"], _u32p), _p(dL_dcolor, _f32p), _p(dL_ddepth, _f32p), ctypes.c_float(inp["z_far"]),
        _p(g["dL_dmeans2D"], _f32p), _p(g["dL_dconic"], _f32p), _p(g["dL_dopacity"], _f32p),
        _p(g["dL_dcolors"], _f32p), _p(g["dL_ddepths"], _f32p))
    cov3D = inp["cov3D_precomp"] if inp["cov3D_precomp"] is not None else st["cov3D"]
    L.dgs_oracle_preprocess_backward(
        P, D, M, W, H, _p(inp["means3D"], _f32p), _p(st["radii"], _i32p), _p(inp["sh"], _f32p),
        _p(st["pre_sigmoid"], _f32p), _p(inp["scales"], _f32p), _p(inp["rotations"], _f32p),
        ctypes.c_float(inp["scale_modifier"]), _p(cov3D, _f32p), _p(inp["viewmatrix"], _f32p),
        _p(inp["projmatrix"], _f32p), _p(inp["campos"], _f32p), ctypes.c_float(inp["tanfovx"]),
        ctypes.c_float(inp["tanfovy"]), int(inp["use_sigmoid"]), _p(g["dL_dmeans2D"], _f32p),
        _p(g["dL_dconic"], _f32p), _p(g["dL_dmeans3D"], _f32p), _p(g["dL_dcolors"], _f32p),
        _p(g["dL_ddepths"], _f32p), _p(g["dL_dcov3D"], _f32p), _p(g["dL_dsh"], _f32p), _p(g["dL_dscales"], _f32p),
        _p(g["dL_drotations"], _f32p), _p(g["dL_dviewmatrix"], _f32p), _p(g["dL_dprojmatrix"], _f32p))
    g["dL_dviewmatrix"] = g["dL_dviewmatrix"].reshape(4, 4)
    g["dL_dprojmatrix"] = g["dL_dprojmatrix"].reshape(4, 4)
    return g


def chain_backward(st, dL_dmeans2D, dL_dconic, dL_dcolors, dL_ddepths=None):
    """Only the per-Gaussian half of the reference backward -- BACKWARD::preprocess = computeCov2DCUDA + preprocessCUDA
    (backward.cu:145-460) -- on GIVEN outputs of the compositing backward for one subframe: dL_dmeans2D [P,3] (NDC-scaled),
    dL_dconic [P,4] (.x, .y, .w used), dL_dcolors [P,3], dL_ddepths [P,1].  Lets a test feed the chain with another
    implementation's compositing results and so check that implementation's chain in isolation."""
    L = lib()
    inp = st["inputs"]
    P, M, D, W, H = st["P"], st["M"], st["D"], st["W"], st["H"]
    g = dict(
        dL_dmeans2D=_f(dL_dmeans2D).reshape(P, 3).copy(), dL_dconic=_f(dL_dconic).reshape(P, 4).copy(),
        dL_dcolors=_f(dL_dcolors).reshape(P, 3).copy(),
        dL_ddepths=np.zeros((P, 1), np.float32) if dL_ddepths is None else _f(dL_ddepths).reshape(P, 1).copy(),
        dL_dmeans3D=np.zeros((P, 3), np.float32), dL_dcov3D=np.zeros((P, 6), np.float32),
        dL_dsh=np.zeros((P, M, 3), np.float32), dL_dscales=np.zeros((P, 3), np.float32),
        dL_drotations=np.zeros((P, 4), np.float32), dL_dviewmatrix=np.zeros(16, np.float32),
        dL_dprojmatrix=np.zeros(16, np.float32))
    cov3D = inp["cov3D_precomp"] if inp["cov3D_precomp"] is not None else st["cov3D"]
    L.dgs_oracle_preprocess_backward(
        P, D, M, W, H, _p(inp["means3D"], _f32p), _p(st["radii"], _i32p), _p(inp["sh"], _f32p),
        _p(st["pre_sigmoid"], _f32p), _p(inp["scales"], _f32p), _p(inp["rotations"], _f32p),
        ctypes.c_float(inp["scale_modifier"]), _p(cov3D, _f32p), _p(inp["viewmatrix"], _f32p),
        _p(inp["projmatrix"], _f32p), _p(inp["campos"], _f32p), ctypes.c_float(inp["tanfovx"]),
        ctypes.c_float(inp["tanfovy"]), int(inp["use_sigmoid"]), _p(g["dL_dmeans2D"], _f32p),
        _p(g["dL_dconic"], _f32p), _p(g["dL_dmeans3D"], _f32p), _p(g["dL_dcolors"], _f32p),
        _p(g["dL_ddepths"], _f32p), _p(g["dL_dcov3D"], _f32p), _p(g["dL_dsh"], _f32p), _p(g["dL_dscales"], _f32p),
        _p(g["dL_drotations"], _f32p), _p(g["dL_dviewmatrix"], _f32p), _p(g["dL_dprojmatrix"], _f32p))
    g["dL_dviewmatrix"] = g["dL_dviewmatrix"].reshape(4, 4)
    g["dL_dprojmatrix"] = g["dL_dprojmatrix"].reshape(4, 4)
    return g


def unstable(st, alpha_tol=5e-7, power_tol=1e-4, T_tol=1e-8):
    """[H,W] bool: pixels where some pair of the oracle's own traversal sits within a margin of one of the reference's
    three thresholds (dgs_oracle_unstable; same rule as tests/helpers.unstable_pixels, usable at BASELINE sizes)."""
    W, H = st["W"], st["H"]
    flag = np.zeros(W * H, np.uint8)
    lib().dgs_oracle_unstable(W, H, _p(st["ranges"], _u32p), _p(st["point_list"], _u32p), _p(st["means2D"], _f32p),
                              _p(st["conic_opacity"], _f32p), ctypes.c_float(alpha_tol), ctypes.c_float(power_tol),
                              ctypes.c_float(T_tol), _p(flag, _u8p))
    return flag.reshape(H, W).astype(bool)


def mark_visible(means3D, viewmatrix):
    means3D = _f(means3D).reshape(-1, 3)
    out = np.zeros(means3D.shape[0], np.uint8)
    lib().dgs_oracle_mark_visible(means3D.shape[0], _p(means3D, _f32p), _p(_f(viewmatrix).reshape(16), _f32p),
                                  _p(out, _u8p))
    return out.astype(bool)
