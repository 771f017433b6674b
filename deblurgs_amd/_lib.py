"""ctypes binding of libdgs_hip.so (include/dgs_hip.h).

This is the ONLY way the package reaches device code.  There is no CPU or eager-PyTorch fallback: if the
library is missing or a call fails, a RuntimeError is raised (the CPU oracle under /oracle is test
infrastructure and is never imported from here).
"""
import contextlib
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DGS_LIB_PATH: A/B builds of the same library for kernel experiments (tools only; the default is the in-tree build)
LIB_PATH = os.environ.get("DGS_LIB_PATH") or os.path.join(_HERE, "libdgs_hip.so")

DGS_MAX_K = 128
STAGES = ["preprocess", "scan", "duplicate", "sort", "ranges", "composite_fwd", "composite_bwd", "geometry_bwd",
          "depth_order", "tile_cull", "contrib_reduce"]

_c_f32p = ctypes.c_void_p  # device pointers are passed as integers


class DgsProblem(ctypes.Structure):
    _fields_ = [
        ("P", ctypes.c_int32), ("D", ctypes.c_int32), ("M", ctypes.c_int32), ("W", ctypes.c_int32),
        ("H", ctypes.c_int32), ("K", ctypes.c_int32),
        ("tanfovx", ctypes.c_float), ("tanfovy", ctypes.c_float), ("scale_modifier", ctypes.c_float),
        ("z_near", ctypes.c_float), ("z_far", ctypes.c_float),
        ("use_sigmoid", ctypes.c_int32), ("prefiltered", ctypes.c_int32), ("debug", ctypes.c_int32),
        ("tile_cull", ctypes.c_int32), ("raw_params", ctypes.c_int32), ("scale_lb", ctypes.c_float),
        ("wide_records", ctypes.c_int32), ("forward_only", ctypes.c_int32),
        ("means3D", ctypes.c_void_p), ("shs", ctypes.c_void_p), ("shs_rest", ctypes.c_void_p),
        ("colors_precomp", ctypes.c_void_p),
        ("opacities", ctypes.c_void_p), ("scales", ctypes.c_void_p), ("rotations", ctypes.c_void_p),
        ("cov3D_precomp", ctypes.c_void_p), ("viewmatrix", ctypes.c_void_p), ("projmatrix", ctypes.c_void_p),
        ("campos", ctypes.c_void_p), ("bg", ctypes.c_void_p),
        ("geom_state", ctypes.c_void_p), ("geom_bytes", ctypes.c_size_t),
        ("image_state", ctypes.c_void_p), ("image_bytes", ctypes.c_size_t),
        ("binning_state", ctypes.c_void_p), ("binning_bytes", ctypes.c_size_t),
        ("context", ctypes.c_void_p),
    ]


MAX_BWD_PARTS = 8          # DGS_MAX_BWD_PARTS


class DgsContextOptions(ctypes.Structure):
    _fields_ = [("bwd_overlap", ctypes.c_int32), ("bwd_n_parts", ctypes.c_int32),
                ("bwd_parts", ctypes.c_int32 * (MAX_BWD_PARTS - 1))]


class DgsForwardOut(ctypes.Structure):
    _fields_ = [("out_color", ctypes.c_void_p), ("out_depth", ctypes.c_void_p), ("radii", ctypes.c_void_p),
                ("num_rendered_host", ctypes.c_void_p), ("drop_counter", ctypes.c_void_p),
                ("status_dev", ctypes.c_void_p), ("status_host_indirect", ctypes.c_void_p),
                ("debug_contrib_checksum", ctypes.c_void_p)]


class DgsBackwardIO(ctypes.Structure):
    _fields_ = [
        ("num_rendered", ctypes.c_uint32), ("radii", ctypes.c_void_p), ("dL_dout_color", ctypes.c_void_p),
        ("dL_dout_depth", ctypes.c_void_p), ("scratch", ctypes.c_void_p), ("scratch_bytes", ctypes.c_size_t),
        ("dL_dmeans3D", ctypes.c_void_p), ("dL_dmeans2D", ctypes.c_void_p), ("dL_dsh", ctypes.c_void_p),
        ("dL_dsh_rest", ctypes.c_void_p),
        ("dL_dcolors", ctypes.c_void_p), ("dL_dopacity", ctypes.c_void_p), ("dL_dscales", ctypes.c_void_p),
        ("dL_drotations", ctypes.c_void_p), ("dL_dcov3D", ctypes.c_void_p), ("dL_dviewmatrix", ctypes.c_void_p),
        ("dL_dprojmatrix", ctypes.c_void_p), ("opacity_hinge_scale", ctypes.c_float),
        ("stats_max_radii2D", ctypes.c_void_p), ("stats_grad_accum", ctypes.c_void_p), ("stats_denom", ctypes.c_void_p),
        ("stats_K_total", ctypes.c_int32),
    ]


class DgsLayout(ctypes.Structure):
    _fields_ = [(n, ctypes.c_size_t) for n in (
        "geom_rows", "cov3D", "pre_sigmoid", "tiles_touched", "point_offsets", "scan_tmp", "num_rendered",
        "gsort_keys", "gsort_keys_alt", "gsort_vals", "gsort_vals_alt", "tt_sorted", "offs_sorted", "tt_tight", "offs_tight", "gsort_tmp",
        "cull_rec", "cull_cnt", "geom_total", "final_T", "n_contrib", "ranges", "image_total", "keys_sorted", "point_list",
        "keys_unsorted", "vals_unsorted", "sort_tmp", "binning_total")] + [
        ("sort_bits", ctypes.c_int32), ("sort_passes", ctypes.c_int32), ("pack_g_shift", ctypes.c_int32),
        ("pack_tile_shift", ctypes.c_int32)]


class DgsAdamGroup(ctypes.Structure):
    _fields_ = [("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p),
                ("exp_avg_sq", ctypes.c_void_p), ("numel", ctypes.c_uint64), ("lr", ctypes.c_double),
                ("step", ctypes.c_int32)]


class DgsCloudArrays(ctypes.Structure):
    _fields_ = [("param", ctypes.c_void_p * 6), ("exp_avg", ctypes.c_void_p * 6), ("exp_avg_sq", ctypes.c_void_p * 6)]


ADAM_MAX_GROUPS = 16
ABI_VERSION = 14           # DGS_ABI_VERSION of include/dgs_hip.h (tests/test_abi.py keeps the two in step)

# every symbol include/dgs_hip.h declares (tests check that the library exports exactly these)
EXPORTS = {
    "dgs_abi_version": (ctypes.c_int, []),
    "dgs_last_error": (ctypes.c_char_p, []),
    "dgs_build_id": (ctypes.c_char_p, []),
    "dgs_context_create": (ctypes.c_int, [ctypes.POINTER(DgsContextOptions), ctypes.POINTER(ctypes.c_void_p)]),
    "dgs_context_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "dgs_geom_state_bytes": (ctypes.c_size_t, [ctypes.c_int32, ctypes.c_int32]),
    "dgs_image_state_bytes": (ctypes.c_size_t, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    "dgs_image_state_bytes_forward_only": (ctypes.c_size_t, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    "dgs_binning_state_bytes": (ctypes.c_size_t, [ctypes.c_uint64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    "dgs_backward_scratch_bytes": (ctypes.c_size_t, [ctypes.c_uint64, ctypes.c_int32, ctypes.c_int32]),
    "dgs_layout": (ctypes.c_int, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_uint64,
                                  ctypes.c_int32, ctypes.POINTER(DgsLayout)]),
    "dgs_backward_scratch_layout": (ctypes.c_int, [ctypes.c_uint64, ctypes.c_int32, ctypes.c_int32,
                                                   ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]),
    "dgs_forward_geometry": (ctypes.c_int, [ctypes.POINTER(DgsProblem), ctypes.POINTER(DgsForwardOut),
                                            ctypes.c_void_p]),
    "dgs_forward_render": (ctypes.c_int, [ctypes.POINTER(DgsProblem), ctypes.POINTER(DgsForwardOut),
                                          ctypes.c_uint32, ctypes.c_void_p]),
    "dgs_backward": (ctypes.c_int, [ctypes.POINTER(DgsProblem), ctypes.POINTER(DgsBackwardIO), ctypes.c_void_p]),
    "dgs_backward_composite": (ctypes.c_int, [ctypes.POINTER(DgsProblem), ctypes.POINTER(DgsBackwardIO),
                                              ctypes.c_void_p]),
    "dgs_backward_geometry": (ctypes.c_int, [ctypes.POINTER(DgsProblem), ctypes.POINTER(DgsBackwardIO), ctypes.c_int32,
                                             ctypes.c_int32, ctypes.c_void_p]),
    "dgs_backward_pose": (ctypes.c_int, [ctypes.POINTER(DgsProblem), ctypes.POINTER(DgsBackwardIO), ctypes.c_void_p]),
    "dgs_mark_visible": (ctypes.c_int, [ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_void_p, ctypes.c_void_p]),
    "dgs_scan_tmp_bytes": (ctypes.c_size_t, [ctypes.c_uint64]),
    "dgs_exclusive_scan_u32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p,
                                              ctypes.c_void_p, ctypes.c_void_p]),
    "dgs_sort_tmp_bytes": (ctypes.c_size_t, [ctypes.c_uint64]),
    "dgs_depth_order_tmp_bytes": (ctypes.c_size_t, [ctypes.c_int32, ctypes.c_int32]),
    "dgs_depth_order": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_void_p]),
    "dgs_sort_pairs": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_uint64, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p,
                                      ctypes.POINTER(ctypes.c_int32), ctypes.c_void_p]),
    "dgs_blur_loss_grad": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                          ctypes.c_int32, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "dgs_blur_loss_grad_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                              ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                              ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "dgs_rank_ordered_sum": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64,
                                            ctypes.c_int32, ctypes.c_int32, ctypes.c_float, ctypes.c_void_p]),
    "dgs_blur_loss_slice_grad": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                                ctypes.c_int32, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p,
                                                ctypes.c_void_p]),
    "dgs_copy_words": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]),
    "dgs_backward_parts": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_uint64, ctypes.c_int32]),
    "dgs_adam_scalars": (ctypes.c_int, [ctypes.POINTER(DgsAdamGroup), ctypes.c_int32, ctypes.c_double, ctypes.c_double,
                                        ctypes.POINTER(ctypes.c_float)]),
    "dgs_adam_step_dev": (ctypes.c_int, [ctypes.POINTER(DgsAdamGroup), ctypes.c_int32, ctypes.c_double, ctypes.c_double,
                                         ctypes.c_double, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_void_p]),
    "dgs_cloud_activations": (ctypes.c_int, [ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_void_p]),
    "dgs_densify_stats": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_void_p]),
    "dgs_adam_step": (ctypes.c_int, [ctypes.POINTER(DgsAdamGroup), ctypes.c_int32, ctypes.c_double, ctypes.c_double,
                                     ctypes.c_double, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]),
    "dgs_forward": (ctypes.c_int, [ctypes.POINTER(DgsProblem), ctypes.POINTER(DgsForwardOut), ctypes.c_uint32,
                                   ctypes.c_void_p]),
    "dgs_forward_lists": (ctypes.c_int, [ctypes.POINTER(DgsProblem), ctypes.POINTER(DgsForwardOut), ctypes.c_uint32,
                                         ctypes.c_void_p]),
    "dgs_forward_composite": (ctypes.c_int, [ctypes.POINTER(DgsProblem), ctypes.POINTER(DgsForwardOut), ctypes.c_uint32,
                                             ctypes.c_void_p]),
    "dgs_densify_tmp_bytes": (ctypes.c_size_t, [ctypes.c_int32]),
    "dgs_densify_plan": (ctypes.c_int, [ctypes.c_int32] + [ctypes.c_void_p] * 4 + [ctypes.c_float] * 4 +
                         [ctypes.c_int32] + [ctypes.c_void_p] * 6),
    "dgs_densify_apply": (ctypes.c_int, [ctypes.c_int32, ctypes.c_int32] + [ctypes.c_void_p] * 3 +
                          [ctypes.POINTER(DgsCloudArrays), ctypes.POINTER(DgsCloudArrays), ctypes.c_void_p,
                           ctypes.c_float, ctypes.c_int32, ctypes.c_void_p]),
    "dgs_knn_tmp_bytes": (ctypes.c_size_t, [ctypes.c_int32]),
    "dgs_knn_mean_dist2": (ctypes.c_int, [ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_void_p]),
    "dgs_alignment_forward": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                             ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "dgs_alignment_backward": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                              ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "dgs_pose_scratch_bytes": (ctypes.c_size_t, [ctypes.c_int32]),
    "dgs_pose_forward": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int32] * 3 + [ctypes.c_void_p] * 4),
    "dgs_pose_backward": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int32] * 3 + [ctypes.c_void_p] * 7),
    "dgs_profile_enable": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32]),
    "dgs_profile_reset": (ctypes.c_int, [ctypes.c_void_p]),
    "dgs_profile_read": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int32),
                                        ctypes.c_int32]),
}

_lib = None


def lib():
    """Load libdgs_hip.so (fails loudly: there is no fallback path)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m deblurgs_amd.build` (hipcc, gfx950). "
                "deblurgs_amd has no CPU fallback.")
        # torch first: it brings its own copy of the HIP runtime, and the device memory and streams this library is handed
        # come from that copy -- loaded before torch, the library would bind to the system's libamdhip64 instead and its
        # first call on a torch stream would fail with "no ROCm-capable device is detected"
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in EXPORTS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.dgs_abi_version() != ABI_VERSION:
            raise RuntimeError("libdgs_hip.so ABI version mismatch")
        if not os.environ.get("DGS_LIB_PATH"):   # (an explicitly chosen A/B or sanitizer build is the caller's business)
            verify_build_id(L)
        _lib = L
    return _lib


def verify_build_id(L, csrc=None, header=None):
    """The stale-binary guard: the library carries the SHA-256 of the sources and flags it was built from
    (dgs_build_id()); the same hash is recomputed from the sources next to this file, and a binary that does not match is
    refused -- *.so files are git-ignored but travel to the GPU box, so a commit made after the last local build would
    otherwise be tested and benchmarked with the previous binary and nothing would notice."""
    from . import build as _build
    have = L.dgs_build_id().decode()
    want = _build.build_id(**({} if csrc is None else {"csrc": csrc}), **({} if header is None else {"header": header}))
    if have != want:
        raise RuntimeError(
            f"{LIB_PATH} is stale: it was built from sources with id {have[:16]}..., the sources here have id {want[:16]}... "
            "-- rebuild with `python -m deblurgs_amd.build` (or __graft_entry__.build()).")
    return have


def build_id():
    return lib().dgs_build_id().decode()


# ---- the caller-owned context (include/dgs_hip.h, ABI 14).  The library reads no environment variable; THIS module does,
# once per context, to fill the options struct:
#   DGS_BWD_OVERLAP = 0 | 1 | 2 | 3    DgsContextOptions.bwd_overlap (default 1)
#   DGS_BWD_PARTS   = "10,4"           DgsContextOptions.bwd_parts (subframes per part; the rest is the last part)
_contexts = {}
_context_modes = {}     # handle value -> bwd_overlap it was created with (the struct is opaque on this side)


def context_options_from_env(environ=None):
    env = os.environ if environ is None else environ
    o = DgsContextOptions()
    o.bwd_overlap = int(env.get("DGS_BWD_OVERLAP", "1"))
    spec = [int(x) for x in env.get("DGS_BWD_PARTS", "").split(",") if x.strip()]
    o.bwd_n_parts = min(len(spec), MAX_BWD_PARTS - 1)
    for i in range(o.bwd_n_parts):
        o.bwd_parts[i] = spec[i]
    return o


def create_context(options=None):
    h = ctypes.c_void_p()
    check(lib().dgs_context_create(ctypes.byref(options) if options is not None else None, ctypes.byref(h)),
          "dgs_context_create")
    _context_modes[h.value] = int(options.bwd_overlap) if options is not None else 1
    return h


def destroy_context(h):
    _context_modes.pop(h.value, None)
    check(lib().dgs_context_destroy(h), "dgs_context_destroy")


def context(device=None):
    """The package's context for a device (created on first use with the options the environment asks for; one per
    device: its side stream belongs to the device that is current when the first large backward runs).  Callers that want
    their own policy pass their own handle in DgsProblem.context."""
    if device is None:
        import torch
        device = torch.cuda.current_device() if torch.cuda.is_available() else -1
    key = int(device)
    if key not in _contexts:
        _contexts[key] = create_context(context_options_from_env())
    return _contexts[key]


@contextlib.contextmanager
def context_options(bwd_overlap=1, bwd_parts=(), device=None):
    """Swaps the package's context of a device for one with these options for the duration of the block (tests and A/B
    tools: what DGS_BWD_OVERLAP / DGS_BWD_PARTS do for a whole process).  The device must be idle at both ends."""
    import torch
    if device is None:
        device = torch.cuda.current_device() if torch.cuda.is_available() else -1
    key = int(device)
    o = DgsContextOptions()
    o.bwd_overlap = int(bwd_overlap)
    o.bwd_n_parts = min(len(bwd_parts), MAX_BWD_PARTS - 1)
    for i in range(o.bwd_n_parts):
        o.bwd_parts[i] = int(bwd_parts[i])
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    old = _contexts.get(key)
    _contexts[key] = create_context(o)
    try:
        yield _contexts[key]
    finally:
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        destroy_context(_contexts[key])
        if old is not None:
            _contexts[key] = old
        else:
            del _contexts[key]


def context_overlap_mode(device=None):
    """bwd_overlap of the package's context for the device."""
    h = context(device)
    return _context_modes.get(h.value, 1)


def reset_contexts():
    """Destroys the package's contexts (tests that change DGS_BWD_* between runs; device must be idle)."""
    for h in _contexts.values():
        destroy_context(h)
    _contexts.clear()


def check(rc, what):
    if rc != 0:
        msg = lib().dgs_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libdgs_hip {what} failed (code {rc}): {msg}")


def layout(P, W, H, K, R, wide_records=None):
    """wide_records=None: the package-wide default (diff_gaussian_rasterization.WIDE_RECORDS)."""
    if wide_records is None:
        from . import diff_gaussian_rasterization as dgr
        wide_records = dgr.WIDE_RECORDS
    L = DgsLayout()
    check(lib().dgs_layout(P, W, H, K, R, int(bool(wide_records)), ctypes.byref(L)), "dgs_layout")
    return L


def backward_scratch_layout(R, P, K):
    """(sums_offset, partials_offset) of DgsBackwardIO.scratch, in bytes."""
    so, po = ctypes.c_size_t(), ctypes.c_size_t()
    check(lib().dgs_backward_scratch_layout(int(R), P, K, ctypes.byref(so), ctypes.byref(po)), "scratch_layout")
    return so.value, po.value


def profile_enable(on=True, ctx=None):
    check(lib().dgs_profile_enable(ctx if ctx is not None else context(), 1 if on else 0), "profile_enable")


def profile_reset(ctx=None):
    check(lib().dgs_profile_reset(ctx if ctx is not None else context()), "profile_reset")


def profile_read(ctx=None):
    n = len(STAGES)
    ms = (ctypes.c_float * n)()
    calls = (ctypes.c_int32 * n)()
    check(lib().dgs_profile_read(ctx if ctx is not None else context(), ms, calls, n), "profile_read")
    return {STAGES[i]: (float(ms[i]), int(calls[i])) for i in range(n)}
