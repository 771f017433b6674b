"""Structured overlap experiment: can the HBM-bound list building of one subframe group hide under
the VALU-bound compositing backward of another when the two run on CU-partitioned streams
(hipExtStreamCreateWithCUMask)?  A plain two-stream probe did not overlap them in round 2, because a compositing launch
occupies every CU for its whole duration.

Metric scene, subframe groups A (8 subframes) and B (7).  Measured, per CU split (compositing CUs / list-building CUs):
  t_bwd   dgs_backward_composite(A) alone on its stream          (VALU-bound, ~53 % of a step's backward)
  t_fwd   dgs_forward_lists(B) (preprocess ... tile ranges: the HBM-bound half of the forward) alone on its stream
  t_both  both started together
gain = t_bwd + t_fwd - t_both is what a two-group pipeline could win per such pair BEFORE paying for the split itself
(two preprocess / sort / geometry set-ups instead of one: 0.36 ms per step, measured in round 2).

    python tools/cu_mask_probe.py            # prints one line per split and a keep / drop verdict
"""
import ctypes
import math
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def masked_stream(hip, n_cu_total, pick):
    """Stream restricted to the CUs in `pick` (hipExtStreamCreateWithCUMask; bit i of the mask = CU i)."""
    words = (n_cu_total + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for cu in pick:
        mask[cu // 32] |= (1 << (cu % 32))
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), words, mask)
    assert rc == 0, f"hipExtStreamCreateWithCUMask failed ({rc})"
    return st


def main():
    import numpy as np
    import torch
    from deblurgs_amd import _lib, synthetic
    from deblurgs_amd import diff_gaussian_rasterization as dgr
    from deblurgs_amd.cloud import GaussianCloud
    hip = ctypes.CDLL("libamdhip64.so")
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    sc = synthetic.make_config("metric")
    P, W, H, K = sc["P"], sc["W"], sc["H"], sc["K"]
    cloud = GaussianCloud.from_scene(sc, dev)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    bg = t(sc["bg"])
    f32 = dict(dtype=torch.float32, device=dev)

    def problem(k0, k1):
        Kg = k1 - k0
        view, full, campos = t(sc["viewmatrix"][k0:k1]), t(sc["projmatrix"][k0:k1]), t(sc["campos"][k0:k1])
        geom = torch.empty(L.dgs_geom_state_bytes(P, Kg), dtype=torch.uint8, device=dev)
        image = torch.empty(L.dgs_image_state_bytes(W, H, Kg), dtype=torch.uint8, device=dev)
        p = _lib.DgsProblem()
        p.P, p.D, p.M, p.W, p.H, p.K = P, 2, 9, W, H, Kg
        p.tanfovx, p.tanfovy = sc["tanfovx"], sc["tanfovy"]
        p.scale_modifier, p.z_near, p.z_far = 1.0, 0.2, 100.0
        p.tile_cull, p.raw_params, p.scale_lb = 1, 1, 0.0
        ptr = lambda x: ctypes.c_void_p(x.data_ptr())
        p.means3D, p.shs, p.shs_rest = ptr(cloud._xyz), ptr(cloud._features_dc), ptr(cloud._features_rest)
        p.opacities, p.scales, p.rotations = ptr(cloud._opacity), ptr(cloud._scaling), ptr(cloud._rotation)
        p.viewmatrix, p.projmatrix, p.campos, p.bg = ptr(view), ptr(full), ptr(campos), ptr(bg)
        p.geom_state, p.geom_bytes = ptr(geom), geom.numel()
        p.image_state, p.image_bytes = ptr(image), image.numel()
        color = torch.empty((Kg, 3, H, W), **f32)
        radii = torch.empty((Kg, P), dtype=torch.int32, device=dev)
        host = torch.zeros(8, dtype=torch.int32).pin_memory()
        out = _lib.DgsForwardOut()
        out.out_color, out.out_depth, out.radii = ptr(color), None, ptr(radii)
        out.num_rendered_host = ctypes.c_void_p(host.data_ptr())
        keep = [view, full, campos, geom, image, color, radii, host]
        return p, out, host, keep

    s0 = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    # ---- group A: forward once (exact), backward inputs
    pA, oA, hA, keepA = problem(0, 8)
    _lib.check(L.dgs_forward_geometry(ctypes.byref(pA), ctypes.byref(oA), s0), "geomA")
    torch.cuda.synchronize()
    RA = int(hA[0]) & 0xFFFFFFFF
    binA = torch.empty(L.dgs_binning_state_bytes(RA, W, H, 8), dtype=torch.uint8, device=dev)
    pA.binning_state, pA.binning_bytes = ctypes.c_void_p(binA.data_ptr()), binA.numel()
    _lib.check(L.dgs_forward_render(ctypes.byref(pA), ctypes.byref(oA), RA, s0), "renderA")
    gC = torch.randn((8, 3, H, W), **f32) / (3 * W * H)
    scratch = torch.empty(L.dgs_backward_scratch_bytes(RA, P, 8), dtype=torch.uint8, device=dev)
    io = _lib.DgsBackwardIO()
    io.num_rendered, io.radii, io.dL_dout_color = RA, ctypes.c_void_p(keepA[6].data_ptr()), ctypes.c_void_p(gC.data_ptr())
    io.scratch, io.scratch_bytes = ctypes.c_void_p(scratch.data_ptr()), scratch.numel()
    outs = {n: torch.empty(shape, **f32) for n, shape in dict(
        m3=(P, 3), m2=(8, P, 3), dc=(P, 1, 3), rest=(P, 8, 3), col=(P, 3), op=(P, 1), sc_=(P, 3), rot=(P, 4), cov=(P, 6),
        view=(8, 4, 4), proj=(8, 4, 4)).items()}
    g = lambda n: ctypes.c_void_p(outs[n].data_ptr())
    io.dL_dmeans3D, io.dL_dmeans2D, io.dL_dsh, io.dL_dsh_rest, io.dL_dcolors = g("m3"), g("m2"), g("dc"), g("rest"), g("col")
    io.dL_dopacity, io.dL_dscales, io.dL_drotations, io.dL_dcov3D = g("op"), g("sc_"), g("rot"), g("cov")
    io.dL_dviewmatrix, io.dL_dprojmatrix = g("view"), g("proj")
    # ---- group B: capacity from an exact count
    pB, oB, hB, keepB = problem(8, 15)
    _lib.check(L.dgs_forward_geometry(ctypes.byref(pB), ctypes.byref(oB), s0), "geomB")
    torch.cuda.synchronize()
    RB = int(hB[0]) & 0xFFFFFFFF
    capB = RB + RB // 8
    binB = torch.empty(L.dgs_binning_state_bytes(capB, W, H, 7), dtype=torch.uint8, device=dev)
    pB.binning_state, pB.binning_bytes = ctypes.c_void_p(binB.data_ptr()), binB.numel()

    def bwdA(stream):
        _lib.check(L.dgs_backward_composite(ctypes.byref(pA), ctypes.byref(io), stream), "bwdA")

    def fwdB(stream):
        _lib.check(L.dgs_forward_lists(ctypes.byref(pB), ctypes.byref(oB), capB, stream), "fwdB")   # HBM-bound half only

    def timed(fn_list, reps=8):
        """fn_list: [(fn, stream_handle)]; all started together; returns mean wall ms until all have finished."""
        evs = []
        for _ in range(reps + 2):
            torch.cuda.synchronize()
            start = torch.cuda.Event(enable_timing=True)
            start.record()
            ends = []
            for fn, st in fn_list:
                ext = torch.cuda.ExternalStream(st.value)
                ext.wait_event(start)
                fn(st)
                e = torch.cuda.Event(enable_timing=True)
                e.record(ext)
                ends.append(e)
            torch.cuda.synchronize()
            evs.append(max(start.elapsed_time(e) for e in ends))
        return float(np.mean(evs[2:]))

    print(f"CUs: {n_cu}; R_A = {RA}, R_B = {RB}")
    rows = []
    for n_small, layout in ((0, "none"), (32, "stride"), (32, "block"), (16, "stride"), (64, "stride")):
        if n_small == 0:
            sA = masked_stream(hip, n_cu, range(n_cu))
            sB = masked_stream(hip, n_cu, range(n_cu))
        else:
            if layout == "stride":
                step = n_cu // n_small
                small = [i for i in range(n_cu) if i % step == 0][:n_small]
            else:
                small = list(range(n_small))
            big = [i for i in range(n_cu) if i not in set(small)]
            sA, sB = masked_stream(hip, n_cu, big), masked_stream(hip, n_cu, small)
        t_b = timed([(bwdA, sA)])
        t_f = timed([(fwdB, sB)])
        t_x = timed([(bwdA, sA), (fwdB, sB)])
        rows.append((n_small, layout, t_b, t_f, t_x))
        print(f"list-building CUs {n_small:3d} ({layout:6s}): t_bwd {t_b:6.3f} ms  t_fwd {t_f:6.3f} ms  both {t_x:6.3f} ms  "
              f"gain vs serial-on-all-CUs {rows[0][2] + rows[0][3] - t_x:+.3f} ms", flush=True)
    best = max(rows[0][2] + rows[0][3] - r[4] for r in rows)
    print(f"best gain per (backward A, forward B) pair: {best:+.3f} ms; a two-group step has one such pair plus the mirrored "
          f"one and pays ~0.36 ms for the split -> {'KEEP: worth building the pipeline' if 2 * best - 0.36 > 0.5 else 'DROP'}")


if __name__ == "__main__":
    main()
