"""Measuring tool: forward + backward of the K subframes as ONE problem on one stream against TWO subframe groups
pipelined over two streams -- group B's duplicate lists (HBM-bound) under group A's compositing (VALU-bound), group A's
per-Gaussian backward half under group B's compositing backward (dgs_forward_phase, DgsBackwardIO.phase).
The pipelined variant here is a timing skeleton: its per-Gaussian gradients are those of the last group only.
    python tools/pipeline_probe.py [config] [KA]
"""
import ctypes
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import synthetic, _t
from deblurgs_amd import _lib
from deblurgs_amd.cloud import GaussianCloud

cfg = sys.argv[1] if len(sys.argv) > 1 else "metric"
sc = synthetic.make_config(cfg)
K = sc["K"]
KA = int(sys.argv[2]) if len(sys.argv) > 2 else (K + 1) // 2
c = GaussianCloud.from_scene(sc, "cuda")
dev = c._xyz.device
L = _lib.lib()
P, W, H = sc["P"], sc["W"], sc["H"]
view, proj, cam, bg = _t(sc["viewmatrix"]), _t(sc["projmatrix"]), _t(sc["campos"]), _t(sc["bg"])
f32 = dict(dtype=torch.float32, device=dev)
ptr = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
Mr = c._features_rest.shape[1]


class Group:
    def __init__(self, k0, k1):
        self.k0, self.k1, self.K = k0, k1, k1 - k0
        Kg = self.K
        self.color = torch.empty((Kg, 3, H, W), **f32)
        self.depth = torch.empty((Kg, 1, H, W), **f32)
        self.radii = torch.empty((Kg, P), dtype=torch.int32, device=dev)
        self.geom = torch.empty(L.dgs_geom_state_bytes(P, Kg), dtype=torch.uint8, device=dev)
        self.image = torch.empty(L.dgs_image_state_bytes(W, H, Kg), dtype=torch.uint8, device=dev)
        self.v, self.p, self.cp = view[k0:k1].contiguous(), proj[k0:k1].contiguous(), cam[k0:k1].contiguous()
        pr = _lib.DgsProblem()
        pr.P, pr.D, pr.M, pr.W, pr.H, pr.K = P, 2, 1 + Mr, W, H, Kg
        pr.tanfovx, pr.tanfovy = math.tan(sc["FoVx"] * 0.5), math.tan(sc["FoVy"] * 0.5)
        pr.scale_modifier, pr.z_near, pr.z_far = 1.0, sc["z_near"], sc["z_far"]
        pr.tile_cull, pr.raw_params, pr.scale_lb = 1, 1, 0.0
        pr.means3D, pr.shs, pr.shs_rest = ptr(c._xyz), ptr(c._features_dc), ptr(c._features_rest)
        pr.opacities, pr.scales, pr.rotations = ptr(c._opacity), ptr(c._scaling), ptr(c._rotation)
        pr.viewmatrix, pr.projmatrix, pr.campos, pr.bg = ptr(self.v), ptr(self.p), ptr(self.cp), ptr(bg)
        pr.geom_state, pr.geom_bytes = ptr(self.geom), self.geom.numel()
        pr.image_state, pr.image_bytes = ptr(self.image), self.image.numel()
        self.prob = pr
        self.host = torch.zeros(4, dtype=torch.int32).pin_memory()
        out = _lib.DgsForwardOut()
        out.out_color, out.out_depth, out.radii = ptr(self.color), ptr(self.depth), ptr(self.radii)
        out.num_rendered_host = ctypes.c_void_p(self.host.data_ptr())
        self.out = out
        # learn the count once (exact two-phase forward), then size ahead
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(L.dgs_forward_geometry(ctypes.byref(pr), ctypes.byref(out), st), "geometry")
        torch.cuda.synchronize()
        self.cap = int(self.host[0]) + int(self.host[0]) // 10 + 4096
        self.binning = torch.empty(L.dgs_binning_state_bytes(self.cap, W, H, Kg), dtype=torch.uint8, device=dev)
        pr.binning_state, pr.binning_bytes = ptr(self.binning), self.binning.numel()
        self.dsub = torch.randn((Kg, 3, H, W), **f32) * 1e-6
        self.flat = torch.empty(P * 40 + 64, **f32)
        self.g2d = torch.empty((Kg, P, 3), **f32)
        self.gcol, self.gcov = torch.empty((P, 3), **f32), torch.empty((P, 6), **f32)
        self.gv, self.gp = torch.empty((Kg, 4, 4), **f32), torch.empty((Kg, 4, 4), **f32)
        self.scratch = torch.empty(L.dgs_backward_scratch_bytes(self.cap, P, Kg), dtype=torch.uint8, device=dev)
        io = _lib.DgsBackwardIO()
        io.num_rendered = self.cap
        io.radii, io.dL_dout_color, io.dL_dout_depth = ptr(self.radii), ptr(self.dsub), None
        io.scratch, io.scratch_bytes = ptr(self.scratch), self.scratch.numel()
        fl = self.flat
        seg = lambda a, n: ctypes.c_void_p(fl.data_ptr() + 4 * a)
        io.dL_dmeans3D, io.dL_dsh, io.dL_dsh_rest = seg(0, 0), seg(3 * P, 0), seg(6 * P, 0)
        io.dL_dopacity, io.dL_dscales, io.dL_drotations = seg((6 + 3 * Mr) * P, 0), seg((7 + 3 * Mr) * P, 0), seg((10 + 3 * Mr) * P, 0)
        io.dL_dmeans2D, io.dL_dcolors, io.dL_dcov3D = ptr(self.g2d), ptr(self.gcol), ptr(self.gcov)
        io.dL_dviewmatrix, io.dL_dprojmatrix = ptr(self.gv), ptr(self.gp)
        self.io = io

    def fwd(self, phase, stream):
        _lib.check(L.dgs_forward_phase(ctypes.byref(self.prob), ctypes.byref(self.out), self.cap, phase,
                                       ctypes.c_void_p(stream.cuda_stream)), "forward")

    def bwd(self, phase, stream):
        self.io.phase = phase
        _lib.check(L.dgs_backward(ctypes.byref(self.prob), ctypes.byref(self.io), ctypes.c_void_p(stream.cuda_stream)),
                   "backward")


whole, A, B = Group(0, K), Group(0, KA), Group(KA, K)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def single():
    whole.fwd(0, s1)
    whole.bwd(0, s1)


def pipelined(fwd_overlap=True, bwd_overlap=True):
    A.fwd(1, s1)
    e1 = torch.cuda.Event(); e1.record(s1)
    A.fwd(2, s1)
    sB = s2 if fwd_overlap else s1
    if fwd_overlap:
        s2.wait_event(e1)
    B.fwd(1, sB)
    B.fwd(2, sB)
    if fwd_overlap:
        eB = torch.cuda.Event(); eB.record(s2); s1.wait_event(eB)
    # (the loss kernel over all K subframes would sit here)
    A.bwd(1, s1)
    e2 = torch.cuda.Event(); e2.record(s1)
    B.bwd(1, s1)
    sG = s2 if bwd_overlap else s1
    if bwd_overlap:
        s2.wait_event(e2)
    A.bwd(2, sG)
    if bwd_overlap:
        e3 = torch.cuda.Event(); e3.record(s2); s1.wait_event(e3)
    B.bwd(2, s1)


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s1)
    for _ in range(n):
        f()
    b.record(s1)
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


print(f"{cfg}: K={K} split {KA}+{K - KA}; counts whole {int(whole.host[0])}, A {int(A.host[0])}, B {int(B.host[0])}")
for rep in range(2):
    print("one problem, one stream          %.3f ms" % timeit(single))
    print("two groups, one stream           %.3f ms" % timeit(lambda: pipelined(False, False)))
    print("two groups, forward overlapped   %.3f ms" % timeit(lambda: pipelined(True, False)))
    print("two groups, backward overlapped  %.3f ms" % timeit(lambda: pipelined(False, True)))
    print("two groups, both overlapped      %.3f ms" % timeit(lambda: pipelined(True, True)))
