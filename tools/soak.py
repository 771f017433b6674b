"""Soak run of the training loop (fused render + loss + densification + fused Adam) to catch rare failures, leaks and
cloud-size blow-ups: N iterations on a cfg2-sized scene with a densify every 20 iterations and an opacity reset.
usage: python tools/soak.py [N] [auto|always] [replays]   -- "replays": 1M Gaussians, no densification (four captures, then
replays only: is device memory lost per capture or per launch?)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from deblurgs_amd import synthetic
from deblurgs_amd.cloud import GaussianCloud
from deblurgs_amd.motion import CameraMotionModule, RefCamera
from deblurgs_amd.training import TrainingLoop, default_optimization_params
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
GRAPH = sys.argv[2] if len(sys.argv) > 2 else "auto"      # "always": capture whenever possible (stresses re-capture + pool release)
REPLAYS_ONLY = len(sys.argv) > 3 and sys.argv[3] == "replays"
dev = torch.device("cuda", 0)
sc = synthetic.make_scene(1_000_000 if REPLAYS_ONLY else 100_000, 800, 800, K=9, curve_order=5, seed=3, sigma_px=2.0)
ref = RefCamera(sc["W"], sc["H"], sc["FoVx"], sc["FoVy"], device=dev)
views = 4
with torch.no_grad():
    cloud_gt = GaussianCloud.from_scene(sc, dev)
    m_gt = CameraMotionModule(ref, torch.zeros(views, 3, sc["H"], sc["W"], device=dev), curve_order=5, num_subframes=9, device=dev)
    m_gt._trans._control_points.copy_(torch.from_numpy(sc["ctrl_trans"])[None].to(dev) + 0.02 * torch.randn(views, 6, 3, device=dev))
    m_gt.link_gaussian(cloud_gt)
    gts = torch.stack([m_gt.query(v, "all", background=torch.zeros(3, device=dev))["blurred"] for v in range(views)])
cloud = GaussianCloud.from_scene(sc, dev)
with torch.no_grad():
    cloud._features_dc.add_(torch.randn_like(cloud._features_dc) * 0.2)
    cloud._xyz.add_(torch.randn_like(cloud._xyz) * 0.005)
m = CameraMotionModule(ref, gts, curve_order=5, num_subframes=9, device=dev)
opt = default_optimization_params(iterations=N + 1, curve_start_iter=20, densify_from_iter=10**9 if REPLAYS_ONLY else 30, densification_interval=20,
                                  densify_until_iter=N, opacity_reset_interval=150, densify_grad_threshold_init=1e-5,
                                  densify_grad_threshold_final=5e-6)
loop = TrainingLoop(cloud, m, opt, cameras_extent=2.0, graph=GRAPH)
t0 = time.time()
first = None
for it in range(1, N + 1):
    out = loop.step(it, it % views)
    if it % 25 == 0:
        torch.cuda.synchronize()
        l1 = float(out["l1"])
        first = first or l1
        assert l1 == l1, "NaN loss"
        free_b, total_b = torch.cuda.mem_get_info(dev)      # device-level: also what the runtime holds outside torch's allocator
        print(f"it {it:4d}  l1 {l1:.5f}  points {out['num_points']:7d}  mem {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB  "
              f"reserved {torch.cuda.memory_reserved() / 2**30:.2f}  device in use {(total_b - free_b) / 2**30:.2f} GiB  "
              f"{time.time() - t0:.1f} s", flush=True)
assert float(out["l1"]) < first
print("soak ok")
