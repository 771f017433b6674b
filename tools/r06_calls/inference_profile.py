"""Where a single-camera inference call (gaussian_renderer.render under no_grad, metric scene) spends its 0.5 ms:
cProfile of 300 calls + the GPU time of the same calls (events)."""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deblurgs_amd import gaussian_renderer, synthetic  # noqa: E402
from deblurgs_amd.cloud import GaussianCloud  # noqa: E402
from deblurgs_amd.motion import CameraMotionModule, RefCamera  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "metric"
sc = synthetic.make_config(name, seed=0)
dev = torch.device("cuda:0")
cloud = GaussianCloud.from_scene(sc, "cuda")
ref = RefCamera(sc["W"], sc["H"], sc["FoVx"], sc["FoVy"], device="cuda")
K = sc["K"]
gt = torch.rand(1, 3, sc["H"], sc["W"], device="cuda")
m = CameraMotionModule(ref, gt, curve_order=3, num_subframes=K, device="cuda")
bg = torch.zeros(3, device="cuda")
with torch.no_grad():
    cams = m.get_trajectory(0)
    for c in cams[:3]:
        gaussian_renderer.render(c, cloud, bg)
    torch.cuda.synchronize()
    n = 300
    t0 = time.time()
    for i in range(n):
        gaussian_renderer.render(cams[i % K], cloud, bg)
    torch.cuda.synchronize()
    print(f"{name}: {(time.time() - t0) / n * 1e3:.3f} ms per call")
    if len(sys.argv) > 2 and sys.argv[2] == "noprof":
        sys.exit(0)
    pr = cProfile.Profile()
    pr.enable()
    for i in range(n):
        gaussian_renderer.render(cams[i % K], cloud, bg)
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(18)
