"""Why did bench.py's value_with_depth region take 19.5 ms per step?  Per-step wall times of a TrainingLoop at the metric size
with FusedStep.always_depth switched on after an invalidate, as bench.py's side_region does."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench   # noqa: E402

args = bench.parse_args(["--steps", "5", "--warmup", "3", "--no-cpu-baseline"])
# re-use bench's own set-up by running its pieces by hand is long; instead time the regions through the public switch
from deblurgs_amd import synthetic   # noqa: E402
from deblurgs_amd.cloud import GaussianCloud   # noqa: E402
from deblurgs_amd.motion import CameraMotionModule, RefCamera   # noqa: E402
from deblurgs_amd.training import TrainingLoop, default_optimization_params   # noqa: E402

dev = torch.device("cuda", 0)
scene = synthetic.make_config("metric", seed=0, sh_degree=2)
W, H, K = scene["W"], scene["H"], scene["K"]
cloud = GaussianCloud.from_scene(scene, dev)
ref = RefCamera(W, H, scene["FoVx"], scene["FoVy"], device=dev)
gt = torch.rand((1, 3, H, W)).to(dev)
mo = CameraMotionModule(ref, gt, curve_order=3, num_subframes=K, device=dev)
mo.link_gaussian(cloud)
far = 10 ** 9
opt = default_optimization_params(iterations=far, lambda_hinge=0.1, curve_start_iter=1, curve_end_iter=far,
                                  densify_from_iter=far, densify_until_iter=far, opacity_reset_interval=far)
loop = TrainingLoop(cloud, mo, opt, cameras_extent=1.0, spatial_lr_scale=1.0, log_losses=False, graph="auto")
for g in cloud.optimizer.param_groups:
    g["lr"] *= 1e-6
it = [0]


def run(n, tag):
    ts = []
    for _ in range(n):
        it[0] += 1
        torch.cuda.synchronize()
        t0 = time.time()
        loop.step(it[0], 0)
        torch.cuda.synchronize()
        ts.append((time.time() - t0) * 1e3)
    print(tag, " ".join(f"{t:.1f}" for t in ts), "| dropped", loop._fused.dropped, "retried", loop.retried, flush=True)


run(8, "plain       ")
fs = loop._fused
fs._poll(block=True)
fs.invalidate()
fs.always_depth = True
run(16, "with depth  ")
fs._poll(block=True)
fs.invalidate()
fs.always_depth = False
run(8, "plain again ")
