"""Functional check of the data-parallel training loop on a ONE-GPU box: every rank trains on its own blurry view
(TrainingLoop(distributed=True)), with densification, and the replicas must stay bit-identical.
    DGS_DIST_BACKEND=gloo DGS_DIST_ONE_DEVICE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \\
        --master-addr 127.0.0.1 --master-port 29541 tools/dist_training_check.py
"""
import os, sys
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from deblurgs_amd import sharding, synthetic
from deblurgs_amd.cloud import GaussianCloud
from deblurgs_amd.motion import CameraMotionModule, RefCamera
from deblurgs_amd.training import TrainingLoop, default_optimization_params

rank, world, local = sharding.init_distributed("cuda")
dev = torch.device("cuda", 0)
sc = synthetic.make_scene(3000, 128, 96, K=5, seed=21, sigma_px=3.0)
cloud = GaussianCloud.from_scene(sc, dev)
ref = RefCamera(sc["W"], sc["H"], sc["FoVx"], sc["FoVy"], device=dev)
torch.manual_seed(100 + rank)
gt = torch.rand(1, 3, sc["H"], sc["W"], device=dev) * 0.5
m = CameraMotionModule(ref, gt, curve_order=3, num_subframes=5, device=dev)
with torch.no_grad():
    m._trans._control_points.copy_(torch.from_numpy(sc["ctrl_trans"])[None].to(dev) + 0.01 * rank)
    m._rot._control_points.copy_(torch.from_numpy(sc["ctrl_rot"])[None].to(dev))
opt = default_optimization_params(iterations=40, curve_start_iter=2, densify_from_iter=5, densification_interval=6,
                                  densify_until_iter=30, densify_grad_threshold_init=2e-5, densify_grad_threshold_final=1e-5,
                                  opacity_reset_interval=1000)
loop = TrainingLoop(cloud, m, opt, cameras_extent=1.0, distributed=world > 1)
sizes = []
for it in range(1, 31):
    torch.manual_seed(it)              # same random background on every rank
    out = loop.step(it, 0)
    sizes.append(out["num_points"])
inplace = []
_orig = sharding.flat_allreduce_grads
def _spy(params, **kw):
    r = _orig(params, **kw)
    inplace.append(sharding._shared_flat([p.grad for p in params]) is not None)
    return r
sharding.flat_allreduce_grads = _spy
for it in range(31, 34):
    torch.manual_seed(it)
    loop.step(it, 0)
P = cloud._xyz.shape[0]
sig = torch.stack([p.detach().double().sum() for p in cloud.hot_parameters()] + [torch.tensor(float(P), device=dev, dtype=torch.float64)])
if world > 1:
    allsig = [torch.zeros_like(sig) for _ in range(world)]
    dist.all_gather(allsig, sig)
    same = all(torch.equal(allsig[0], s) for s in allsig)
else:
    same = True
if rank == 0:
    print("ranks", world, "points", sizes[0], "->", sizes[-1], "densified:", len(set(sizes)) > 1, "replicas identical:", same,
          "gradient bucket reduced in place:", inplace)
    assert same and len(set(sizes)) > 1 and (world == 1 or all(inplace))
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
