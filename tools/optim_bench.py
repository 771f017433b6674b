"""Times the training-side kernels at the metric cloud size: fused multi-tensor Adam against
torch.optim.Adam (foreach and fused flavours), and densify_and_prune.  Prints one JSON line."""
import json
import sys, os
import time
import types

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from deblurgs_amd.cloud import GaussianCloud
from deblurgs_amd.optim import FusedAdam

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=g)
shapes = [(P, 3), (P, 1, 3), (P, 8, 3), (P, 1), (P, 3), (P, 4)]
names = ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"]


def make(kind):
    ps = [torch.nn.Parameter(r(*s)) for s in shapes]
    groups = [{"params": [p], "lr": 1e-3, "name": n} for p, n in zip(ps, names)]
    if kind == "dgs":
        opt = FusedAdam(groups, lr=0.0, eps=1e-15)
    else:
        opt = torch.optim.Adam(groups, lr=0.0, eps=1e-15, foreach=(kind == "foreach"), fused=(kind == "fused"))
    for p in ps:
        p.grad = r(*p.shape) * 1e-2
    return ps, opt


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3


out = {"P": P, "floats_per_gaussian": 38}
bytes_adam = P * 38 * 28
for kind in ("dgs", "foreach", "fused", "single"):
    ps, opt = make(kind)
    ms = timeit(opt.step)
    out[f"adam_{kind}_ms"] = round(ms, 4)
    out[f"adam_{kind}_GBps"] = round(bytes_adam / ms / 1e6, 1)
    del ps, opt

# densify_and_prune on a cloud where ~10 % clone, ~5 % split, ~3 % are pruned
rng = np.random.default_rng(0)
cloud = GaussianCloud(r(P, 3), r(P, 1, 3), r(P, 8, 3), torch.log(torch.rand(P, 3, device=dev, generator=g) * 0.08 + 0.002),
                      r(P, 4), torch.rand(P, 1, device=dev, generator=g) * 0.3 - 0.01, sh_degree=2)
targs = types.SimpleNamespace(iterations=150_000, position_lr_init=0.00016, position_lr_final=0.0000016, feature_lr=0.0025,
                              opacity_lr=0.05, scaling_lr=0.005, rotation_lr=0.001, percent_dense=0.01)
cloud.training_setup(targs)
for p in cloud.hot_parameters():
    p.grad = torch.randn_like(p) * 1e-3
cloud.optimizer.step()
ms = []
for it in range(5):
    Pn = cloud._xyz.shape[0]
    cloud.xyz_gradient_accum = torch.rand(Pn, 1, device=dev, generator=g) * 4.5e-4
    cloud.denom = torch.ones(Pn, 1, device=dev)
    torch.cuda.synchronize()
    t0 = time.time()
    counts = cloud.densify_and_prune(4e-4, 4.0)
    torch.cuda.synchronize()
    ms.append((time.time() - t0) * 1e3)
    out.setdefault("densify_counts", []).append([Pn] + counts)
out["densify_ms"] = [round(m, 3) for m in ms]
# bytes: read 3 x 38 floats per source Gaussian, write 3 x 38 per new one, + flags/offsets/stats
Pn0, (nk, nc, ns, ma) = out["densify_counts"][-1][0], out["densify_counts"][-1][1:]
out["densify_alg_bytes_last"] = int(Pn0 * (38 * 12 + 56) + (nk + nc + 2 * ns) * 38 * 12)
out["densify_GBps_last"] = round(out["densify_alg_bytes_last"] / ms[-1] / 1e6, 1)
print(json.dumps(out))
