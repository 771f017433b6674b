"""Round-2 fixtures, produced by RUNNING reference code on the CPU in the build container (never on the GPU box):

  chkpnt_ref.pth       the file train.py:214-216 writes -- torch.save((gaussians.capture(), iteration)) -- from the
                       reference's own GaussianModel (scene/gaussian_model.py:80-95) after training_setup and three
                       torch.optim.Adam steps, plus what the reference holds after ONE MORE step on stated gradients
                       (chkpnt_ref_next.npz), so that a restore()d cloud can be stepped and compared.
  schedule2_golden.npz get_scheduler samples (utils/general_utils.py:72-101), the curve_alignment learning-rate
                       schedule of train.py:90-94.

The fixtures are data: tensors, an optimiser state dict and numbers.  Import shims as in make_golden_densify.py
(plyfile / simple_knn._C are imported at module import time by the reference and are not on this path).
    python tests/golden/make_golden_r2.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden_densify import REF, _CpuDevice, _shims   # noqa: E402


def _load_gaussian_model():
    import importlib.util
    _shims()
    sys.path.insert(0, REF)
    spec = importlib.util.spec_from_file_location("ref_gaussian_model", os.path.join(REF, "scene/gaussian_model.py"))
    gm = importlib.util.module_from_spec(spec)
    act_spec = importlib.util.spec_from_file_location("scene.gaussian_activation",
                                                      os.path.join(REF, "scene/gaussian_activation.py"))
    scene_pkg = types.ModuleType("scene")
    scene_pkg.__path__ = [os.path.join(REF, "scene")]
    sys.modules["scene"] = scene_pkg
    act = importlib.util.module_from_spec(act_spec)
    sys.modules["scene.gaussian_activation"] = act
    act_spec.loader.exec_module(act)
    spec.loader.exec_module(gm)
    return gm


def main():
    gm = _load_gaussian_model()
    from utils import general_utils
    rng = np.random.default_rng(7)
    torch.manual_seed(7)
    P, M = 48, 9
    margs = types.SimpleNamespace(sh_degree=2, z_near=0.2, z_far=100.0, alpha_lower_bound=0.0, scale_lb=0.0,
                                  scale_ub=-1.0, use_isotrophic=False, activation="relu")
    targs = types.SimpleNamespace(iterations=150_000, position_lr_init=0.00016, position_lr_final=0.0000016,
                                  feature_lr=0.0025, opacity_lr=0.05, scaling_lr=0.005, rotation_lr=0.001,
                                  percent_dense=0.01)
    g = gm.GaussianModel(margs)
    sched = types.SimpleNamespace(curve_start_iter=1000, curve_lr_half_iter=15_000)
    f32 = lambda a: torch.tensor(a, dtype=torch.float32)
    names = ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"]
    with _CpuDevice():
        g._xyz = torch.nn.Parameter(f32(rng.normal(0, 1, (P, 3))))
        g._features_dc = torch.nn.Parameter(f32(rng.normal(0, 0.3, (P, 1, 3))))
        g._features_rest = torch.nn.Parameter(f32(rng.normal(0, 0.1, (P, M - 1, 3))))
        g._scaling = torch.nn.Parameter(f32(np.log(rng.uniform(0.002, 0.08, (P, 3)))))
        g._rotation = torch.nn.Parameter(f32(rng.normal(0, 1, (P, 4))))
        g._opacity = torch.nn.Parameter(f32(rng.uniform(0.0, 0.9, (P, 1))))
        g.max_radii2D = torch.zeros((P,))
        g.spatial_lr_scale = 2.5
        g.active_sh_degree = 1
        g.training_setup(targs)
        params = {n: grp["params"][0] for n, grp in zip(names, g.optimizer.param_groups)}
        for it in range(3):
            for n, p in params.items():
                p.grad = f32(rng.normal(0, 1e-2, tuple(p.shape)))
            g.update_learning_rate(100 + it, sched)
            g.optimizer.step()
            g.optimizer.zero_grad(set_to_none=True)
        g.xyz_gradient_accum = f32(rng.uniform(0, 1e-3, (P, 1)))
        g.denom = f32(rng.integers(0, 4, (P, 1)).astype(np.float32))
        g.max_radii2D = f32(rng.uniform(0, 30, (P,)))
        torch.save((g.capture(), 103), os.path.join(HERE, "chkpnt_ref.pth"))
        nxt = {}
        for n, p in params.items():
            gr = rng.normal(0, 1e-2, tuple(p.shape)).astype(np.float32)
            nxt["grad_" + n] = gr
            p.grad = f32(gr)
        g.update_learning_rate(104, sched)
        nxt["lr_xyz"] = np.array(g.optimizer.param_groups[0]["lr"])
        g.optimizer.step()
        for n, p in params.items():
            nxt["param_" + n] = p.detach().numpy().copy()
            nxt["m_" + n] = g.optimizer.state[p]["exp_avg"].numpy().copy()
            nxt["v_" + n] = g.optimizer.state[p]["exp_avg_sq"].numpy().copy()
        nxt["step"] = np.array(float(g.optimizer.state[params["xyz"]]["step"]))
    np.savez_compressed(os.path.join(HERE, "chkpnt_ref_next.npz"), **nxt)

    steps = np.array([1, 2, 29_999, 30_000, 30_001, 40_000, 90_000, 150_000, 150_001, 200_000])
    f = general_utils.get_scheduler(lr_init=1e-3, lr_final=1e-7, warmup_ratio=0.0, step_warmup=30_000,
                                    step_final=150_000)
    f0 = general_utils.get_scheduler(lr_init=0.0, lr_final=1e-7, warmup_ratio=0.0, step_warmup=30_000,
                                     step_final=150_000)
    np.savez_compressed(os.path.join(HERE, "schedule2_golden.npz"), steps=steps,
                        alignment=np.array([f(int(s)) for s in steps]),
                        alignment_lr0=np.array([f0(int(s)) for s in steps]))
    print("written", os.listdir(HERE))


if __name__ == "__main__":
    main()
