"""Regenerates tests/golden/densify_golden.npz by RUNNING the reference's GaussianModel (scene/gaussian_model.py)
on the CPU in the build container: training_setup -> a few torch.optim.Adam steps -> add_densification_stats ->
densify_and_prune -> reset_opacity.  The fixture is data only (seeded inputs, the captured normal samples, and
the reference's outputs).

The reference hard-codes device="cuda" and imports two packages this image lacks at module import time
(`plyfile`, `simple_knn._C`); neither is on the densification path, so the script (a) registers empty import
shims for those two names and (b) redirects device="cuda" factory calls to the CPU while the reference code runs.
torch.normal is wrapped to record the unit-normal draws z = samples / std that densify_and_split consumed.
    python tests/golden/make_golden_densify.py
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _shims():
    ply = types.ModuleType("plyfile")
    ply.PlyData = ply.PlyElement = None
    sys.modules["plyfile"] = ply
    knn = types.ModuleType("simple_knn")
    knn_c = types.ModuleType("simple_knn._C")
    knn_c.distCUDA2 = None
    sys.modules["simple_knn"] = knn
    sys.modules["simple_knn._C"] = knn_c


class _CpuDevice:
    """Redirect device='cuda' in tensor factories to the CPU while the reference runs."""
    NAMES = ("zeros", "ones", "empty", "tensor", "zeros_like", "ones_like")

    def __enter__(self):
        self.saved = {n: getattr(torch, n) for n in self.NAMES}
        for n, fn in self.saved.items():
            def wrap(*a, __fn=fn, **kw):
                if "device" in kw and str(kw["device"]).startswith("cuda"):
                    kw["device"] = "cpu"
                return __fn(*a, **kw)
            setattr(torch, n, wrap)
        self.empty_cache = torch.cuda.empty_cache
        torch.cuda.empty_cache = lambda: None
        self.normal = torch.normal
        self.draws = []

        def normal(mean=None, std=None, **kw):
            out = self.normal(mean=mean, std=std, **kw)
            self.draws.append((out / std).detach().clone())
            return out
        torch.normal = normal
        return self

    def __exit__(self, *exc):
        for n, fn in self.saved.items():
            setattr(torch, n, fn)
        torch.cuda.empty_cache = self.empty_cache
        torch.normal = self.normal


def main():
    _shims()
    sys.path.insert(0, REF)
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_gaussian_model", os.path.join(REF, "scene/gaussian_model.py"))
    gm = importlib.util.module_from_spec(spec)
    # scene/__init__.py pulls dataset readers; load the module by path with its package-relative import satisfied
    act_spec = importlib.util.spec_from_file_location("scene.gaussian_activation",
                                                      os.path.join(REF, "scene/gaussian_activation.py"))
    scene_pkg = types.ModuleType("scene")
    scene_pkg.__path__ = [os.path.join(REF, "scene")]
    sys.modules["scene"] = scene_pkg
    act = importlib.util.module_from_spec(act_spec)
    sys.modules["scene.gaussian_activation"] = act
    act_spec.loader.exec_module(act)
    spec.loader.exec_module(gm)

    out = {}
    # case c: use_isotrophic (one shared scale per Gaussian = column 0 of _scaling, scene/gaussian_model.py:115-118)
    for case, (scale_lb, alpha_lb, seed, iso) in {"a": (0.0, 0.0, 0, False), "b": (0.002, 0.01, 1, False),
                                                  "c": (0.001, 0.0, 2, True)}.items():
        rng = np.random.default_rng(seed)
        torch.manual_seed(seed)
        P, M = 400, 9
        margs = types.SimpleNamespace(sh_degree=2, z_near=0.2, z_far=100.0, alpha_lower_bound=alpha_lb, scale_lb=scale_lb,
                                      scale_ub=-1.0, use_isotrophic=iso, activation="relu")
        targs = types.SimpleNamespace(iterations=150_000, position_lr_init=0.00016, position_lr_final=0.0000016,
                                      feature_lr=0.0025, opacity_lr=0.05, scaling_lr=0.005, rotation_lr=0.001,
                                      percent_dense=0.01)
        g = gm.GaussianModel(margs)
        f32 = lambda a: torch.tensor(a, dtype=torch.float32)
        inp = dict(xyz=rng.normal(0, 1, (P, 3)), f_dc=rng.normal(0, 0.3, (P, 1, 3)), f_rest=rng.normal(0, 0.1, (P, M - 1, 3)),
                   scaling=np.log(rng.uniform(0.002, 0.08, (P, 3))), rotation=rng.normal(0, 1, (P, 4)),
                   opacity=rng.uniform(-0.05, 0.6, (P, 1)))
        inp["opacity"][:40] = rng.uniform(0.0, 0.012, (40, 1))      # around min_opacity
        inp = {k: v.astype(np.float32) for k, v in inp.items()}
        with _CpuDevice() as dev:
            g._xyz = torch.nn.Parameter(f32(inp["xyz"]))
            g._features_dc = torch.nn.Parameter(f32(inp["f_dc"]))
            g._features_rest = torch.nn.Parameter(f32(inp["f_rest"]))
            g._scaling = torch.nn.Parameter(f32(inp["scaling"]))
            g._rotation = torch.nn.Parameter(f32(inp["rotation"]))
            g._opacity = torch.nn.Parameter(f32(inp["opacity"]))
            g.max_radii2D = torch.zeros((P,))
            g.spatial_lr_scale = 1.3
            g.training_setup(targs)
            g.optimizer.param_groups[0]["lr"] = targs.position_lr_init * 1.3
            names = ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"]
            params0 = {n: grp["params"][0] for n, grp in zip(names, g.optimizer.param_groups)}
            grads = []
            for it in range(3):       # three Adam steps with seeded gradients
                gs = {n: rng.normal(0, 1e-2, tuple(p.shape)).astype(np.float32) for n, p in params0.items()}
                if it == 1:
                    gs["opacity"][:] = 0.0            # an all-zero gradient still moves the moments
                grads.append(gs)
                for n, p in params0.items():
                    p.grad = f32(gs[n])
                g.optimizer.step()
                g.optimizer.zero_grad(set_to_none=True)
            after_adam = {n: p.detach().numpy().copy() for n, p in params0.items()}
            # what the rasteriser is fed, and how a gradient on it reaches the raw parameter (autograd of the getter)
            act = g.get_scaling
            up = f32(np.random.default_rng(seed + 100).normal(0, 1, tuple(act.shape)).astype(np.float32))
            (act * up).sum().backward()
            get_scaling, get_scaling_up, get_scaling_grad = (act.detach().numpy().copy(), up.numpy().copy(),
                                                              g._scaling.grad.numpy().copy())
            g._scaling.grad = None
            m_adam = {n: g.optimizer.state[p]["exp_avg"].numpy().copy() for n, p in params0.items()}
            v_adam = {n: g.optimizer.state[p]["exp_avg_sq"].numpy().copy() for n, p in params0.items()}
            # densification statistics: accum / denom with zeros (NaN), large and small ratios
            accum = rng.uniform(0, 1e-3, (P, 1)).astype(np.float32)
            denom = rng.integers(0, 4, (P, 1)).astype(np.float32)
            accum[denom[:, 0] == 0] = 0.0
            g.xyz_gradient_accum = f32(accum)
            g.denom = f32(denom)
            max_grad, extent = 2.5e-4, 4.0
            g.densify_and_prune(max_grad, extent)
            z = torch.cat(dev.draws, 0).numpy() if dev.draws else np.zeros((0, 3), np.float32)
            after = {n: grp["params"][0].detach().numpy().copy() for n, grp in zip(names, g.optimizer.param_groups)}
            m_after = {n: g.optimizer.state[grp["params"][0]]["exp_avg"].numpy().copy()
                       for n, grp in zip(names, g.optimizer.param_groups)}
            v_after = {n: g.optimizer.state[grp["params"][0]]["exp_avg_sq"].numpy().copy()
                       for n, grp in zip(names, g.optimizer.param_groups)}
            step_after = float(g.optimizer.state[g.optimizer.param_groups[0]["params"][0]]["step"])
            g.reset_opacity()
            op_reset = g._opacity.detach().numpy().copy()
            m_op_reset = g.optimizer.state[g._opacity]["exp_avg"].numpy().copy()
        pre = case + "_"
        out[pre + "cfg"] = np.array([scale_lb, alpha_lb, max_grad, extent, targs.percent_dense, 1.3], np.float64)
        for n in names:
            out[pre + "in_" + n] = inp[n]
            for it in range(3):
                out[pre + f"grad{it}_" + n] = grads[it][n]
            out[pre + "adam_" + n] = after_adam[n]
            out[pre + "adam_m_" + n] = m_adam[n]
            out[pre + "adam_v_" + n] = v_adam[n]
            out[pre + "out_" + n] = after[n]
            out[pre + "out_m_" + n] = m_after[n]
            out[pre + "out_v_" + n] = v_after[n]
        out[pre + "accum"], out[pre + "denom"], out[pre + "noise"] = accum, denom, z.astype(np.float32)
        out[pre + "step_after"] = np.array(step_after)
        out[pre + "iso"] = np.array(int(iso))
        out[pre + "get_scaling"], out[pre + "get_scaling_up"], out[pre + "get_scaling_grad"] = (
            get_scaling, get_scaling_up, get_scaling_grad)
        out[pre + "opacity_reset"], out[pre + "opacity_reset_m"] = op_reset, m_op_reset
        print(case, "P", P, "->", after["xyz"].shape[0], "split draws", z.shape[0])
    np.savez_compressed(os.path.join(HERE, "densify_golden.npz"), **out)


if __name__ == "__main__":
    main()
