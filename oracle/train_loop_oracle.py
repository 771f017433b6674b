"""TEST INFRASTRUCTURE ONLY -- the reference's training iteration (train.py:104-222) on the CPU, for a toy deblurring
scene: the stand-in for "PSNR within 0.05 dB of the reference on ExBlur" (ExBlur and the CUDA rasteriser are not
available here).  Nothing under deblurgs_amd/ may import this file.

Every piece is either real torch or a pinned restatement:
  rasteriser + its backward   oracle/torch_naive.py under float32 autograd (dense, every pixel x every Gaussian)
  parameter activations       clamp / exp + lb / normalize / cat as scene/gaussian_model.py:36-50,114-137
  trajectory                  Bezier (scene/bezier.py:54-83) -> oracle/pose_oracle.se3_exp_map -> cameras
                              (scene/motion.py:209-294), differentiable through torch
  loss                        L1(mean_k) + lambda_t * L1(adjacent subframes) + lambda_hinge * hinge_l2(opacity)
                              (train.py:143-165, utils/loss_utils.py)
  optimiser                   torch.optim.Adam(lr=0, eps=1e-15), one group per tensor (scene/gaussian_model.py:182-195)
  densification               statistics as train.py:188-193, densify_and_prune = oracle/train_oracle.py (pinned against the
                              reference's GaussianModel) with the optimiser-state surgery it implies
The schedules (learning rates, lambda_t, densification threshold) are passed in as callables.
"""
import math

import numpy as np
import torch

from . import pose_oracle, torch_naive, train_oracle

FIELDS = train_oracle.FIELDS


def hinge_l2(x):
    zero = torch.zeros_like(x)
    return (torch.where(x <= 0.0, x ** 2, zero) + torch.where(x >= 1.0, (x - 1.0) ** 2, zero)).mean()


def bezier(ctrl, t):
    """ctrl [C+1,d], t [K] -> [K,d]; control point 0 is reached at t = 1 (scene/bezier.py:54-64); float64 coefficients."""
    C = ctrl.shape[0] - 1
    binom = torch.tensor([float(math.comb(C, k)) for k in range(C + 1)], dtype=torch.float64)
    coeff = (t[:, None] ** torch.arange(C, -1, -1)) * ((1 - t)[:, None] ** torch.arange(0, C + 1)) * binom
    return (coeff[:, :, None] * ctrl[None]).sum(dim=1)


def cameras(ctrl_trans, ctrl_rot, nu, proj_T):
    """scene/motion.py:248-294 for K subframes: (world_view [K,4,4], full_proj [K,4,4], campos [K,3]), float32."""
    se3 = torch.cat([bezier(ctrl_trans, nu), bezier(ctrl_rot, nu)], dim=1)
    c2w = pose_oracle.se3_exp_map(se3)
    rots, transes = c2w[:, :3, :3].transpose(-2, -1), c2w[:, 3, :3]
    K = rots.shape[0]
    wv = torch.eye(4, dtype=torch.float32)[None].repeat(K, 1, 1)
    wv[:, :3, :3] = rots.to(torch.float32)
    wv[:, 3, :3] = (-torch.bmm(transes[:, None, :], rots)[:, 0, :]).to(torch.float32)
    full = torch.matmul(wv, proj_T.to(wv)[None])
    campos = torch.linalg.inv(wv)[:, 3, :3]
    return wv, full, campos


class ReferenceTrainer:
    def __init__(self, params, ctrl_trans, ctrl_rot, nu_raw, gt_images, cam, proj_T, opt, extent, lr_funcs, sh_degree,
                 background, z_far=100.0, spatial_lr_scale=1.0):
        """params: dict over FIELDS of float32 arrays (raw parameters); ctrl_* [n,C+1,3]; nu_raw [n,f-2]; gt_images
        [n,3,H,W]; cam: dict(W, H, tanfovx, tanfovy); lr_funcs: dict(xyz=f(it), threshold=f(it), lambda_t=f(it),
        alignment=f(it))."""
        t = lambda a: torch.tensor(np.asarray(a, np.float32), requires_grad=True)
        self.p = {k: t(v) for k, v in params.items()}
        self.ctrl_trans, self.ctrl_rot, self.nu_raw = t(ctrl_trans), t(ctrl_rot), t(nu_raw)
        self.gt = torch.tensor(np.asarray(gt_images, np.float32))
        self.cam, self.proj_T, self.opt, self.extent, self.f = cam, torch.tensor(proj_T), opt, extent, lr_funcs
        self.D, self.bg, self.z_far = sh_degree, torch.tensor(np.asarray(background, np.float32)), z_far
        self.n_sub = self.nu_raw.shape[1] + 2
        o = opt
        lrs = dict(xyz=o.position_lr_init * spatial_lr_scale, f_dc=o.feature_lr, f_rest=o.feature_lr / 20.0,
                   opacity=o.opacity_lr, scaling=o.scaling_lr, rotation=o.rotation_lr)
        groups = [{"params": [self.p[k]], "lr": lrs[k], "name": k} for k in FIELDS]
        groups += [{"params": [self.ctrl_rot], "lr": o.curve_rotation_lr, "name": "curve_rot"},
                   {"params": [self.ctrl_trans], "lr": o.curve_controlpoints_lr, "name": "curve_trans"},
                   {"params": [self.nu_raw], "lr": o.curve_alignment_lr, "name": "curve_alignment"}]
        self.optim = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
        self._reset_stats()
        self.optimizing = False          # train.py:102: curve gradients off at the start
        for q in (self.ctrl_trans, self.ctrl_rot, self.nu_raw):
            q.requires_grad_(False)

    def _reset_stats(self):
        P = self.p["xyz"].shape[0]
        self.accum, self.denom, self.max_radii = np.zeros((P, 1), np.float32), np.zeros((P, 1), np.float32), np.zeros(P, np.float32)

    def _nu(self, cam_idx):
        mid = torch.sigmoid(self.nu_raw[cam_idx])
        return torch.cat([torch.zeros(1), mid, torch.ones(1)]).clamp(0.0, 1.0).sort().values

    def render(self, cam_idx, nu, with_offsets=False):
        p = self.p
        wv, full, campos = cameras(self.ctrl_trans[cam_idx], self.ctrl_rot[cam_idx], nu, self.proj_T)
        sh = torch.cat([p["f_dc"], p["f_rest"]], dim=1)
        op = p["opacity"].clamp(0.0, 1.0)
        sc = torch.exp(p["scaling"])
        rot = torch.nn.functional.normalize(p["rotation"])
        W, H = self.cam["W"], self.cam["H"]
        frames, radii, offs = [], [], []
        for k in range(nu.shape[0]):
            off = torch.zeros((p["xyz"].shape[0], 2), requires_grad=True) if with_offsets else None
            c, _, r = torch_naive.rasterize(p["xyz"], op, wv[k], full[k], campos[k], self.bg, W, H, self.cam["tanfovx"],
                                            self.cam["tanfovy"], z_far=self.z_far, sh=sh, scales=sc, rotations=rot,
                                            sh_degree=self.D, pix_offset=off)
            frames.append(c)
            radii.append(r)
            offs.append(off)
        return torch.stack(frames), radii, offs

    def step(self, iteration, cam_idx, split_noise=None):
        o = self.opt
        for g in self.optim.param_groups:                                     # scene/gaussian_model.py:197-210
            if g["name"] == "xyz":
                g["lr"] = self.f["xyz"](iteration)
            elif g["name"] in ("curve_rot", "curve_trans") and iteration >= o.curve_start_iter:
                g["lr"] = g["lr"] * (0.5) ** (1 / o.curve_lr_half_iter)
            elif g["name"] == "curve_alignment":
                g["lr"] = self.f["alignment"](iteration)
        threshold, lambda_t = self.f["threshold"](iteration), self.f["lambda_t"](iteration)
        if iteration == o.curve_start_iter or iteration == o.curve_end_iter:
            self.optimizing = not self.optimizing
            for q in (self.ctrl_trans, self.ctrl_rot, self.nu_raw):
                q.requires_grad_(self.optimizing)
        nu = self._nu(cam_idx)
        if iteration < o.curve_start_iter:
            nu = nu[torch.linspace(0, nu.shape[0] - 1, 1).long()]              # scene/motion.py:129-131: index 0
        sub, radii, offs = self.render(cam_idx, nu, with_offsets=True)
        K = sub.shape[0]
        blur = sub.mean(dim=0)
        l1 = (blur - self.gt[cam_idx]).abs().mean()
        sm = (sub[1:] - sub[:-1]).abs().mean() if K > 1 else torch.zeros(())
        loss = l1 + lambda_t * sm + o.lambda_hinge * hinge_l2(self.p["opacity"])
        self.optim.zero_grad(set_to_none=True)
        loss.backward()
        if iteration < o.densify_until_iter:                                    # train.py:186-201
            W, H = self.cam["W"], self.cam["H"]
            for k in range(K):
                vis = radii[k].numpy() > 0
                g2 = offs[k].grad.numpy() * np.array([0.5 * W, 0.5 * H], np.float32)   # NDC-scaled (backward.cu:535)
                self.max_radii[vis] = np.maximum(self.max_radii[vis], radii[k].numpy()[vis].astype(np.float32))
                self.accum[vis] += np.linalg.norm(g2[vis], axis=1, keepdims=True).astype(np.float32)
                self.denom[vis] += np.float32(1.0 / K)
            if iteration > o.densify_from_iter and iteration % o.densification_interval == 0:
                self._densify(threshold, split_noise)
        if iteration < o.iterations:
            self.optim.step()
        return {"l1": float(l1.detach()), "smooth": float(sm.detach()), "num_points": self.p["xyz"].shape[0]}

    def _densify(self, threshold, split_noise):
        st = self.optim.state
        have = all(len(st[self.p[k]]) > 0 for k in FIELDS)
        cur = {k: self.p[k].detach().numpy() for k in FIELDS}
        m = {k: st[self.p[k]]["exp_avg"].numpy() for k in FIELDS} if have else None
        v = {k: st[self.p[k]]["exp_avg_sq"].numpy() for k in FIELDS} if have else None
        # the split-selected Gaussians, in the reference's order, to size the normal draws (scene/gaussian_model.py:394-399)
        with np.errstate(all="ignore"):
            grads = np.nan_to_num((self.accum / self.denom).reshape(-1), nan=0.0)
        scal = np.exp(cur["scaling"])
        n_clone = int(((np.abs(grads) >= np.float32(threshold)) &
                       (scal.max(1) <= np.float32(self.opt.percent_dense * self.extent))).sum())
        n_split = int(((grads >= np.float32(threshold)) & (scal.max(1) > np.float32(self.opt.percent_dense * self.extent))).sum())
        noise = split_noise(n_split) if callable(split_noise) else np.random.default_rng(0).normal(size=(2 * n_split, 3))
        newp, newm, newv = train_oracle.densify_and_prune(cur, m, v, self.accum, self.denom, threshold, self.extent,
                                                          self.opt.percent_dense, np.asarray(noise, np.float32))
        for g in self.optim.param_groups:
            k = g["name"]
            if k not in FIELDS:
                continue
            old = self.p[k]
            new = torch.tensor(newp[k], requires_grad=True)
            state = st.pop(old, None)
            g["params"][0] = new
            if have and state is not None:
                state["exp_avg"], state["exp_avg_sq"] = torch.tensor(newm[k]), torch.tensor(newv[k])
                st[new] = state
            self.p[k] = new
        self._reset_stats()
        return n_clone, n_split


def psnr(a, b):
    mse = float(((a - b) ** 2).mean())
    return 10.0 * math.log10(1.0 / max(mse, 1e-20))
