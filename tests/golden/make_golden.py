"""Regenerates tests/golden/*.npz by IMPORTING the reference's pure-torch utilities from /root/reference.

Run in the build container only (the reference never travels to the GPU box):
    python tests/golden/make_golden.py
The fixtures are data (seeded inputs + the reference's outputs); no reference source text is stored.
Reference functions exercised:
  utils/pytorch3d_functions.py: se3_exp_map, se3_log_map, so3_exp_map
  utils/sh_utils.py:            eval_sh, RGB2SH
  utils/general_utils.py:       build_rotation (device-patched), get_expon_lr_func, get_scheduler, inverse_sigmoid
  utils/graphics_utils.py:      getProjectionMatrix, getWorld2View2, fov2focal
  utils/loss_utils.py:          l1_loss, batchwise_smoothness_loss, tv_loss, hinge_l2 (+ autograd grads)
  scene/gaussian_activation.py: Clamp, LowerBoundExponent  (loaded by file path)
  scene/tonemapping.py:         ToneMapping("gamma") and its inverse (loaded by file path)
"""
import importlib.util
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, path))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    torch.manual_seed(0)
    rng = np.random.default_rng(0)
    import utils.pytorch3d_functions as t3d
    from utils import sh_utils, general_utils, graphics_utils, loss_utils

    # ---- (1) pose path
    se3 = torch.tensor(rng.normal(0, 0.3, (24, 6)), dtype=torch.float32)
    se3[0] = 0.0                       # exact zero rotation (eps clamp)
    se3[1, 3:] = 1e-4                  # near-zero rotation
    se3[2, 3:] = torch.tensor([2.5, -1.0, 0.3])  # large angle
    se3_64 = se3.double()
    pose = dict(se3=se3.numpy(), exp32=t3d.se3_exp_map(se3).numpy(), exp64=t3d.se3_exp_map(se3_64).numpy(),
                so3_exp32=t3d.so3_exp_map(se3[:, 3:]).numpy(),
                log_of_exp64=t3d.se3_log_map(t3d.se3_exp_map(se3_64)).numpy())
    np.savez_compressed(os.path.join(HERE, "pose_golden.npz"), **pose)

    # ---- (2) SH evaluation (note the reference layout [..., C, coeff])
    P = 64
    dirs = torch.tensor(rng.normal(size=(P, 3)), dtype=torch.float32)
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    sh = torch.tensor(rng.normal(size=(P, 16, 3)), dtype=torch.float32)  # kernel layout [P,M,3]
    sh_out = {f"deg{d}": sh_utils.eval_sh(d, sh.transpose(1, 2), dirs).numpy() for d in range(4)}
    rgb = torch.tensor(rng.random((8, 3)), dtype=torch.float32)
    np.savez_compressed(os.path.join(HERE, "sh_golden.npz"), dirs=dirs.numpy(), sh=sh.numpy(),
                        rgb=rgb.numpy(), rgb2sh=sh_utils.RGB2SH(rgb).numpy(),
                        rgb2sh_sigmoid=sh_utils.RGB2SH(rgb, use_sigmoid=True).numpy(), **sh_out)

    # ---- (3) scale/rotation -> covariance (the reference hard-codes device="cuda"; patch zeros to CPU)
    _zeros = torch.zeros
    torch.zeros = lambda *a, **k: _zeros(*a, **{kk: vv for kk, vv in k.items() if kk != "device"})
    try:
        q = torch.tensor(rng.normal(size=(P, 4)), dtype=torch.float32)
        s = torch.tensor(np.exp(rng.normal(-2, 1, size=(P, 3))), dtype=torch.float32)
        L = general_utils.build_scaling_rotation(1.7 * s, q)
        cov = general_utils.strip_symmetric(L @ L.transpose(1, 2))
        Rm = general_utils.build_rotation(q)
    finally:
        torch.zeros = _zeros
    np.savez_compressed(os.path.join(HERE, "cov3d_golden.npz"), q=q.numpy(), s=s.numpy(), mod=np.float32(1.7),
                        cov6=cov.numpy(), R=Rm.numpy())

    # ---- (4) matrix conventions
    Rw = np.linalg.qr(rng.normal(size=(3, 3)))[0]
    tw = rng.normal(size=3)
    gfx = dict(R=Rw, t=tw, w2v=graphics_utils.getWorld2View2(Rw, tw),
               w2v_ts=graphics_utils.getWorld2View2(Rw, tw, np.array([0.1, -0.2, 0.3]), 1.5),
               proj=graphics_utils.getProjectionMatrix(0.01, 100.0, 1.0471975511965976, 0.6).numpy(),
               proj_args=np.array([0.01, 100.0, 1.0471975511965976, 0.6]),
               fov2focal=np.float64(graphics_utils.fov2focal(1.0471975511965976, 1920)))
    np.savez_compressed(os.path.join(HERE, "graphics_golden.npz"), **gfx)

    # ---- (5) losses, values and autograd grads
    K, h, w = 5, 12, 10
    sub = torch.tensor(rng.random((K, 3, h, w)), dtype=torch.float32, requires_grad=True)
    gt = torch.tensor(rng.random((3, h, w)), dtype=torch.float32)
    # NB train.py:152 feeds tv_loss a 5-D [f,1,1,h,w] tensor (depths are stacked [1,H,W] maps), for which the
    # reference's slicing yields an empty tensor and a NaN loss; lambda_depth_tv defaults to 0 so it never runs.
    # The fixture pins tv_loss on the 4-D input its docstring describes.
    dep = torch.tensor(rng.random((K, h, w)) * 5, dtype=torch.float32, requires_grad=True)
    opa = torch.tensor(rng.normal(0.5, 0.6, (40, 1)), dtype=torch.float32, requires_grad=True)
    blur = sub.mean(0)
    l1 = loss_utils.l1_loss(blur, gt)
    sm = loss_utils.batchwise_smoothness_loss(sub)
    tv = loss_utils.tv_loss(dep[:, None, :, :])
    hg = loss_utils.hinge_l2(opa)
    lam_t, lam_tv, lam_h = 1e-3, 0.05, 0.1
    total = l1 + lam_t * sm + lam_tv * tv + lam_h * hg
    total.backward()
    np.savez_compressed(os.path.join(HERE, "loss_golden.npz"), sub=sub.detach().numpy(), gt=gt.numpy(),
                        dep=dep.detach().numpy(), opa=opa.detach().numpy(), l1=l1.item(), smooth=sm.item(),
                        tv=tv.item(), hinge=hg.item(), total=total.item(), lam=np.array([lam_t, lam_tv, lam_h]),
                        g_sub=sub.grad.numpy(), g_dep=dep.grad.numpy(), g_opa=opa.grad.numpy(),
                        smooth_k1=loss_utils.batchwise_smoothness_loss(sub[:1].detach()).numpy())

    # ---- (6) schedules
    f1 = general_utils.get_expon_lr_func(4e-4, 2e-4, max_steps=25000)
    f2 = general_utils.get_expon_lr_func(1e-3, 1e-5, max_steps=150000)
    steps = np.array([0, 1, 10, 500, 1000, 12500, 25000, 30000, 150000, 200000])
    np.savez_compressed(os.path.join(HERE, "schedule_golden.npz"), steps=steps,
                        densify_threshold=np.array([f1(int(s)) for s in steps]),
                        lambda_t=np.array([f2(int(s)) for s in steps]),
                        inverse_sigmoid_in=np.array([0.1, 0.5, 0.9], np.float32),
                        inverse_sigmoid=general_utils.inverse_sigmoid(torch.tensor([0.1, 0.5, 0.9])).numpy())

    # ---- (7) activations / tone mapping (loaded by path: scene/__init__.py needs plyfile)
    act = _load("scene/gaussian_activation.py", "ref_gaussian_activation")
    tm = _load("scene/tonemapping.py", "ref_tonemapping")
    x = torch.tensor(rng.normal(0.5, 1.0, (50,)), dtype=torch.float32)
    img = torch.tensor(rng.random((3, 6, 5)), dtype=torch.float32)
    gamma = tm.ToneMapping("gamma")
    np.savez_compressed(os.path.join(HERE, "activation_golden.npz"), x=x.numpy(), clamp=act.Clamp()(x).numpy(),
                        lbexp=act.LowerBoundExponent(0.0)(x).numpy(), img=img.numpy(), gamma=gamma(img).numpy(),
                        inv_gamma=gamma.inverse()(img).numpy(),
                        normalize=torch.nn.functional.normalize(x.reshape(10, 5)[:, :4]).numpy())
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
