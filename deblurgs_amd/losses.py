"""Loss block of the blur-integration loop (train.py:143-165).

Torch expressions mirror utils/loss_utils.py (l1_loss :17-18, tv_loss :66-78, batchwise_smoothness_loss
:80-93, hinge_l2 :96-104) and scene/tonemapping.py; they are pinned by tests/golden/loss_golden.npz.
`blur_l1_smooth` is the fused device path: blur image, both loss values and dL/dsubframes
in one HIP kernel, wrapped as an autograd.Function so that `loss.backward()` feeds the rasteriser directly.
"""
import ctypes

import torch

from . import _lib


def l1_loss(network_output, gt):
    return torch.abs((network_output - gt)).mean()


def l2_loss(network_output, gt):
    return ((network_output - gt) ** 2).mean()


def tv_loss(x: torch.Tensor):
    """x: [b,c,h,w]"""
    horizontal_loss = l2_loss(x[:, :, :-1, :], x[:, :, 1:, :])
    vertical_loss = l2_loss(x[:, :, :, :-1], x[:, :, :, 1:])
    return horizontal_loss + vertical_loss


def batchwise_smoothness_loss(x: torch.Tensor):
    """x: [b,3,h,w] -> L1 between consecutive subframes (zeros(1) when b == 1)."""
    if x.shape[0] == 1:
        return torch.zeros(1, device=x.device)
    return l1_loss(x[1:], x[:-1])


def hinge_l2(x: torch.Tensor):
    """Same values and gradients as the reference's masked assignments (utils/loss_utils.py:96-104), written
    with torch.where so that no boolean-index nonzero() forces a host synchronisation per step."""
    zero = torch.zeros_like(x)
    return (torch.where(x <= 0.0, x ** 2, zero) + torch.where(x >= 1.0, (x - 1.0) ** 2, zero)).mean()


class ToneMapping(torch.nn.Module):
    """scene/tonemapping.py:4-33 (gamma 1/2.2 CRF and its inverse)."""

    def __init__(self, tone_mapping_type: str, eps=1e-8, bound=0):
        super().__init__()
        self.tone_mapping_type = tone_mapping_type
        self.eps = eps
        self.bound = bound

    def forward(self, x):
        if self.tone_mapping_type == "gamma":
            return ((x - self.bound) / (1.0 - 2.0 * self.bound)).clamp_min(self.eps) ** (1 / 2.2)
        elif self.tone_mapping_type == "reverse_gamma":
            return x.clamp_min(self.eps) ** (2.2) * (1.0 - 2.0 * self.bound) + self.bound
        elif self.tone_mapping_type in ["identity", "reverse_identity"]:
            return x
        raise NotImplementedError("Unknown tone mapping type.")

    def inverse(self):
        if "reverse" in self.tone_mapping_type:
            return ToneMapping(self.tone_mapping_type[:8])
        return ToneMapping("reverse_" + self.tone_mapping_type)


def blur_loss_torch(subframes, gt, lambda_t):
    """Reference expression of the image part of the loss: returns (loss, blur, l1, smooth)."""
    blur = subframes.mean(dim=0)
    l1 = l1_loss(blur, gt)
    sm = batchwise_smoothness_loss(subframes)
    return l1 + lambda_t * sm, blur, l1, sm


class _BlurL1Smooth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, subframes, gt, lambda_t):
        if subframes.device.type != "cuda":
            raise RuntimeError("blur_l1_smooth needs device tensors (use blur_loss_torch on CPU)")
        sub = subframes.contiguous().float()
        g = gt.contiguous().float()
        K, C = sub.shape[0], sub.shape[1]
        HW = sub[0, 0].numel()
        blur = torch.empty_like(g)
        work = torch.empty(8, dtype=torch.float32, device=sub.device)     # [l1, smooth | accumulators, counter]
        losses = work[:2]
        st = ctypes.c_void_p(torch.cuda.current_stream(sub.device).cuda_stream)
        _lib.check(_lib.lib().dgs_blur_loss_grad(sub.data_ptr(), g.data_ptr(), K, C, HW, float(lambda_t), None,
                                                 blur.data_ptr(), None, work.data_ptr(), st), "dgs_blur_loss_grad")
        ctx.save_for_backward(sub, g, blur)
        ctx.lambda_t = float(lambda_t)
        total = losses[0] + float(lambda_t) * losses[1]
        ctx.mark_non_differentiable(blur, losses)
        return total, blur, losses

    @staticmethod
    def backward(ctx, g_total, _g_blur, _g_losses):
        sub, g, blur = ctx.saved_tensors
        K, C = sub.shape[0], sub.shape[1]
        HW = sub[0, 0].numel()
        dsub = torch.empty_like(sub)
        up = g_total.detach().float().reshape(1).contiguous()     # device scalar: read by the kernel, no sync
        st = ctypes.c_void_p(torch.cuda.current_stream(sub.device).cuda_stream)
        _lib.check(_lib.lib().dgs_blur_loss_grad(sub.data_ptr(), g.data_ptr(), K, C, HW, ctx.lambda_t, up.data_ptr(),
                                                 blur.data_ptr(), dsub.data_ptr(), None, st), "dgs_blur_loss_grad")
        return dsub, None, None


def blur_l1_smooth(subframes, gt, lambda_t):
    """Fused `L1(mean_k subframes, gt) + lambda_t * L1(subframes[1:] - subframes[:-1])`.
    Returns (loss, blur [C,H,W], losses = [l1, smooth])."""
    return _BlurL1Smooth.apply(subframes, gt, lambda_t)
