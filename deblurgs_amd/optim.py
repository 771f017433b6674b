"""Optimiser step and densification of the Gaussian cloud on device.

`FusedAdam` is a torch.optim.Optimizer with torch.optim.Adam's state layout (state[p] = {"step", "exp_avg",
"exp_avg_sq"}, one parameter per named group as the reference builds them, scene/gaussian_model.py:182-193), so
the reference's optimiser surgery (replace_tensor_to_optimizer / _prune_optimizer / cat_tensors_to_optimizer,
scene/gaussian_model.py:301-387) keeps working on it; step() updates every group with ONE kernel
(dgs_adam_step) instead of ~10 elementwise launches per tensor.

`densify_and_prune` restates GaussianModel.densify_and_prune (scene/gaussian_model.py:436-448) as plan + apply
over the whole cloud and both Adam moments (dgs_densify_plan / dgs_densify_apply).
"""
import ctypes

import torch

from . import _lib

FIELDS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")   # group names of the reference


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class FusedAdam(torch.optim.Optimizer):
    # torch.optim.Adam's per-group keys: state_dict()["param_groups"] carries them, so that a checkpoint written with
    # FusedAdam loads into torch.optim.Adam (the reference's restore, scene/gaussian_model.py:97-112) and the other way
    # round.  Only these default values are implemented; step() rejects anything else.
    _ADAM_DEFAULTS = dict(weight_decay=0, amsgrad=False, maximize=False, foreach=None, capturable=False,
                          differentiable=False, fused=None, decoupled_weight_decay=False)

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, clip_value=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, **self._ADAM_DEFAULTS))
        self.clip_value = float(clip_value)   # > 0: fused torch.nn.utils.clip_grad_value_ (train.py:204-205)
        # optional device pointer (int) of a flag word: when non-zero at kernel time the step leaves everything
        # untouched (include/dgs_hip.h, dgs_forward: gradients of a forward whose duplicate capacity overflowed)
        self.skip_flag_ptr = None

    def note_skipped_steps(self, n=1):
        """`n` earlier step() calls turned out to be no-ops on the device (skip_flag_ptr was set when their kernels ran):
        take them out of the per-parameter step counters again, so that state["step"] -- the bias-correction exponent and
        what a checkpoint stores -- equals the number of updates that were applied."""
        for st in self.state.values():
            if "step" in st and float(st["step"]) >= n:
                st["step"] -= n

    def _check_group(self, group):
        for key, default in self._ADAM_DEFAULTS.items():
            v = group.get(key, default)
            if key in ("foreach", "fused"):
                continue                       # implementation selectors of torch.optim.Adam: no effect on the maths
            if v != default and not (key == "weight_decay" and float(v) == 0.0):
                raise RuntimeError(f"FusedAdam implements torch.optim.Adam with {key}={default!r} only (got {v!r})")

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            self._check_group(group)
        batches = {}
        keep = []
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.device.type != "cuda" or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("FusedAdam needs contiguous float32 HIP tensors (no CPU fallback)")
                state = self.state[p]
                if len(state) == 0:
                    state["step"] = torch.tensor(0.0)
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state["step"] += 1
                grad = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                keep.append(grad)
                g = _lib.DgsAdamGroup(p.data_ptr(), grad.data_ptr(), state["exp_avg"].data_ptr(),
                                      state["exp_avg_sq"].data_ptr(), p.numel(), float(group["lr"]),
                                      int(state["step"].item()))
                batches.setdefault((p.device, float(beta1), float(beta2), float(group["eps"])), []).append(g)
        L = _lib.lib()
        for (device, beta1, beta2, eps), gs in batches.items():
            for i in range(0, len(gs), _lib.ADAM_MAX_GROUPS):
                chunk = gs[i:i + _lib.ADAM_MAX_GROUPS]
                arr = (_lib.DgsAdamGroup * len(chunk))(*chunk)
                _lib.check(L.dgs_adam_step(arr, len(chunk), beta1, beta2, eps, self.clip_value,
                                           ctypes.c_void_p(self.skip_flag_ptr) if self.skip_flag_ptr else None,
                                           _stream(device)), "dgs_adam_step")
        return loss

    # ---- captured (hipGraph) steps: the launch is recorded once, the per-step scalars travel through device memory
    def _active(self):
        """[(group, param)] that step() would update now, in launch order; one (beta1, beta2, eps) for all of them."""
        act, hyper = [], None
        for group in self.param_groups:
            self._check_group(group)
            for p in group["params"]:
                if p.grad is None:
                    continue
                h = (float(group["betas"][0]), float(group["betas"][1]), float(group["eps"]))
                if hyper is not None and h != hyper:
                    raise RuntimeError("captured FusedAdam steps need one (betas, eps) for all groups")
                hyper = h
                act.append((group, p))
        if len(act) > _lib.ADAM_MAX_GROUPS:
            raise RuntimeError("captured FusedAdam steps support at most ADAM_MAX_GROUPS parameters")
        return act, hyper

    def _groups_array(self, act, bump):
        gs = []
        for group, p in act:
            state = self.state[p]
            if len(state) == 0:
                state["step"] = torch.tensor(0.0)
                state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            if bump:
                state["step"] += 1
            gs.append(_lib.DgsAdamGroup(p.data_ptr(), p.grad.data_ptr(), state["exp_avg"].data_ptr(),
                                        state["exp_avg_sq"].data_ptr(), p.numel(), float(group["lr"]),
                                        max(int(state["step"].item()), 1)))
        return (_lib.DgsAdamGroup * len(gs))(*gs)

    @torch.no_grad()
    def ensure_state(self, params):
        """Creates the Adam moments of `params` now (a captured step must not allocate them inside its graph)."""
        for p in params:
            state = self.state[p]
            if len(state) == 0:
                state["step"] = torch.tensor(0.0)
                state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)

    @torch.no_grad()
    def signature(self):
        """What a captured step bakes in: which tensors are updated, through which buffers."""
        act, hyper = self._active()
        sig = [hyper, self.clip_value]
        for _, p in act:
            st = self.state.get(p, {})
            sig.append((p.data_ptr(), p.grad.data_ptr(), p.numel(),
                        st["exp_avg"].data_ptr() if "exp_avg" in st else 0))
        return tuple(sig)

    @torch.no_grad()
    def step_scalars(self, out):
        """Host half of a captured step: counts the step (state["step"] += 1) and writes this step's
        (-(lr / (1 - beta1^t)), sqrt(1 - beta2^t)) per active parameter into `out` (float32 numpy view, 2 per
        parameter, dgs_adam_scalars) for the caller to copy into the device block the captured kernel reads."""
        import numpy as np
        act, hyper = self._active()
        arr = self._groups_array(act, bump=True)
        tmp = (ctypes.c_float * (2 * len(act)))()
        _lib.check(_lib.lib().dgs_adam_scalars(arr, len(act), hyper[0], hyper[1], tmp), "dgs_adam_scalars")
        out[:2 * len(act)] = np.frombuffer(tmp, dtype=np.float32)
        return len(act)

    @torch.no_grad()
    def step_enqueue(self, dev_scalars_ptr):
        """Device half: enqueues the update with the scalars read from device memory (capturable; does not count the
        step -- step_scalars does, once per replay)."""
        act, hyper = self._active()
        if not act:
            return
        arr = self._groups_array(act, bump=False)
        _lib.check(_lib.lib().dgs_adam_step_dev(arr, len(act), hyper[0], hyper[1], hyper[2], self.clip_value,
                                                ctypes.c_void_p(self.skip_flag_ptr) if self.skip_flag_ptr else None,
                                                ctypes.c_void_p(dev_scalars_ptr), _stream(act[0][1].device)),
                   "dgs_adam_step_dev")


_pinned_counts = {}


def _counts_buffer(device):
    key = (device.type, device.index)
    if key not in _pinned_counts:
        _pinned_counts[key] = torch.zeros(4, dtype=torch.int32).pin_memory()
    return _pinned_counts[key]


@torch.no_grad()
def densify_plan(xyz_gradient_accum, denom, scaling, opacity, grad_threshold, size_threshold, min_opacity, scale_lb,
                 isotropic=False):
    """Returns (counts [n_keep, n_clone, n_split, m_all], flags u32 [4,P], offsets u32 [4,P]).
    isotropic: get_scaling = activation of column 0 for all three axes (use_isotrophic clouds)."""
    device = scaling.device
    P = scaling.shape[0]
    L = _lib.lib()
    flags = torch.empty((4, P), dtype=torch.int32, device=device)
    offs = torch.empty((4, P), dtype=torch.int32, device=device)
    counts_dev = torch.empty(8, dtype=torch.int32, device=device)
    tmp = torch.empty(L.dgs_densify_tmp_bytes(P), dtype=torch.uint8, device=device)
    host = _counts_buffer(device)
    c = lambda t: t.contiguous().float()
    acc, den, sc, op = c(xyz_gradient_accum), c(denom), c(scaling), c(opacity)
    _lib.check(L.dgs_densify_plan(P, acc.data_ptr(), den.data_ptr(), sc.data_ptr(), op.data_ptr(),
                                  float(grad_threshold), float(size_threshold), float(min_opacity), float(scale_lb),
                                  int(bool(isotropic)), flags.data_ptr(), offs.data_ptr(), counts_dev.data_ptr(), host.data_ptr(),
                                  tmp.data_ptr(), _stream(device)), "dgs_densify_plan")
    torch.cuda.current_stream(device).synchronize()
    return [int(x) & 0xFFFFFFFF for x in host.tolist()], flags, offs


@torch.no_grad()
def densify_apply(counts, flags, offs, params, exp_avgs, exp_avg_sqs, noise, scale_lb, isotropic=False):
    """params: the six raw tensors in FIELDS order; exp_avgs / exp_avg_sqs: their moments (None = no state).
    Returns (new_params, new_exp_avgs, new_exp_avg_sqs) with n_keep + n_clone + 2 n_split rows."""
    device = params[0].device
    P = params[0].shape[0]
    n_keep, n_clone, n_split, m_all = counts
    Pn = n_keep + n_clone + 2 * n_split
    n_rest = params[2][0].numel() if P > 0 else 0
    src, dst = _lib.DgsCloudArrays(), _lib.DgsCloudArrays()
    new_p, new_m, new_v, keep = [], [], [], []
    for f, p in enumerate(params):
        assert p.is_contiguous() and p.dtype == torch.float32 and p.shape[0] == P
        shape = (Pn,) + tuple(p.shape[1:])
        np_, nm, nv = (torch.empty(shape, dtype=torch.float32, device=device) for _ in range(3))
        new_p.append(np_); new_m.append(nm); new_v.append(nv)
        m = None if exp_avgs[f] is None else exp_avgs[f].contiguous()
        v = None if exp_avg_sqs[f] is None else exp_avg_sqs[f].contiguous()
        keep += [m, v]
        src.param[f] = p.data_ptr()
        src.exp_avg[f] = None if m is None else m.data_ptr()
        src.exp_avg_sq[f] = None if v is None else v.data_ptr()
        dst.param[f], dst.exp_avg[f], dst.exp_avg_sq[f] = np_.data_ptr(), nm.data_ptr(), nv.data_ptr()
    if noise is not None:
        noise = noise.contiguous().float()
        assert noise.shape == (2 * m_all, 3)
    carr = (ctypes.c_uint32 * 4)(*counts)
    _lib.check(_lib.lib().dgs_densify_apply(P, n_rest, ctypes.cast(carr, ctypes.c_void_p), flags.data_ptr(),
                                            offs.data_ptr(), ctypes.byref(src), ctypes.byref(dst),
                                            None if noise is None else noise.data_ptr(), float(scale_lb),
                                            int(bool(isotropic)), _stream(device)), "dgs_densify_apply")
    return new_p, new_m, new_v
