"""One training iteration of the reference (train.py:104-208) on the fused path: scheduled hyper-parameters ->
CameraMotionModule.query (K subframes, one launch chain) -> fused blur loss + hinge -> backward -> densification
statistics (one kernel) -> densify_and_prune / reset_opacity on their schedule -> one fused Adam launch for the
Gaussians and the trajectory.  This is the caller-side glue of SURVEY 8f rows f1-f3, not a replacement of train.py's
dataset / logging / checkpoint code."""
import types

import torch

from . import losses
from .cloud import get_expon_lr_func, get_scheduler
from .densify_stats import add_densification_stats_subframes


def default_optimization_params(**overrides):
    """arguments/__init__.py:84-123 (OptimizationParams defaults)."""
    d = dict(iterations=150_000, position_lr_init=0.00016, position_lr_final=0.0000016, feature_lr=0.0025,
             opacity_lr=0.05, scaling_lr=0.005, rotation_lr=0.001, percent_dense=0.01, lambda_t_smooth_init=1e-3,
             lambda_t_smooth_final=1e-5, lambda_depth_tv=0.0, lambda_hinge=0.1, densification_interval=200,
             opacity_reset_interval=3000, densify_from_iter=500, densify_until_iter=75_000,
             densify_grad_threshold_init=4e-4, densify_grad_threshold_final=2e-4, densify_annealing_until=25_000,
             clip_grad=-1.0, curve_controlpoints_lr=1e-2, curve_rotation_lr=1e-3, curve_alignment_lr=0.0,
             curve_lr_half_iter=15_000, curve_start_iter=1000, curve_end_iter=100_000, curve_alignment_start=30_000,
             random_sample_until=100_000, noise_init=0.0, noise_final=0.0)
    d.update(overrides)
    return types.SimpleNamespace(**d)


class TrainingLoop:
    def __init__(self, gaussians, cam_motion_module, opt, cameras_extent, white_background=False, spatial_lr_scale=None,
                 distributed=False, tone_mapping=None, fused_step="auto", speculative=True, log_losses=True, graph="auto",
                 ar_chunks=1, graph_min_reuse=8, emulate_shard=None, mesh=None):
        """distributed = "views" (or True): every rank steps on its own view (the caller passes each rank its cam_idx);
        the per-Gaussian AND trajectory gradients are averaged over ranks (one flat all-reduce + one few-KB one) before
        the Adam step and the densification statistics are combined before every densify_and_prune, so the replicas
        (cloud, curves, alignment) stay identical.  This is a G-view mini-batch instead of the reference's one view per
        step (train.py:126).
        distributed = "subframes": every rank gets the SAME cam_idx and rasterises its share of that view's K subframes;
        the loss is formed across ranks (sharding.subframe_sharded_loss_backward: partial blur sum + boundary
        subframes) and the gradients are SUMMED -- exactly the reference's single-view step, K split over the ranks.
        tone_mapping: the scene's ToneMapping (losses.ToneMapping("gamma"), arguments/__init__.py:71); its inverse is
        applied to the ground truth as train.py:141-145 does every iteration (here once per image, cached).  None =
        the ground-truth images are already linear.
        fused_step: "auto" (default) enqueues the iteration's device work through deblurgs_amd.fused_step.FusedStep (no
        autograd graph, no host synchronisation; `speculative` sizes the duplicate arrays ahead, see that module) whenever
        the cloud allows it (fused activations), and falls
        back to the autograd path (CameraMotionModule.query + losses) otherwise; False forces the autograd path.
        ar_chunks > 1 (sharded runs on the fused step): the all-reduce of the per-Gaussian gradient bucket is issued in
        that many Gaussian-index chunks on a side stream, each as soon as the backward has produced it, instead of one
        collective after the backward (FusedStep.run, `ar`).
        graph: "auto" (default) replays the fused iteration -- including the densification statistics and the optimiser
        launch -- as ONE captured hipGraph per (view, subframe selection, SH degree, ...) whenever that is possible
        (single process, no depth-smoothness term, no ground-truth noise, not an iteration that densifies or resets
        opacities), falling back to the eager fused step otherwise; False never captures; "always" captures whenever
        possible.  FusedStep.replay has the details.  "auto" also leaves LARGE views to the eager fused step (those whose
        compositing backward the library runs in parts beside its row totals when it is enqueued eagerly, which it cannot
        do inside a capture: dgs_backward_parts > 1, i.e. tile culling, K >= 6, >= 4 M duplicates): at those sizes the
        host is far ahead of the device and the eager step is the faster one; "always" replays them too.
        A capture costs tens of milliseconds (more for large clouds) and every densification invalidates all of them, so
        while the cloud is still being densified "auto" only captures when a view can expect to be replayed often enough
        before the next densification: densification_interval / number of views >= graph_min_reuse (the reference's
        defaults, 100 iterations and tens of views, stay eager until densify_until_iter and replay afterwards).
        log_losses=False skips forming the scalar "loss" entry of step()'s result (two tiny launches).
        emulate_shard=(rank, world) (measurement only, bench.py --emulate-shard): in a ONE-rank process group, run the
        sharded step of `distributed` mode as rank `rank` of `world` would -- in "subframes" mode its slice of the view's
        subframes (shard_range), in "views" mode the whole view through the sharded code path -- with every collective
        degenerate (one rank): what a rank computes between its exchanges, timed on one GPU.  The loss couples the
        subframes, so the values of an emulated "subframes" step are meaningless; its launches are the real rank's.
        Not carried over from train.py: logging / visualiser / checkpoint-saving calls and `args.flag`."""
        self.gaussians, self.motion, self.opt, self.extent = gaussians, cam_motion_module, opt, cameras_extent
        self.mode = {True: "views", False: None, None: None}.get(distributed, distributed)
        if self.mode not in (None, "views", "subframes", "mesh"):
            raise ValueError("distributed must be False, 'views', 'subframes' or 'mesh'")
        # distributed = "mesh", mesh = (Gv, Gs) (round 6): rank = v * Gs + s; the Gs ranks of row v split the subframes of
        # ONE view (the caller passes every rank of a row the same cam_idx) and exchange the loss block inside the row's own
        # process group; the Gv rows are a mini-batch of views.  Gradients: summed over all ranks, divided by Gv.
        self.mesh = None
        self._row_group, self._row_src = None, 0
        if self.mode == "mesh":
            from . import sharding
            import torch.distributed as dist
            gv, gs = (int(mesh[0]), int(mesh[1]))
            v, s_, grp = sharding.mesh_groups(gv, gs)
            if gs == 1:
                self.mode = "views"
            elif gv == 1:
                self.mode = "subframes"
            else:
                self.mesh = (gv, gs, v, s_)
                self._row_group, self._row_src = grp, v * gs
        # how the gradient bucket is combined: mean over the views of a mini-batch, sum over the slices of one view
        self._avg = {"views": True, "subframes": False, "mesh": float(self.mesh[0]) if self.mesh else False}.get(self.mode, False)
        self.distributed = self.mode is not None
        self._stat_prev = None
        self.white_background = white_background
        self.tone_inverse = None if tone_mapping is None else tone_mapping.inverse()
        self._gt_linear = {}
        gaussians.training_setup(opt, spatial_lr_scale=cameras_extent if spatial_lr_scale is None else spatial_lr_scale)
        cam_motion_module.link_gaussian(gaussians)
        cam_motion_module.add_training_setup(gaussians, {"curve_rot": opt.curve_rotation_lr,
                                                         "curve_trans": opt.curve_controlpoints_lr,
                                                         "curve_alignment": opt.curve_alignment_lr})
        self.densify_threshold_func = get_expon_lr_func(opt.densify_grad_threshold_init, opt.densify_grad_threshold_final,
                                                        max_steps=opt.densify_annealing_until)
        self.lambda_t_smooth_func = get_expon_lr_func(opt.lambda_t_smooth_init, opt.lambda_t_smooth_final,
                                                      max_steps=opt.iterations)
        self.noise_func = get_expon_lr_func(getattr(opt, "noise_init", 0.0), getattr(opt, "noise_final", 0.0),
                                            max_steps=opt.iterations)
        self.alignment_func = get_scheduler(lr_init=opt.curve_alignment_lr, lr_final=1e-7, warmup_ratio=0.0,
                                            step_warmup=getattr(opt, "curve_alignment_start", 30_000),
                                            step_final=opt.iterations)                  # train.py:90-94
        cam_motion_module.alternate_optimization()      # train.py:102, unconditional: curve gradients off at the start
        self.log_losses = log_losses
        self.graph = bool(graph)
        self._graph_always = graph == "always"      # capture whenever possible, whatever the expected re-use
        self.graph_min_reuse = int(graph_min_reuse)
        self.ar_chunks = int(ar_chunks)
        self.fixed_background = None    # a [3] tensor here replaces the random background (scene/motion.py:112-113)
        self.split_noise_fn = None      # f(iteration, m) -> [2 m, 3] standard normals for densify_and_split (tests)
        self.retried = 0            # dropped fused steps that were re-run through the exact path
        self.dist_dropped = 0       # sharded runs: steps every rank dropped together (no make-up; counters corrected)
        self._dist_flags = []
        self._flag_words = []
        self._last_iteration = None
        self._front_failed = False  # the captured front of sharded steps was refused once: stay eager
        self.emulate_shard = None if emulate_shard is None else (int(emulate_shard[0]), int(emulate_shard[1]))
        self._fused = None
        if fused_step:
            try:
                from .fused_step import FusedStep
                if gaussians._xyz.device.type == "cuda":
                    self._fused = FusedStep(gaussians, cam_motion_module, lambda_hinge=max(opt.lambda_hinge, 0.0),
                                            speculative=speculative)
                    self._fused.capture_large = self._graph_always
                    self._fused.loss_group = self._row_group
            except NotImplementedError:
                if fused_step is True:
                    raise

    def _ground_truth(self, cam_idx, gt, iteration):
        """train.py:141-145: gt = tone_mapping.inverse()(gt) + randn * noise(iteration)."""
        if self.tone_inverse is not None:
            if cam_idx not in self._gt_linear:
                self._gt_linear[cam_idx] = self.tone_inverse(gt)
            gt = self._gt_linear[cam_idx]
        noise = self.noise_func(iteration)
        if noise > 0.0:
            gt = gt + torch.randn_like(gt) * noise
        return gt

    def step(self, iteration, cam_idx):
        g, opt = self.gaussians, self.opt
        self._last_iteration = iteration
        g.update_learning_rate(iteration, opt, alignment_lr=self.alignment_func(iteration))     # train.py:109
        densification_threshold = self.densify_threshold_func(iteration)
        lambda_t_smooth = self.lambda_t_smooth_func(iteration)
        if iteration == opt.curve_start_iter or iteration == opt.curve_end_iter:
            self.motion.alternate_optimization()
        if iteration == getattr(opt, "random_sample_until", -1):
            self.motion.curve_random_sample = False                                              # train.py:118-119
        if iteration % 1000 == 0:
            g.oneupSHdegree()
        subframe_indice = "all" if iteration >= opt.curve_start_iter else 1
        # The step's random numbers -- background (scene/motion.py:112-113) and alignment jitter (:213-214) -- are drawn
        # from the HOST generator, once, whichever path runs the step: the eager fused step, the captured one and the
        # autograd path then consume the same stream and can be compared step for step
        self._bg_host = torch.rand(3) if self.fixed_background is None else self.fixed_background.detach().float().cpu()
        m_ = self.motion
        n_jit = m_._nu.shape[1] if (m_.curve_random_sample and m_._nu.ndim == 2) else 0
        self._uni_host = torch.rand(n_jit) if n_jit > 0 else None
        # The opacity hinge is built BEFORE the render: autograd then runs its backward AFTER the rasteriser's, so the
        # rasteriser's gradient (a view of the flat gradient buffer) becomes `_opacity.grad` and the hinge term is added
        # into it in place -- the six gradients stay one contiguous bucket for the all-reduce.
        if self._fused is not None:
            return self._step_fused(iteration, cam_idx, subframe_indice, lambda_t_smooth, densification_threshold)
        L_hinge = losses.hinge_l2(g._opacity) if opt.lambda_hinge > 0.0 else None
        if self.mode in ("subframes", "mesh"):
            return self._step_subframe_sharded(iteration, cam_idx, subframe_indice, lambda_t_smooth, L_hinge,
                                               densification_threshold)
        dev = g._xyz.device
        r = self.motion.query(cam_idx=cam_idx, subframe_indice=subframe_indice, compute_blurred=False,
                              background=self._bg_host.to(dev),
                              uniform=None if self._uni_host is None else self._uni_host.to(dev))
        gt = self._ground_truth(cam_idx, r["gt"], iteration)
        total, blur, lv = losses.blur_l1_smooth(r["subframes"], gt, lambda_t_smooth)
        Ll1, L_t = lv[0], lv[1]
        loss = total
        if L_hinge is not None:
            loss = loss + opt.lambda_hinge * L_hinge
        if opt.lambda_depth_tv > 0.0:
            loss = loss + opt.lambda_depth_tv * losses.tv_loss(r["depths"])
        loss.backward()
        if self.distributed:
            from . import sharding
            # the trajectory parameters are replicated too: their gradients (rows of the views other ranks rendered
            # are zero here) join the reduction, so every replica takes the same curve / alignment step
            sharding.flat_allreduce_grads(g.hot_parameters(), average=True,
                                          extra=[p for p in self.motion.parameters() if p.requires_grad])
        self._tail(iteration, r, densification_threshold)
        return {"loss": loss.detach(), "l1": Ll1, "smooth": L_t, "hinge": L_hinge, "num_points": g._xyz.shape[0]}

    def _shared_draws(self, bg, uniform=None):
        """"subframes" sharding: the ranks rasterise slices of ONE view, so the random background (scene/motion.py:112-113)
        and the alignment jitter (scene/motion.py:213-214, curve_random_sample) must be the same draw on all of them:
        rank 0's, sent in one small broadcast.  Returns (bg [3], uniform [f-2] or None)."""
        import torch.distributed as dist
        m = self.motion
        n = m._nu.shape[1] if (m.curve_random_sample and m._nu.ndim == 2) else 0
        if n > 0 and uniform is None:
            uniform = torch.rand(n).to(bg.device)
        buf = torch.cat([bg.reshape(3).float()] + ([uniform.reshape(n).float()] if n > 0 else []))
        dist.broadcast(buf, src=self._row_src, group=self._row_group)    # ("mesh": the first rank of this view's row)
        return buf[:3].contiguous(), (buf[3:].contiguous() if n > 0 else None)

    def _step_fused(self, iteration, cam_idx, subframe_indice, lambda_t_smooth, densification_threshold, exact=False):
        """The same iteration with its device work enqueued by FusedStep (module docstring there)."""
        g = self.gaussians
        dev = g._xyz.device
        gt = self._ground_truth(cam_idx, self.motion.get_gt_image(cam_idx), iteration)
        bg_host, uni_host = self._bg_host, self._uni_host        # drawn by step()
        if self.graph and not exact and self._graphable(iteration):
            out = self._step_graph(iteration, cam_idx, subframe_indice, lambda_t_smooth, gt, bg_host, uni_host)
            if out is not None:
                return out
        # The draws as device tensors, only where a device tensor is needed (the shared draws of a "subframes" step, the
        # eager step): a copy from PAGEABLE host memory blocks the host until the stream has drained -- twice per step it
        # kept the host from running ahead of a sharded run, whose ~60 eager launches per step then sat on the device's
        # timeline (0.6 ms of an 11.5 ms step, bench.py --emulate-shard) -- so they go through pinned memory, asynchronously.
        def draws_on_device():
            b = bg_host.pin_memory().to(dev, non_blocking=True)
            u = None if uni_host is None else uni_host.pin_memory().to(dev, non_blocking=True)
            return b, u
        bg, uniform = None, None
        shard = None
        if self.distributed:
            self._drain_dist_flags(lag=2)
        if self.mode in ("subframes", "mesh"):
            import torch.distributed as dist
            shard = self.emulate_shard or ((self.mesh[3], self.mesh[1]) if self.mesh else
                                           (dist.get_rank(), dist.get_world_size()))
            bg, uniform = draws_on_device()
            bg, uniform = self._shared_draws(bg, uniform)
        ar = None
        if self.distributed and self.ar_chunks > 1:      # the bucket is reduced inside run(), chunk by chunk
            ar = {"chunks": self.ar_chunks, "average": self._avg}
        # single process: the backward updates the densification statistics itself (no [K,P,3] screen gradient stored);
        # sharded runs keep the separate launch (the statistics are snapshotted before it for their all-reduce)
        stats = None
        if iteration < self.opt.densify_until_iter and not self.distributed:
            stats = (g.max_radii2D, g.xyz_gradient_accum, g.denom)
        fr = None
        if (self.graph and not exact and self.distributed and not self._front_failed and
                self._graphable(iteration, sharded=True)):
            # sharded step: everything up to the first collective as one hipGraph, the reductions and what depends on
            # them eagerly (FusedStep.replay_front); None = no duplicate count learnt for this view yet
            from .fused_step import CaptureRefused
            try:
                fr = self._fused.replay_front(cam_idx, lambda_t_smooth, gt, subframe_indice, ar, background=bg_host,
                                              uniform=uni_host, shard=shard,
                                              background_dev=bg if shard is not None else None,
                                              uniform_dev=uniform if shard is not None else None)
            except CaptureRefused as ex:
                # A capture that the runtime refuses (e.g. a collective library that does not tolerate a capturing
                # stream next to it) must not take an N-rank run down: nothing of the step has been enqueued when a
                # CAPTURE fails, so the eager step below runs it, and the front is not tried again.  Only the capture
                # block raises this (FusedStep._capture_front); an error of a cached graph's replay or of the eager part
                # behind it -- where this rank may already have issued collectives the others are waiting in --
                # propagates.
                import warnings
                warnings.warn(f"captured front disabled for this run (falling back to the eager sharded step): {ex}")
                self._front_failed = True
                fr = None
        if fr is None:
            if bg is None:
                bg, uniform = draws_on_device()
            fr = self._fused.run(cam_idx, lambda_t_smooth, gt, bg, subframe_indice, uniform=uniform,
                                 lambda_depth_tv=max(float(self.opt.lambda_depth_tv), 0.0), shard=shard, exact=exact,
                                 ar=ar, stats=stats)
        skip = fr["skip_flag_ptr"]
        if self.distributed:
            from . import sharding
            import torch.distributed as dist
            if self._fused.speculative:
                # a rank whose duplicate capacity overflowed holds a meaningless gradient: every rank must drop the
                # step (a rank that took the exact path this iteration, or rasterised nothing, brings a zero)
                if fr.get("skip_flag") is not None:
                    flag = fr["skip_flag"]
                else:
                    flag = torch.zeros(1, dtype=torch.int32, device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                self._flag_keep = flag
                skip = flag.data_ptr()
                # Sharded runs do not re-run a dropped step (the replicas would have to agree on the make-up view); they
                # account for it: the reduced flag travels to pinned memory behind the step and is read a FIXED number of
                # iterations later (_drain_dist_flags), on every rank at the same iteration, so that Adam's step counters
                # -- the bias-correction exponent, the value stored in checkpoints -- count applied updates only and stay
                # identical on all replicas.
                h = self._flag_words.pop() if self._flag_words else torch.zeros(1, dtype=torch.int32).pin_memory()
                if flag.is_cuda:     # by a kernel: an asynchronous copy is a hand-over to the copy engine
                    import ctypes
                    from . import _lib
                    _lib.check(_lib.lib().dgs_copy_words(ctypes.c_void_p(h.data_ptr()), ctypes.c_void_p(flag.data_ptr()), 1,
                                                         ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
                               "dgs_copy_words")
                else:
                    h.copy_(flag[:1], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(dev))
                self._dist_flags.append((h, ev, flag))
            self._fused.retry.clear()       # (this rank's own overflows: covered by the reduced flag above)
            # "views": a mini-batch of views, gradients averaged; "subframes": partial sums of one view's gradient
            extra = [p for p in self.motion.parameters() if p.requires_grad]
            if ar is not None:         # (the bucket was reduced inside run(), also by a rank without subframes)
                sharding.allreduce_small_grads(extra, average=self._avg)
            else:
                sharding.flat_allreduce_grads(g.hot_parameters(), average=self._avg, extra=extra)
        g.optimizer.skip_flag_ptr = skip
        r = {"viewspace_points_all": fr["viewspace_grad"], "radii_all": fr["radii"], "K_total": fr["K"],
             "skip_flag_ptr": skip}
        self._tail(iteration, r, densification_threshold, makeup=exact)
        out = {"l1": fr["losses"][0], "smooth": fr["losses"][1], "hinge": None, "num_points": g._xyz.shape[0],
               "loss": None, "dropped": self._fused.dropped, "retried": self.retried}
        # A step whose duplicate capacity overflowed was a no-op on the device; once its flag has come back (usually one
        # step later) the view is re-run here through the exact path, so that no view loses its update.  Single-process
        # only: replicas would have to agree on the make-up step (sharded runs drop the step on every rank instead).
        if not exact and not self.distributed:
            while self._fused.retry:
                cam_r, sub_r = self._fused.retry.pop(0)
                g.optimizer.note_skipped_steps(1)
                self.retried += 1
                self._step_fused(iteration, cam_r, sub_r, lambda_t_smooth, densification_threshold, exact=True)
            out["retried"] = self.retried
        if self.log_losses:
            out["loss"] = fr["losses"][0] + lambda_t_smooth * fr["losses"][1]    # (without the hinge term's value)
            if fr["depth_tv"] is not None:
                out["loss"] = out["loss"] + self.opt.lambda_depth_tv * fr["depth_tv"]
        return out

    def _drain_dist_flags(self, lag=0):
        """Sharded runs: turn the (MAX-reduced, hence rank-independent) drop flags of all but the last `lag` steps into
        step-counter corrections.  Waits for a flag's copy if it has to (with lag = 2 it never does in practice)."""
        while len(self._dist_flags) > lag:
            h, ev, _ = self._dist_flags.pop(0)
            ev.synchronize()
            if int(h[0]) != 0:
                self.dist_dropped += 1
                self.gaussians.optimizer.note_skipped_steps(1)
            self._flag_words.append(h)          # (pinned words are recycled: allocating one costs a driver call)

    def flush(self):
        """Call at the end of training and before writing a checkpoint (on every rank of a sharded run at the same
        iteration): waits for the count / drop words still in flight, makes up for the dropped steps of a single-process
        run (as step() does one iteration late) and settles the step counters of a sharded one."""
        if self._fused is None:
            return
        self._fused._poll(block=True)
        if self.distributed:
            self._drain_dist_flags(lag=0)
            self._fused.retry.clear()
            return
        it = self._last_iteration
        while self._fused.retry and it is not None:
            cam_r, sub_r = self._fused.retry.pop(0)
            self.gaussians.optimizer.note_skipped_steps(1)
            self.retried += 1
            self._step_fused(it, cam_r, sub_r, self.lambda_t_smooth_func(it), self.densify_threshold_func(it), exact=True)

    # ---- the iteration as one captured hipGraph
    def _graphable(self, iteration, sharded=False):
        """sharded=True: the question for a sharded ("views") step, whose captured part ends before the first collective."""
        opt = self.opt
        if not hasattr(self.gaussians.optimizer, "step_enqueue"):
            return False
        if self.distributed != sharded:
            return False
        if opt.lambda_depth_tv > 0.0 or self.noise_func(iteration) > 0.0 or iteration >= opt.iterations:
            return False
        if iteration < opt.densify_until_iter:      # densify_and_prune / reset_opacity go BETWEEN statistics and step
            if iteration > opt.densify_from_iter and iteration % opt.densification_interval == 0:
                return False
            n_views = int(self.motion.gt_images.shape[0])       # captures die with every densification: see __init__
            if not self._graph_always and opt.densification_interval < self.graph_min_reuse * max(n_views, 1):
                return False
            if iteration % opt.opacity_reset_interval == 0 or (self.white_background and
                                                               iteration == opt.densify_from_iter):
                return False
        return True

    def _step_graph(self, iteration, cam_idx, subframe_indice, lambda_t_smooth, gt, bg_host, uni_host):
        g, opt = self.gaussians, self.opt
        stats_on = iteration < opt.densify_until_iter

        stats = (g.max_radii2D, g.xyz_gradient_accum, g.denom) if stats_on else None

        def tail(fr, dev_scalars_ptr):         # recorded into the graph right after the backward (train.py:186-208);
            g.optimizer.skip_flag_ptr = fr["skip_flag_ptr"]       # the statistics are updated by the backward itself
            g.optimizer.step_enqueue(dev_scalars_ptr)

        sig = (stats_on, g.max_radii2D.data_ptr(), g.xyz_gradient_accum.data_ptr(), g.denom.data_ptr(),
               float(self._fused.lambda_hinge), float(g.optimizer.clip_value))
        fr = self._fused.replay(cam_idx, lambda_t_smooth, gt, subframe_indice, g.optimizer, tail, signature=sig,
                                background=bg_host, uniform=uni_host, stats=stats)
        if fr is None:
            return None
        g.optimizer.skip_flag_ptr = None
        out = {"l1": fr["losses"][0], "smooth": fr["losses"][1], "hinge": None, "num_points": g._xyz.shape[0],
               "loss": None, "dropped": self._fused.dropped, "retried": self.retried, "replayed": True}
        if self.log_losses:
            out["loss"] = fr["losses"][0] + lambda_t_smooth * fr["losses"][1]
        while self._fused.retry:               # a replayed step that overflowed is made up for like an eager one
            cam_r, sub_r = self._fused.retry.pop(0)
            g.optimizer.note_skipped_steps(1)
            self.retried += 1
            self._step_fused(iteration, cam_r, sub_r, lambda_t_smooth, self.densify_threshold_func(iteration), exact=True)
        out["retried"] = self.retried
        return out

    def _step_subframe_sharded(self, iteration, cam_idx, subframe_indice, lambda_t_smooth, L_hinge,
                               densification_threshold):
        import torch.distributed as dist
        from . import sharding
        g, opt = self.gaussians, self.opt
        rank, world = self.emulate_shard or ((self.mesh[3], self.mesh[1]) if self.mesh else
                                             (dist.get_rank(), dist.get_world_size()))
        # one view, one background, one alignment jitter: every rank uses rank 0's draws (the fused path does the same)
        bg, uniform = self._shared_draws(self._bg_host.to(g._xyz.device),
                                         None if self._uni_host is None else self._uni_host.to(g._xyz.device))
        r = self.motion.query(cam_idx=cam_idx, subframe_indice=subframe_indice, compute_blurred=False,
                              shard=(rank, world), background=bg, uniform=uniform)
        gt = self._ground_truth(cam_idx, r["gt"], iteration)
        dS, l1, sm = sharding.subframe_sharded_loss_grad(r["subframes"], gt, r["K_total"], lambda_t_smooth, self._row_group)
        roots, seeds = [], []
        if r["subframes"].shape[0] > 0:
            roots.append(r["subframes"])
            seeds.append(dS)
        depth_tv = None
        if opt.lambda_depth_tv > 0.0:
            # tv_loss averages per-image terms over the view's K depth images (utils/loss_utils.py:66-78): this rank's
            # slice contributes k_loc / K of it and needs no other rank's depths; one backward pass for both terms
            depth_tv = torch.zeros((), device=g._xyz.device)
            if r["depths"].shape[0] > 0:
                local = losses.tv_loss(r["depths"]) * (r["depths"].shape[0] / float(r["K_total"]))
                roots.append(opt.lambda_depth_tv * local)
                seeds.append(None)
                depth_tv = local.detach()
            dist.all_reduce(depth_tv, group=self._row_group)
        if roots:
            torch.autograd.backward(roots, seeds)
        l1, sm = float(l1), float(sm)
        if L_hinge is not None:
            # every rank holds the whole cloud: the hinge gradient is added once, on rank 0, and reaches the others
            # through the gradient sum (after the rasteriser's backward, so that it accumulates into the bucket)
            (opt.lambda_hinge * L_hinge * (1.0 if rank == 0 else 0.0)).backward()
        sharding.flat_allreduce_grads(g.hot_parameters(), average=self._avg,
                                      extra=[p for p in self.motion.parameters() if p.requires_grad])
        self._tail(iteration, r, densification_threshold)
        loss = l1 + lambda_t_smooth * sm + (opt.lambda_hinge * float(L_hinge) if L_hinge is not None else 0.0)
        if depth_tv is not None:
            loss = loss + opt.lambda_depth_tv * float(depth_tv)
        return {"loss": torch.tensor(loss), "l1": torch.tensor(l1), "smooth": torch.tensor(sm), "hinge": L_hinge,
                "num_points": g._xyz.shape[0]}

    def _tail(self, iteration, r, densification_threshold, makeup=False):
        """train.py:186-208: densification statistics, densify / reset on their schedule, the optimiser step.
        makeup: the re-run of a dropped step -- statistics and the optimiser step only (the iteration's densification /
        opacity reset already happened)."""
        from . import sharding
        g, opt = self.gaussians, self.opt
        with torch.no_grad():
            if iteration < opt.densify_until_iter:
                if self.distributed and self._stat_prev is None:
                    self._stat_prev = (g.xyz_gradient_accum.clone(), g.denom.clone())
                if r["radii_all"].shape[0] > 0 and r["viewspace_points_all"] is not None:   # (else: done by the backward)
                    add_densification_stats_subframes(r["viewspace_points_all"], r["radii_all"], g.max_radii2D,
                                                      g.xyz_gradient_accum, g.denom, K_total=r["K_total"],
                                                      skip_flag_ptr=r.get("skip_flag_ptr"))
                if makeup:
                    pass
                elif iteration > opt.densify_from_iter and iteration % opt.densification_interval == 0:
                    if self.distributed:
                        sharding.allreduce_densification_stats(g, self._stat_prev)
                        self._stat_prev = None          # densify_and_prune resets the statistics to zeros
                    gen = None
                    if self.distributed:                # the split's normal draws must be the same on every rank
                        gen = torch.Generator(device=g._xyz.device)
                        gen.manual_seed(1_000_003 * int(iteration) + 17)
                    g.densify_and_prune(densification_threshold, self.extent, generator=gen,
                                        noise=None if self.split_noise_fn is None else
                                        (lambda m_sel, _it=iteration: self.split_noise_fn(_it, m_sel)))
                    if self._fused is not None:      # the cloud changed size: learn the duplicate counts afresh
                        self._fused.invalidate()
                    if self.distributed:             # ... and the point-to-point all-reduce's receive blocks are re-sized
                        sharding.release_buffers()
                if not makeup and (iteration % opt.opacity_reset_interval == 0 or
                                   (self.white_background and iteration == opt.densify_from_iter)):
                    g.reset_opacity()
            if iteration < opt.iterations:
                g.optimizer.step()                      # clip_grad_value_ is fused into the step (FusedAdam.clip_value)
                g.optimizer.zero_grad(set_to_none=True)
            if hasattr(g.optimizer, "skip_flag_ptr"):
                g.optimizer.skip_flag_ptr = None

