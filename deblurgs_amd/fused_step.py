"""The device work of one training iteration enqueued straight through the C ABI -- no torch autograd graph, no
elementwise torch glue, no host synchronisation:

    alignment (nu = sort(clamp(cat(0, sigmoid(_nu[view]), 1)))            dgs_alignment_forward
    -> Bezier / se3_exp_map / K cameras                                   dgs_pose_forward
    -> fused K-subframe rasterisation of the cloud's raw parameters       dgs_forward (capacity sized ahead) or the
                                                                          two-phase dgs_forward_geometry / _render
    -> blur image, L1 + temporal-smoothness values and dL/dsubframes      dgs_blur_loss_grad (one pass)
    -> rasteriser backward (+ the opacity-hinge gradient)                 dgs_backward
    -> camera gradients back to the control points and alignment          dgs_pose_backward, dgs_alignment_backward

i.e. scene/motion.py:78-160 + train.py:143-165 of the reference for one blurry view.  The gradients are left in `.grad`
of the cloud's and the motion module's parameters exactly where `loss.backward()` of the autograd path
(CameraMotionModule.query + losses.blur_l1_smooth + losses.hinge_l2) leaves them -- the six per-Gaussian ones as views
of one flat bucket -- so the optimiser step, the sharded all-reduce and the densification statistics that follow are
unchanged.  tests/test_gpu_train.py checks the two paths against each other.

Why: at DeblurGS's real scene sizes (1e4..1e5 Gaussians) the autograd path is host-bound -- ~45 small torch launches,
autograd bookkeeping and one blocking read of num_rendered per step cost more than the kernels.  Here a step is ~12
ctypes calls.

Sizing ahead (`speculative=True`): the duplicate arrays of a step are sized from the duplicate counts observed in earlier
steps OF THE SAME (view, subframe count) (+25 %), which are read from pinned memory only once their copy has completed --
never blocking.  A (view, subframe count) that has no count yet -- a view drawn for the first time, the switch from one
subframe to all of them at curve_start_iter, any view after a densification -- takes the exact two-phase forward (one
host read), so those changes never overflow.  If a step's count still exceeds its capacity the kernels set a device flag
instead of writing out of bounds; the flag makes the optimiser step and the densification statistics of THAT step
no-ops (FusedAdam.skip_flag_ptr), so a truncated gradient is never applied.  Such a step is counted (`dropped`) and
queued (`retry`): TrainingLoop re-runs the view through the exact path as soon as the flag has been read, so every view
still gets its update (the reference applies every step).
"""
import ctypes
import os
import math

import torch

from . import _lib
from . import diff_gaussian_rasterization as dgr


def _ptr(t, offset_elems=0):
    return None if t is None else ctypes.c_void_p(t.data_ptr() + 4 * offset_elems)


class CaptureRefused(RuntimeError):
    """The runtime refused to capture a step into a hipGraph.  Raised only from inside the capture block: nothing of the
    step has been enqueued, so the caller may run the step eagerly instead.  Errors of a cached graph's replay or of the
    eager part behind it are NOT of this kind and propagate as they are."""


class _Pending:
    __slots__ = ("host", "event", "capacity", "speculative", "key", "generation", "request")


class FusedStep:
    def __init__(self, cloud, motion, lambda_hinge=0.0, speculative=True, tile_cull=None):
        if not getattr(cloud, "fused_activations", False):
            raise NotImplementedError("FusedStep needs a cloud with fused_activations")
        self.cloud, self.motion = cloud, motion
        self.lambda_hinge = float(lambda_hinge)
        self.speculative = bool(speculative)
        self.tile_cull = tile_cull
        self.dropped = 0            # steps whose duplicate count exceeded the capacity (their update was skipped)
        self.retry = []             # their (cam_idx, subframe_indice): to be re-run by the caller (pop from here)
        self._seen = {}             # (cam_idx, subframe count) -> recent duplicate counts of that view
        self._generation = 0        # bumped when the cloud changes size: counts of older forwards are not learnt
        self._pending = []          # forwards whose counts have not been read back yet
        self._free_hosts = []
        self._keep = None           # buffers of the last step (the skip flag lives in its geometry blob)
        self.last_capacity = None
        # replay() / replay_front() decline (None: the caller enqueues the step eagerly) when the library would run this
        # view's compositing backward in parts if it were enqueued eagerly (dgs_backward_parts > 1: large views).  Inside a
        # capture the backward is one launch -- a forked executable graph does not return its memory on this runtime --
        # and beside its row totals the eagerly enqueued step is the faster one at these sizes (metric configuration:
        # 10.54-10.62 against 10.76-10.87 ms replayed).  True (TrainingLoop(graph="always")) captures anyway.
        self.capture_large = False
        # the reference composites the depth image with every render (forward.cu:373,390) and query() returns it; the default
        # loss never reads it (train.py:150-153, lambda_depth_tv = 0), so the fused step leaves it out unless asked
        # (run(need_depth=True), lambda_depth_tv > 0) -- or unless this is set (bench.py's value_with_depth region)
        self.always_depth = False
        # "mesh" sharding: the process group of this rank's ROW (the ranks that split one view's subframes); None = the
        # default group ("subframes" mode: all ranks share the view)
        self.loss_group = None
        self.eager_preferred = 0    # steps declined for that reason
        self._sel_cache = {}
        self._side = None           # side stream of the chunked gradient all-reduce
        self.time_allreduce = False
        self.ar_events = []
        self._drops_dev = None      # device word: captured forwards that overflowed so far (DgsForwardOut.drop_counter)
        self._bucket = None         # (generation, numel, tensor): the gradient bucket all captured steps write to
        self._front_shared = None   # (key, radii, screen gradients, skip word) of the captured sharded fronts
        self._graphs = {}           # captured steps by what they bake in (replay)
        self._pool = None           # one memory pool for all of them: replays never overlap
        self._release_pool = False  # a dropped pool's blocks go back to the driver before the next capture
        self._ring, self._ring_pos = [], 0
        self.max_graphs = 256
        self.captured = 0
        self.replayed = 0

    # ------------------------------------------------------------------------------------------ capacity policy
    def _poll(self, block=False):
        still = []
        for pnd in self._pending:
            if block:
                pnd.event.synchronize()
            if pnd.event.query():
                # Every entry owns its pinned words.  An eager step's are written by dgs_forward's own copies; a REPLAYED
                # step's by a copy of the graph's DgsForwardOut.status_dev enqueued right behind that replay (_track_replay)
                # -- the pinned block baked into a graph is rewritten by every later replay of the same view while the host
                # runs several replays ahead, so it can never say which replay overflowed (with two views in
                # flight a running drop counter read from those blocks went backwards and queued 2^32 - 1 retries).
                R, hi, overflow = (int(x) & 0xFFFFFFFF for x in pnd.host[:3].tolist())
                if hi != 0:
                    raise RuntimeError("num_rendered exceeds 32 bits: render fewer subframes per call")
                if pnd.speculative and overflow:
                    self.dropped += 1
                    self.retry.append(pnd.request)
                if pnd.generation == self._generation:
                    self._seen[pnd.key] = (self._seen.get(pnd.key, []) + [R])[-4:]
                self._free_hosts.append(pnd.host)
            else:
                still.append(pnd)
        self._pending = still

    def _track_replay(self, slot, ckey, request, cap, dev):
        """Behind a replay: the step joins the pending list with the pinned slot that its forward's finalize kernel writes
        the count / overflow words into (DgsForwardOut.status_host_indirect: the slot's address travelled to the graph
        through the scalar block, no copy behind the replay)."""
        pnd = _Pending()
        pnd.host = slot
        pnd.speculative, pnd.key, pnd.generation = True, ckey, self._generation
        pnd.request, pnd.capacity = request, cap
        pnd.event = torch.cuda.Event()
        pnd.event.record(torch.cuda.current_stream(dev))
        self._pending.append(pnd)

    def _push_scalars(self, ent, hbuf, dev):
        """This step's scalar block -- and the address of the pinned slot its count words go to (words [4:6]) -- from the
        pinned ring buffer to the graph's device block by a KERNEL (dgs_copy_words): an asynchronous copy in front of every
        graph launch costs a hand-over between the copy engine and the compute queue, tens of microseconds per step."""
        slot = self._host_words()
        hbuf.numpy().view("uint64")[2] = slot.data_ptr()
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(_lib.lib().dgs_copy_words(ctypes.c_void_p(ent["hyper"].data_ptr()), ctypes.c_void_p(hbuf.data_ptr()),
                                             int(hbuf.numel()), stream), "dgs_copy_words")
        return slot

    def _eager_is_faster(self, K, cap, cull):
        if self.capture_large or _lib.context_overlap_mode() == 3:   # (3: the library forks inside a capture too)
            return False
        if _lib.lib().dgs_backward_parts(_lib.context(), int(K), int(cap), int(bool(cull))) <= 1:
            return False
        self.eager_preferred += 1
        return True

    def _capacity(self, key):
        seen = self._seen.get(key)
        if not seen:
            return None
        need = max(seen)
        return need + need // 4 + 16384

    def invalidate(self):
        """The cloud changed (densification, pruning, a restored checkpoint): forget the learnt counts -- also those still
        on their way back -- so that every view takes the exact path once."""
        self._generation += 1
        self._seen = {}
        if self._graphs or self._pool is not None:
            # the captured steps die with the cloud they were captured for, and their memory pool with them: a pool only
            # ever grows (a block freed by one capture is re-used by the next only if it is large enough), so a cloud that
            # grows a little at every densification would otherwise leave one full set of step buffers behind per
            # generation (soak run, 100 k -> 3.4 M Gaussians: 159 GiB instead of 25)
            self._graphs = {}
            self._pool = None
            self._bucket = None
            self._front_shared = None
            self._release_pool = True

    def _host_words(self):
        if self._free_hosts:
            h = self._free_hosts.pop()
            h.zero_()
            return h
        return torch.zeros(8, dtype=torch.int32).pin_memory()

    def _empty_slice(self, gt, lambda_t, K_total, P, H, W, ct_all, cr_all, nu_raw, need_blur, lambda_depth_tv=0.0,
                     ar=None):
        """"subframes" sharding with more ranks than subframes: this rank rasterises nothing but takes part in the loss
        block's exchanges; all its gradients are zero."""
        from . import sharding
        cloud, m = self.cloud, self.motion
        dev = cloud._xyz.device
        f32 = dict(dtype=torch.float32, device=dev)
        color = torch.zeros((0, 3, H, W), **f32)
        _, l1, sm = sharding.subframe_sharded_loss_grad(color, gt.to(dev, torch.float32).contiguous(), K_total,
                                                        float(lambda_t), self.loss_group)
        # The collectives below are issued in EXACTLY the order run() issues them on a rank that holds subframes -- loss
        # block, depth-smoothness value, gradient bucket (whole or in chunks) -- because RCCL pairs collectives by issue
        # order on the communicator: a different order here would pair a one-element all-reduce with a bucket slice.
        depth_tv = None
        if lambda_depth_tv > 0.0:      # this rank's share of the depth-smoothness value is zero
            import torch.distributed as dist
            depth_tv = torch.zeros((), **f32)
            dist.all_reduce(depth_tv, group=self.loss_group)
        if ar is None:
            for p in cloud.hot_parameters():
                p.grad = torch.zeros_like(p)
        else:      # take part in the other ranks' (chunked) reduction of the gradient bucket with a bucket of zeros
            hot = list(cloud.hot_parameters())
            sizes = [p.numel() for p in hot]
            offs = [0]
            for n in sizes:
                offs.append(offs[-1] + (n + 3) // 4 * 4)
            flat = torch.zeros(offs[-1], **f32)
            for i, p in enumerate(hot):
                p.grad = flat[offs[i]:offs[i] + sizes[i]].view(p.shape)
            if int(ar.get("chunks", 1)) <= 1 or P < 512:
                if P > 0:
                    sharding._allreduce(flat, ar.get("average", False), ar.get("group"))
            else:
                widths = [n // max(P, 1) for n in sizes]
                for b0, b1 in sharding.chunk_bounds(P, int(ar["chunks"])):
                    sharding.allreduce_slices([flat[offs[i] + b0 * c:offs[i] + b1 * c] for i, c in enumerate(widths)],
                                              ar.get("average", False), ar.get("group"))
        if m.is_optimizing():
            m._trans._control_points.grad, m._rot._control_points.grad = torch.zeros_like(ct_all), torch.zeros_like(cr_all)
            if nu_raw.numel() > 0:
                m._nu.grad = torch.zeros_like(nu_raw)
        return {"losses": torch.stack([l1.reshape(()), sm.reshape(())]).float(), "blur": None,
                "radii": torch.zeros((0, P), dtype=torch.int32, device=dev), "viewspace_grad": torch.zeros((0, P, 3), **f32),
                "K": K_total, "subframes": color, "depths": torch.zeros((0, 1, H, W), **f32), "skip_flag_ptr": None,
                "skip_flag": None, "depth_tv": depth_tv}

    def _linspace_sel(self, f, n, dev):
        key = (f, n, str(dev))
        if key not in self._sel_cache:
            self._sel_cache[key] = torch.linspace(0, f - 1, n, device=dev).long()
        return self._sel_cache[key]

    def _drop_counter(self, dev):
        if self._drops_dev is None:
            self._drops_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        return self._drops_dev

    # ------------------------------------------------------------------------------------------ captured replay
    HYPER_FLOATS = 64        # [0] lambda_t, [1:4] background, [4:6] address of the step's pinned status slot (one uint64),
                             # [8:8 + 2 * 16] Adam scalars, [48:] alignment jitter (f <= 16)

    def _hyper_words(self, f):
        return (max(self.HYPER_FLOATS, 48 + f) + 1) // 2 * 2        # (even: words [4:6] are written as one uint64)

    def replay(self, cam_idx, lambda_t, gt, subframe_indice, optimizer, tail, signature=(), background=None,
               uniform=None, stats=None):
        """One training iteration as ONE hipGraph launch (SURVEY 8f / DESIGN 2b): alignment -> cameras -> dgs_forward
        (capacity sized ahead) -> loss -> dgs_backward -> camera gradients -> `tail` (densification statistics + the
        optimiser launch).  At DeblurGS's real scene sizes a step is ~60 kernels of a few microseconds each; replayed
        from a graph they cost one launch instead of ~20 ctypes calls and their driver round trips.

        What changes from step to step travels through one small device block written before the launch: lambda_t, the
        random background, the alignment jitter and Adam's step-size scalars (optimizer.step_scalars).  Everything else
        -- the view, the subframe selection, the active SH degree, which parameters are optimised, the capacity of the
        duplicate arrays, every buffer address -- is baked into the graph: one graph per distinct combination, captured
        the first time it is needed (and re-captured when the cloud is rebuilt or a capacity grows).

        Returns the result dict of run(), or None when this step cannot be replayed yet (no duplicate count learnt for
        the view: the caller runs the eager step, which learns it)."""
        cloud, m = self.cloud, self.motion
        dev = cloud._xyz.device
        f = m.n_subframes
        self._poll()
        K_total = f if (isinstance(subframe_indice, str) and subframe_indice == "all") else (
            int(subframe_indice) if isinstance(subframe_indice, int) else len(subframe_indice))
        ckey = (int(cam_idx), K_total, 0)
        cap = self._capacity(ckey) if self.speculative else None
        if cap is None or isinstance(subframe_indice, (list, tuple)) or torch.is_tensor(subframe_indice):
            return None
        q = 1 << max(cap.bit_length() - 5, 10)      # quantised to ~3-6 %: small count drifts do not force a re-capture
        cap = -(-cap // q) * q
        hot = list(cloud.hot_parameters())
        cull = dgr.TILE_CULL if self.tile_cull is None else bool(self.tile_cull)
        if self._eager_is_faster(K_total, cap, cull):
            return None
        gkey = (int(cam_idx), subframe_indice, int(cloud.active_sh_degree), bool(m.is_optimizing()),
                bool(m.curve_random_sample), cap, self._generation, gt.data_ptr(), bool(cull), bool(dgr.WIDE_RECORDS),
                tuple(p.data_ptr() for p in hot), tuple(signature),
                tuple(t.data_ptr() for t in stats) if stats is not None else (), bool(self.always_depth))
        ent = self._graphs.get(gkey)
        if ent is None:
            ent = self._capture(gkey, cam_idx, gt, subframe_indice, cap, optimizer, tail, stats)
        # ---- this step's scalars: host block -> device block (the ring keeps a block alive until its copy has run)
        slot = self._ring[self._ring_pos % len(self._ring)] if self._ring else None
        if slot is None or slot[0].numel() != ent["hyper"].numel():
            self._ring = [(torch.zeros(ent["hyper"].numel(), dtype=torch.float32).pin_memory(), torch.cuda.Event())
                          for _ in range(8)]
            self._ring_pos = 0
            slot = self._ring[0]
        self._ring_pos += 1
        hbuf, hev = slot
        hev.synchronize()                      # (recorded 8 replays ago: never waits in practice)
        hv = hbuf.numpy()
        hv[0] = float(lambda_t)
        hv[1:4] = (torch.rand(3) if background is None else background.detach().float().cpu()).numpy()   # motion.py:112-113
        if m.curve_random_sample and f > 2:                               # scene/motion.py:213-214
            hv[48:48 + f - 2] = (torch.rand(f - 2) if uniform is None else uniform.detach().float().cpu()).numpy()
        for p, g in ent["grads"]:              # the step's gradients live in the graph's buffers
            p.grad = g
        optimizer.skip_flag_ptr = ent["result"]["skip_flag_ptr"]
        optimizer.step_scalars(hv[8:8 + 2 * _lib.ADAM_MAX_GROUPS])
        slot = self._push_scalars(ent, hbuf, dev)
        hev.record(torch.cuda.current_stream(dev))
        ent["graph"].replay()
        self._track_replay(slot, ckey, (cam_idx, subframe_indice), cap, dev)
        self.last_capacity = cap
        self.replayed += 1
        return ent["result"]

    def replay_front(self, cam_idx, lambda_t, gt, subframe_indice, ar, background=None, uniform=None, shard=None,
                     background_dev=None, uniform_dev=None):
        """A SHARDED step ("views" mode) with everything up to its first collective replayed as one hipGraph: alignment ->
        cameras -> dgs_forward (capacity sized ahead) -> loss -> the compositing half of the backward
        (dgs_backward_composite; the whole dgs_backward when the bucket is reduced in one piece) -- then, eagerly, what
        run() does after that point: the per-Gaussian half in Gaussian-index chunks with each chunk's all-reduce behind it
        on the side stream, the camera gradients.  The caller's reductions of the skip flag and the trajectory gradients and
        its optimiser step follow as after run().  Bit-identical to the eager step (the same launches in the same order on
        the same stream); what it saves is the ~20 ctypes calls and driver round trips of the front, 0.2-0.3 ms per step.

        shard = (rank, world): "subframes" sharding.  Its first collective is the loss block, so the captured front ends
        with the forward of this rank's slice of the cameras; the background and the alignment jitter are rank 0's draws and
        arrive as DEVICE tensors (background_dev, uniform_dev: TrainingLoop._shared_draws), copied into the graph's scalar
        block on the stream.  A rank whose slice is empty is not captured (None).

        Returns run()'s result dict ('subframes', 'blur' and 'depths' are None in "views" mode: the images live in the
        graph's pool), or None when the step cannot be replayed yet (no duplicate count learnt for the view)."""
        cloud, m = self.cloud, self.motion
        dev = cloud._xyz.device
        f = m.n_subframes
        self._poll()
        K_total = f if (isinstance(subframe_indice, str) and subframe_indice == "all") else (
            int(subframe_indice) if isinstance(subframe_indice, int) else len(subframe_indice))
        k0 = 0
        if shard is not None:
            from .sharding import shard_range
            k0, k1 = shard_range(K_total, int(shard[0]), int(shard[1]))
            if k1 <= k0:
                return None
        ckey = (int(cam_idx), K_total, k0)
        cap = self._capacity(ckey) if self.speculative else None
        if cap is None or isinstance(subframe_indice, (list, tuple)) or torch.is_tensor(subframe_indice):
            return None
        q = 1 << max(cap.bit_length() - 5, 10)
        cap = -(-cap // q) * q
        hot = list(cloud.hot_parameters())
        cull = dgr.TILE_CULL if self.tile_cull is None else bool(self.tile_cull)
        if shard is None and self._eager_is_faster(K_total, cap, cull):   # ("subframes": the captured front ends before the backward)
            return None
        chunks = 1 if ar is None else int(ar.get("chunks", 1))
        gkey = ("front", int(cam_idx), subframe_indice, int(cloud.active_sh_degree), bool(m.is_optimizing()),
                bool(m.curve_random_sample), cap, self._generation, gt.data_ptr(), bool(cull), bool(dgr.WIDE_RECORDS),
                tuple(p.data_ptr() for p in hot), chunks, float(self.lambda_hinge),
                None if shard is None else (int(shard[0]), int(shard[1])), bool(self.always_depth))
        ent = self._graphs.get(gkey)
        if ent is None:
            ent = self._capture_front(gkey, cam_idx, gt, subframe_indice, cap, ar, K_total, shard)
        slot = self._ring[self._ring_pos % len(self._ring)] if self._ring else None
        if slot is None or slot[0].numel() != ent["hyper"].numel():
            self._ring = [(torch.zeros(ent["hyper"].numel(), dtype=torch.float32).pin_memory(), torch.cuda.Event())
                          for _ in range(8)]
            self._ring_pos = 0
            slot = self._ring[0]
        self._ring_pos += 1
        hbuf, hev = slot
        hev.synchronize()
        hv = hbuf.numpy()
        hv[0] = float(lambda_t)
        hv[1:4] = (torch.rand(3) if background is None else background.detach().float().cpu()).numpy()
        if m.curve_random_sample and f > 2:
            hv[48:48 + f - 2] = (torch.rand(f - 2) if uniform is None else uniform.detach().float().cpu()).numpy()
        slot = self._push_scalars(ent, hbuf, dev)
        hev.record(torch.cuda.current_stream(dev))
        if background_dev is not None:               # (shared draws of a "subframes" step: device to device, stream-ordered)
            ent["hyper"][1:4].copy_(background_dev.reshape(3))
        if uniform_dev is not None and m.curve_random_sample and f > 2:
            ent["hyper"][48:48 + f - 2].copy_(uniform_dev.reshape(f - 2))
        ent["graph"].replay()
        self._track_replay(slot, ckey, (cam_idx, subframe_indice), cap, dev)
        self.last_capacity = cap
        self.replayed += 1
        return ent["finish"](float(lambda_t))

    def _capture_front(self, gkey, cam_idx, gt, subframe_indice, cap, ar, K_total, shard=None):
        cloud, m = self.cloud, self.motion
        dev = cloud._xyz.device
        f = m.n_subframes
        if len(self._graphs) >= self.max_graphs:
            self._graphs.pop(next(iter(self._graphs)))
        hyper = torch.zeros(self._hyper_words(f), dtype=torch.float32, device=dev)
        host = torch.zeros(8, dtype=torch.int32).pin_memory()
        self._drop_counter(dev)
        if self._release_pool:
            self._release_pool = False
            self._keep = None
            torch.cuda.synchronize(dev)
            torch.cuda.empty_cache()
        if self._pool is None:
            self._pool = torch.cuda.graph_pool_handle()
        ent = {"hyper": hyper, "host": host, "status": torch.zeros(4, dtype=torch.int32, device=dev)}
        # What the eager part and the caller read after a replay lives OUTSIDE the capture pool, shared by all graphs of a
        # cloud generation (steps never overlap): the gradient bucket, the loss values, the skip word, and -- sharded runs
        # update the densification statistics in a launch of their own -- the radii and the screen-space gradients.
        Mr_, P_ = cloud._features_rest.shape[1], cloud._xyz.shape[0]
        n_bucket = sum((n + 3) // 4 * 4 for n in (3 * P_, 3 * P_, 3 * Mr_ * P_, P_, 3 * P_, 4 * P_))
        if self._bucket is None or self._bucket[0] != self._generation or self._bucket[1] != n_bucket:
            self._bucket = (self._generation, n_bucket, torch.empty(n_bucket, dtype=torch.float32, device=dev))
        skey = (self._generation, K_total, P_)
        if self._front_shared is None or self._front_shared[0] != skey:
            self._front_shared = (skey, torch.empty((K_total, P_), dtype=torch.int32, device=dev),
                                  torch.empty((K_total, P_, 3), dtype=torch.float32, device=dev),
                                  torch.zeros(1, dtype=torch.int32, device=dev))
        ent["work"] = torch.zeros(8, dtype=torch.float32, device=dev)
        cap_args = {"capacity": cap, "host": host, "status": ent["status"], "lambda_ptr": hyper.data_ptr(),
                    "work": ent["work"], "bucket": self._bucket[2], "split": "composite" if shard is None else "forward",
                    "radii": self._front_shared[1],
                    "means2D": self._front_shared[2], "skipw": self._front_shared[3], "tail": None}
        bg = hyper[1:4]
        uniform = hyper[48:48 + f - 2] if (m.curve_random_sample and f > 2) else None
        for p in list(cloud.hot_parameters()) + (list(m.parameters()) if m.is_optimizing() else []):
            p.grad = None
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(graph, pool=self._pool, capture_error_mode="thread_local"):
                front = self.run(cam_idx, 0.0, gt, bg, subframe_indice, uniform=uniform, _cap=cap_args, ar=ar, shard=shard)
        except RuntimeError as ex:
            raise CaptureRefused(str(ex)) from ex
        # the step's large buffers go back to the pool (the other views' captures re-use the blocks); finish() reaches them
        # through the raw pointers of its DgsProblem / DgsBackwardIO -- valid until the next replay of ANY graph of the pool,
        # which is enqueued after this step's eager part on the same stream
        front["big"][0] = None
        if "fwd_big" in front:
            front["fwd_big"][0] = None
        ent["graph"], ent["finish"], ent["keep"] = graph, front["finish"], front["_keep"]
        self._keep = None
        self._graphs[gkey] = ent
        self.captured += 1
        return ent

    def _capture(self, gkey, cam_idx, gt, subframe_indice, cap, optimizer, tail, stats=None):
        cloud, m = self.cloud, self.motion
        dev = cloud._xyz.device
        f = m.n_subframes
        if len(self._graphs) >= self.max_graphs:      # (views x subframe selections of one run; bounded all the same)
            self._graphs.pop(next(iter(self._graphs)))
        hyper = torch.zeros(self._hyper_words(f), dtype=torch.float32, device=dev)
        host = torch.zeros(8, dtype=torch.int32).pin_memory()
        self._drop_counter(dev)
        if self._release_pool:
            self._release_pool = False
            self._keep = None
            torch.cuda.synchronize(dev)
            torch.cuda.empty_cache()
        if self._pool is None:
            self._pool = torch.cuda.graph_pool_handle()
        ent = {"hyper": hyper, "host": host, "status": torch.zeros(4, dtype=torch.int32, device=dev)}
        # Two things a replay leaves behind live OUTSIDE the capture pool (ordinary allocations made before the capture):
        # the loss kernel's work block -- `losses` handed to the caller is a view of it and holds a replay's two values until
        # the SAME graph is replayed again (in the pool it could be another graph's scratch: all graphs share the pool) --
        # and the gradient bucket, ONE per cloud generation for all graphs (replays never overlap and the optimiser launch
        # at the end of a replay has consumed the gradients; a bucket per graph would be 236 MB per view at P = 1M, SH 3).
        Mr_ = cloud._features_rest.shape[1]
        P_ = cloud._xyz.shape[0]
        n_bucket = sum((n + 3) // 4 * 4 for n in (3 * P_, 3 * P_, 3 * Mr_ * P_, P_, 3 * P_, 4 * P_))
        if self._bucket is None or self._bucket[0] != self._generation or self._bucket[1] != n_bucket:
            self._bucket = (self._generation, n_bucket, torch.empty(n_bucket, dtype=torch.float32, device=dev))
        ent["work"] = torch.zeros(8, dtype=torch.float32, device=dev)
        cap_args = {"capacity": cap, "host": host, "status": ent["status"], "lambda_ptr": hyper.data_ptr(),
                    "work": ent["work"], "bucket": self._bucket[2],
                    "tail": (lambda fr: tail(fr, hyper.data_ptr() + 4 * 8)) if tail is not None else None}
        bg = hyper[1:4]
        uniform = hyper[48:48 + f - 2] if (m.curve_random_sample and f > 2) else None
        params = list(cloud.hot_parameters()) + (list(m.parameters()) if m.is_optimizing() else [])
        optimizer.ensure_state(params)
        for p in params:
            p.grad = None
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, pool=self._pool, capture_error_mode="thread_local"):
            fr = self.run(cam_idx, 0.0, gt, bg, subframe_indice, uniform=uniform, _cap=cap_args, stats=stats)
        # What a replay leaves behind for the host is the two loss values, the count words (pinned) and the gradients of
        # the parameters (the first and the last allocated outside the pool, above); everything else the step allocated
        # is released to the pool here, so that the captures of the other views re-use the same blocks (every replay is a
        # complete step: nothing of one replay is read after the next has started) -- the pool holds ONE step's buffers,
        # not one set per view.
        ent["graph"] = graph
        ent["result"] = {"losses": fr["losses"], "K": fr["K"], "skip_flag_ptr": fr["skip_flag_ptr"], "blur": None,
                         "radii": None, "viewspace_grad": None, "subframes": None, "depths": None, "depth_tv": None}
        ent["grads"] = [(p, p.grad) for p in params if p.grad is not None]
        del fr
        self._keep = None
        self._graphs[gkey] = ent
        self.captured += 1
        return ent

    # ------------------------------------------------------------------------------------------------- the step
    @torch.no_grad()
    def run(self, cam_idx, lambda_t, gt, background, subframe_indice="all", need_blur=False, uniform=None,
            lambda_depth_tv=0.0, shard=None, exact=False, need_depth=False, _cap=None, ar=None, stats=None):
        """gt: [3,H,W] ground truth of view cam_idx (already tone-mapped / noised by the caller); background: [3].
        lambda_depth_tv > 0 adds the reference's optional depth-smoothness term (train.py:150-153,
        utils/loss_utils.py:66-78): its gradient on the K depth images is formed with a few torch ops and handed to the
        rasteriser's backward as dL/ddepth; 'depth_tv' is its value.
        shard=(rank, world): "subframes" sharding (deblurgs_amd.sharding): this rank rasterises subframes
        [floor(rank K / world), floor((rank+1) K / world)) of the view; the loss block runs across the ranks
        (sharding.subframe_sharded_loss_grad: one all-reduce of the partial blur sum, one boundary frame each way), the
        gradients are this rank's PARTIAL sums (the caller adds them over the ranks; the opacity hinge is added on rank 0
        only); 'radii' / 'viewspace_grad' / 'subframes' hold the local slice, 'K' the view's subframe count.
        exact=True forces the two-phase forward (one host read) whatever has been learnt.
        need_depth=True renders the K depth images too ('depths'; implied by lambda_depth_tv > 0), else 'depths' is None.
        ar = {"chunks": G, "average": bool, "group": ...}: a sharded run's all-reduce of the per-Gaussian gradient bucket,
        overlapped with the backward's tail: the per-Gaussian half of the backward runs in G Gaussian-index
        chunks (dgs_backward_geometry) and chunk i is reduced on a side stream while chunk i + 1 computes; the caller must
        then NOT reduce the bucket again (the trajectory gradients remain its job).
        stats = (max_radii2D, xyz_gradient_accum, denom): the cloud's densification accumulators ([P] float32); the
        backward then updates them itself (DgsBackwardIO.stats_*, exactly densify_stats.add_densification_stats_subframes
        on this step's screen gradients) and 'viewspace_grad' comes back None -- the [K,P,3] gradient is never stored.
        _cap: internal, set by replay() while the step is being CAPTURED into a hipGraph -- the capacity, the pinned count
        words, the device words lambda_t is read from and the tail (statistics + optimiser launch) to enqueue; no host
        bookkeeping happens then.
        Returns a dict: 'losses' (device float32 [2]: L1(blur, gt), smoothness -- no host read), 'blur' ([3,H,W] if
        need_blur), 'radii' [K,P] int32, 'viewspace_grad' [K,P,3], 'K', 'skip_flag_ptr' (int or None)."""
        L = _lib.lib()
        cloud, m = self.cloud, self.motion
        dev = cloud._xyz.device
        stream_obj = torch.cuda.current_stream(dev)
        stream = ctypes.c_void_p(stream_obj.cuda_stream)
        f32 = dict(dtype=torch.float32, device=dev)
        f = m.n_subframes
        cam = int(cam_idx)

        # ---- alignment -> nu (all f candidates), then the requested subset
        nu_raw = m._nu
        nu_all = torch.empty(f, **f32)
        src = torch.empty(f, dtype=torch.int32, device=dev)
        nrow = nu_raw.shape[1] if nu_raw.ndim == 2 else 0
        raw_ptr = _ptr(nu_raw, cam * nrow) if nrow > 0 else None
        if uniform is None and m.curve_random_sample and nrow > 0:
            uniform = torch.rand(nrow, **f32)
        if not m.curve_random_sample:
            uniform = None
        _lib.check(L.dgs_alignment_forward(raw_ptr, _ptr(uniform), f, f, _ptr(nu_all), _ptr(src), stream),
                   "dgs_alignment_forward")
        sel = None
        if isinstance(subframe_indice, str) and subframe_indice == "all":
            nu = nu_all
        else:
            if isinstance(subframe_indice, int):     # scene/motion.py:129-131 (1 selects index 0)
                sel = self._linspace_sel(f, subframe_indice, dev)
            else:
                sel = torch.as_tensor(subframe_indice, device=dev).long()
            nu = nu_all[sel].contiguous()
        K = nu.shape[0]

        # ---- cameras
        ct_all, cr_all = m._trans._control_points, m._rot._control_points
        C = ct_all.shape[1] - 1
        row = cam * (C + 1) * 3                       # this view's curve in the translation control points ...
        quat = int(cr_all.shape[-1] == 4)             # ... and in the rotation ones ([C+1,4]: quaternion curve)
        rrow = cam * (C + 1) * cr_all.shape[-1]
        # (stored as a transposed view: made contiguous ONCE -- a 16-float copy kernel per step is a launch and a gap on the
        # device's timeline, 2 % of a cfg2 step)
        pm = m.ref_cam.projection_matrix
        pc = getattr(self, "_proj_cache", None)
        if pc is None or pc[0] is not pm or pc[1] != pm._version or pc[2].device != dev:
            pc = self._proj_cache = (pm, pm._version, pm.to(dev, torch.float32).contiguous().clone())
        proj = pc[2]
        view = torch.empty((K, 4, 4), **f32)
        full = torch.empty((K, 4, 4), **f32)
        campos = torch.empty((K, 3), **f32)
        _lib.check(L.dgs_pose_forward(_ptr(ct_all, row), _ptr(cr_all, rrow), _ptr(nu), _ptr(proj), C, K, quat, _ptr(view),
                                      _ptr(full), _ptr(campos), stream), "dgs_pose_forward")

        K_total, k0 = K, 0
        nu_loc = nu
        if shard is not None:
            from .sharding import shard_range
            k0, k1 = shard_range(K_total, int(shard[0]), int(shard[1]))
            view, full, campos, nu_loc = view[k0:k1], full[k0:k1], campos[k0:k1], nu[k0:k1]   # (contiguous row slices)
            K = k1 - k0
        # ---- forward
        P = cloud._xyz.shape[0]
        H, W = int(m.ref_cam.image_height), int(m.ref_cam.image_width)
        if K == 0:
            return self._empty_slice(gt, lambda_t, K_total, P, H, W, ct_all, cr_all, nu_raw, need_blur, lambda_depth_tv,
                                     ar)
        rest = cloud._features_rest if cloud._features_rest.shape[1] > 0 else None
        Mr = 0 if rest is None else rest.shape[1]
        color = torch.empty((K, 3, H, W), **f32)
        # the depth images are rendered only if somebody reads them (the reference always renders them, and its default
        # loss, lambda_depth_tv = 0, never looks at them: train.py:150-153)
        depth = torch.empty((K, 1, H, W), **f32) if (need_depth or self.always_depth or lambda_depth_tv > 0.0) else None
        radii = (torch.empty((K, P), dtype=torch.int32, device=dev) if (_cap is None or "radii" not in _cap)
                 else _cap["radii"][:K])
        geom = torch.empty(L.dgs_geom_state_bytes(P, K), dtype=torch.uint8, device=dev)
        image = torch.empty(L.dgs_image_state_bytes(W, H, K), dtype=torch.uint8, device=dev)
        bg = background.to(dev, torch.float32).contiguous()
        cull = dgr.TILE_CULL if self.tile_cull is None else bool(self.tile_cull)
        prob = _lib.DgsProblem()
        prob.context = _lib.context(dev.index)
        prob.P, prob.D, prob.M, prob.W, prob.H, prob.K = P, int(cloud.active_sh_degree), 1 + Mr, W, H, K
        prob.tanfovx, prob.tanfovy = math.tan(m.ref_cam.FoVx * 0.5), math.tan(m.ref_cam.FoVy * 0.5)
        prob.scale_modifier, prob.z_near, prob.z_far = 1.0, float(cloud.z_near), float(cloud.z_far)
        prob.use_sigmoid, prob.prefiltered, prob.debug = int(bool(cloud.use_sigmoid)), 0, 0
        prob.tile_cull, prob.scale_lb = int(cull), float(cloud.scale_lower_bound)
        prob.raw_params = 3 if getattr(cloud, "use_isotrophic", False) else 1
        prob.means3D, prob.shs, prob.shs_rest = _ptr(cloud._xyz), _ptr(cloud._features_dc), _ptr(rest)
        prob.opacities, prob.scales, prob.rotations = _ptr(cloud._opacity), _ptr(cloud._scaling), _ptr(cloud._rotation)
        prob.viewmatrix, prob.projmatrix, prob.campos, prob.bg = _ptr(view), _ptr(full), _ptr(campos), _ptr(bg)
        prob.geom_state, prob.geom_bytes = ctypes.c_void_p(geom.data_ptr()), geom.numel()
        prob.image_state, prob.image_bytes = ctypes.c_void_p(image.data_ptr()), image.numel()
        out = _lib.DgsForwardOut()
        out.out_color, out.out_depth, out.radii = _ptr(color), _ptr(depth), ctypes.c_void_p(radii.data_ptr())
        key = (cam, K_total, k0)
        if _cap is None:
            host = self._host_words()
            self._poll()
            cap = self._capacity(key) if (self.speculative and not exact) else None
            pnd = _Pending()
            pnd.host, pnd.speculative, pnd.key, pnd.generation = host, cap is not None, key, self._generation
            pnd.request = (cam_idx, subframe_indice)
        else:
            host, cap, pnd = _cap["host"], int(_cap["capacity"]), None
            out.drop_counter = ctypes.c_void_p(self._drop_counter(dev).data_ptr())
            out.status_dev = ctypes.c_void_p(_cap["status"].data_ptr())
            out.status_host_indirect = ctypes.c_void_p(_cap["lambda_ptr"] + 16)     # words [4:6] of the scalar block
        out.num_rendered_host = ctypes.c_void_p(host.data_ptr())
        if cap is not None:
            binning = torch.empty(L.dgs_binning_state_bytes(cap, W, H, K), dtype=torch.uint8, device=dev)
            prob.binning_state, prob.binning_bytes = ctypes.c_void_p(binning.data_ptr()), binning.numel()
            _lib.check(L.dgs_forward(ctypes.byref(prob), ctypes.byref(out), cap, stream), "dgs_forward")
            R = cap
            skip_ptr = geom.data_ptr() + _lib.layout(P, W, H, K, 0).num_rendered + 20
        else:   # exact two-phase forward: one blocking read of the count (also how the first capacity is learnt)
            _lib.check(L.dgs_forward_geometry(ctypes.byref(prob), ctypes.byref(out), stream), "dgs_forward_geometry")
            stream_obj.synchronize()
            if int(host[1].item()) != 0:
                raise RuntimeError("num_rendered exceeds 32 bits: render fewer subframes per call")
            R = int(host[0].item()) & 0xFFFFFFFF
            binning = torch.empty(L.dgs_binning_state_bytes(R, W, H, K), dtype=torch.uint8, device=dev)
            prob.binning_state, prob.binning_bytes = ctypes.c_void_p(binning.data_ptr()), binning.numel()
            _lib.check(L.dgs_forward_render(ctypes.byref(prob), ctypes.byref(out), R, stream), "dgs_forward_render")
            skip_ptr = None
        if pnd is not None:
            pnd.capacity = R
            pnd.event = torch.cuda.Event()
            pnd.event.record(stream_obj)
            self._pending.append(pnd)
        self.last_capacity = R

        # The forward's three state blobs and the skip word, held through one-element lists: a captured front empties the
        # first after the capture, and what follows reaches them through the raw pointers of prob / io only.
        fwd_big = [(geom, image, binning)]
        fwd_skip = [None]
        if skip_ptr is not None:
            off = skip_ptr - geom.data_ptr()
            fwd_skip[0] = geom[off:off + 4].view(torch.int32)
            if _cap is not None and _cap.get("split") == "forward":     # (the view would pin the whole geometry blob)
                _cap["skipw"].copy_(fwd_skip[0])
                fwd_skip[0] = _cap["skipw"]
        del geom, image, binning

        # Everything after the forward.  Eager steps run it right away.  A "subframes"-sharded step replayed by replay_front
        # captures only what precedes it (its first collective is the loss block) and calls it after every replay; a
        # "views"-sharded one captures on, up to the cut inside (finish()).
        def after_forward(lambda_now=None):
            stream_obj = torch.cuda.current_stream(dev)       # (the stream of THIS call, not of a capture)
            stream = ctypes.c_void_p(stream_obj.cuda_stream)
            lam = float(lambda_t if lambda_now is None else lambda_now)     # (a replayed front hands in this step's weight)
            # ---- loss: blur, both values and dL/dsubframes in one pass (train.py:143-165 image terms)
            gtc = gt.to(dev, torch.float32).contiguous()
            blur = torch.empty((3, H, W), **f32)
            dsub = torch.empty((K, 3, H, W), **f32)
            # dgs_blur_loss_grad's work area: [l1, smooth | accumulators, counter]
            work = torch.empty(8, **f32) if _cap is None else _cap["work"]
            losses = work[:2]
            if _cap is not None and shard is None:      # the scheduled weight is read from device memory when the replayed kernel runs
                _lib.check(L.dgs_blur_loss_grad_dev(_ptr(color), _ptr(gtc), K, 3, H * W, ctypes.c_void_p(_cap["lambda_ptr"]),
                                                    None, _ptr(blur), _ptr(dsub), _ptr(work), stream),
                           "dgs_blur_loss_grad_dev")
            elif shard is None:
                _lib.check(L.dgs_blur_loss_grad(_ptr(color), _ptr(gtc), K, 3, H * W, lam, None, _ptr(blur),
                                                _ptr(dsub), _ptr(work), stream), "dgs_blur_loss_grad")
            else:   # the loss block across the ranks holding the view's other subframes
                from . import sharding
                dsub, l1, sm = sharding.subframe_sharded_loss_grad(color, gtc, K_total, lam, self.loss_group)
                dsub = dsub.contiguous()
                losses = torch.stack([l1.reshape(()), sm.reshape(())]).float()
                blur = None

            # ---- backward: one flat gradient bucket in optimiser-group order (as _RasterizeCloudK.backward)
            sizes = [3 * P, 3 * P, 3 * Mr * P, P, 3 * P, 4 * P]
            offs = [0]
            for n in sizes:
                offs.append(offs[-1] + (n + 3) // 4 * 4)
            flat = torch.empty(offs[-1], **f32) if _cap is None else _cap["bucket"]
            assert flat.numel() == offs[-1]
            seg = lambda i, shape: flat[offs[i]:offs[i] + sizes[i]].view(shape)
            g_xyz, g_dc, g_op, g_sc, g_rot = (seg(0, (P, 3)), seg(1, cloud._features_dc.shape), seg(3, cloud._opacity.shape),
                                              seg(4, (P, 3)), seg(5, (P, 4)))
            g_rest = seg(2, cloud._features_rest.shape)
            g_means2D = None if stats is not None else (torch.empty((K, P, 3), **f32) if (_cap is None or "means2D" not in _cap)
                                                        else _cap["means2D"][:K])
            g_colors = torch.empty((P, 3), **f32)
            g_cov3D = torch.empty((P, 6), **f32)
            g_view, g_proj = torch.empty((K, 4, 4), **f32), torch.empty((K, 4, 4), **f32)
            scratch = torch.empty(L.dgs_backward_scratch_bytes(R, P, K), dtype=torch.uint8, device=dev)
            io = _lib.DgsBackwardIO()
            io.num_rendered = R
            depth_tv, g_depth = None, None
            if lambda_depth_tv > 0.0:
                from . import losses as _losses
                # tv_loss is a mean over the view's K depth images of per-image terms (utils/loss_utils.py:66-78): a rank
                # holding K of the K_total subframes contributes K / K_total of it and needs no other rank's depths
                share = K / float(K_total)
                with torch.enable_grad():
                    dleaf = depth.detach().requires_grad_(True)
                    depth_tv = _losses.tv_loss(dleaf) * share
                    g_depth, = torch.autograd.grad(float(lambda_depth_tv) * depth_tv, dleaf)
                g_depth, depth_tv = g_depth.contiguous(), depth_tv.detach()
                if shard is not None:
                    import torch.distributed as dist
                    dist.all_reduce(depth_tv, group=self.loss_group)   # the value only (logging); the gradient is local
            io.radii, io.dL_dout_color, io.dL_dout_depth = ctypes.c_void_p(radii.data_ptr()), _ptr(dsub), _ptr(g_depth)
            io.scratch, io.scratch_bytes = ctypes.c_void_p(scratch.data_ptr()), scratch.numel()
            io.dL_dmeans3D, io.dL_dmeans2D, io.dL_dsh = _ptr(g_xyz), _ptr(g_means2D), _ptr(g_dc)
            io.dL_dsh_rest = _ptr(g_rest) if Mr > 0 else None
            io.dL_dcolors, io.dL_dopacity, io.dL_dscales, io.dL_drotations = _ptr(g_colors), _ptr(g_op), _ptr(g_sc), _ptr(g_rot)
            io.dL_dcov3D, io.dL_dviewmatrix, io.dL_dprojmatrix = _ptr(g_cov3D), _ptr(g_view), _ptr(g_proj)
            # sharded: the ranks' gradients are summed, so the hinge term is added by one of them only
            io.opacity_hinge_scale = self.lambda_hinge / max(P, 1) if (shard is None or int(shard[0]) == 0) else 0.0
            if stats is not None:
                io.stats_max_radii2D, io.stats_grad_accum, io.stats_denom = (_ptr(stats[0]), _ptr(stats[1]), _ptr(stats[2]))
                io.stats_K_total = int(K_total)
            # ---- the launches up to the first collective ...
            chunked = not (ar is None or int(ar.get("chunks", 1)) <= 1 or P < 512)
            if chunked:
                _lib.check(L.dgs_backward_composite(ctypes.byref(prob), ctypes.byref(io), stream), "dgs_backward_composite")
            else:
                _lib.check(L.dgs_backward(ctypes.byref(prob), ctypes.byref(io), stream), "dgs_backward")
            split = _cap is not None and _cap.get("split") == "composite"
            # the step's large buffers, held through a one-element list: a captured front (replay_front) empties it after the
            # capture so that they go back to the graph pool, and finish() below touches them through prob / io only
            big = [(tuple(fwd_big[0]) if fwd_big[0] is not None else (None, None, None)) +
                   (scratch, color, depth, dsub, blur, g_colors, g_cov3D)]
            skip_flag = fwd_skip[0]
            if split and skip_flag is not None:   # a word of its own, outside the pool (the view would pin the whole geometry blob)
                _cap["skipw"].copy_(skip_flag)
                skip_flag = _cap["skipw"]

            # ---- ... and everything after it (a sharded run's reduction of the bucket, overlapped with the per-Gaussian half
            # of the backward when `chunked`; then the camera gradients).  Eager steps run it right away; a captured front
            # (replay_front) replays the launches above as one hipGraph and calls this after every replay.
            def finish(lambda_now=None):
                # (the stream of THIS call: a captured front was recorded on torch's capture stream, its eager part runs on the
                # caller's stream, behind the replay)
                stream_obj = torch.cuda.current_stream(dev)
                stream = ctypes.c_void_p(stream_obj.cuda_stream)
                if chunked:
                    from . import sharding
                    if self._side is None:
                        self._side = torch.cuda.Stream(device=dev)
                    side = self._side
                    flat.record_stream(side)
                    widths = [3, 3, 3 * Mr, 1, 3, 4]
                    t_ar = None
                    if self.time_allreduce:  # (bench.py: span of the side stream's reductions, first chunk ready -> last done)
                        t_ar = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    for ci, (b0, b1) in enumerate(sharding.chunk_bounds(P, int(ar["chunks"]))):
                        _lib.check(L.dgs_backward_geometry(ctypes.byref(prob), ctypes.byref(io), b0, b1, stream),
                                   "dgs_backward_geometry")
                        ev = torch.cuda.Event()
                        ev.record(stream_obj)
                        with torch.cuda.stream(side):
                            side.wait_event(ev)
                            if t_ar is not None and ci == 0:
                                t_ar[0].record(side)
                            sharding.allreduce_slices([flat[offs[i] + b0 * c:offs[i] + b1 * c] for i, c in enumerate(widths)],
                                                      ar.get("average", False), ar.get("group"))
                    _lib.check(L.dgs_backward_pose(ctypes.byref(prob), ctypes.byref(io), stream), "dgs_backward_pose")
                    done = torch.cuda.Event()
                    done.record(side)
                    if t_ar is not None:
                        t_ar[1].record(side)
                        self.ar_events = (self.ar_events + [t_ar])[-256:]
                    stream_obj.wait_event(done)
                elif ar is not None and P > 0:
                    from . import sharding
                    sharding._allreduce(flat, ar.get("average", False), ar.get("group"))
                if P == 0:
                    flat.zero_()
                    if g_means2D is not None:
                        g_means2D.zero_()
                cloud._xyz.grad, cloud._features_dc.grad, cloud._features_rest.grad = g_xyz, g_dc, g_rest
                cloud._opacity.grad, cloud._scaling.grad, cloud._rotation.grad = g_op, g_sc, g_rot

                # ---- cameras -> control points and alignment (only while the trajectory is being optimised)
                if m.is_optimizing():
                    d_ct_all, d_cr_all = torch.zeros_like(ct_all), torch.zeros_like(cr_all)
                    d_nu = torch.empty(K, **f32)
                    pscratch = torch.empty(L.dgs_pose_scratch_bytes(K), dtype=torch.uint8, device=dev)
                    _lib.check(L.dgs_pose_backward(_ptr(ct_all, row), _ptr(cr_all, rrow), _ptr(nu_loc), _ptr(proj), C, K, quat,
                                                   _ptr(g_view),
                                                   _ptr(g_proj), ctypes.c_void_p(pscratch.data_ptr()), _ptr(d_ct_all, row),
                                                   _ptr(d_cr_all, rrow), _ptr(d_nu), stream), "dgs_pose_backward")
                    m._trans._control_points.grad, m._rot._control_points.grad = d_ct_all, d_cr_all
                    if nrow > 0:
                        d_raw_all = torch.zeros_like(nu_raw)
                        if shard is not None:                    # this rank's slice of the view's subframe times
                            d_all = torch.zeros(K_total, **f32)
                            d_all[k0:k0 + K] = d_nu
                            d_nu = d_all
                        if sel is not None:                      # gradients of the selected subframes back to all f slots
                            d_full = torch.zeros(f, **f32)
                            d_full.index_add_(0, sel, d_nu)
                            d_nu = d_full
                        _lib.check(L.dgs_alignment_backward(raw_ptr, _ptr(uniform), f, f, _ptr(src), _ptr(d_nu),
                                                            _ptr(d_raw_all, cam * nrow), stream), "dgs_alignment_backward")
                        m._nu.grad = d_raw_all
                held = big[0]
                self._keep = (tuple(held[:7]) if held is not None else (None,) * 7) + (view, full, campos, nu, gtc, bg, flat, g_depth)
                return {"losses": losses, "blur": held[7] if (need_blur and held is not None) else None, "radii": radii,
                        "viewspace_grad": g_means2D, "K": K_total, "subframes": held[4] if held is not None else None,
                        "depths": held[5] if held is not None else None, "skip_flag_ptr": skip_ptr, "skip_flag": skip_flag,
                        "depth_tv": depth_tv}

            if split:
                return {"finish": finish, "big": big, "fwd_big": fwd_big,
                        "_keep": (nu_all, src, proj, work, g_view, g_proj, sel, uniform, view, full, campos, nu, gtc, bg)}
            fr = finish()
            if _cap is not None:
                fr["_keep"] = self._keep + (nu_all, src, proj, work, blur, g_colors, g_cov3D, g_view, g_proj, radii,
                                            g_means2D, sel, uniform)
                if _cap.get("tail") is not None:
                    _cap["tail"](fr)
            return fr

        if _cap is not None and _cap.get("split") == "forward":
            return {"finish": after_forward, "big": fwd_big,
                    "_keep": (nu_all, src, proj, view, full, campos, nu, bg, color, depth, radii)}
        return after_forward()
