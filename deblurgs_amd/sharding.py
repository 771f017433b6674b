"""Sharding of the blur-integration loop over the GPUs of one node (new work: the reference is single-GPU,
SURVEY.md 0.4 / 8e).  One process per GPU, `torch.distributed` with backend "nccl" (= RCCL over xGMI) on
device tensors and "gloo" on CPU (tests).

Two modes, both with all Gaussian and trajectory parameters replicated on every rank:

  "views"      rank g renders all K subframes of ITS OWN blurry view (one whole-view fused launch chain per
               rank: weak scaling, no data-path collective) and the per-Gaussian gradients are summed with ONE
               flat all-reduce per step.  This turns the reference's one-view-per-step SGD (train.py:126) into
               a G-view mini-batch; gradients are averaged over the G views.
  "subframes"  the K subframes of ONE view are split over ranks (rank g gets k in [floor(gK/G), floor((g+1)K/G)));
               the loss couples them (mean over K, adjacent-k smoothness), so before the backward there is one
               all-reduce of the partial blur sum [3,H,W] and a point-to-point neighbour exchange of one boundary
               subframe each way (batch_isend_irecv; which rank holds which subframes is computed on the host);
               after it the same flat gradient all-reduce (per-Gaussian + curve parameters).  Semantics are
               exactly the reference's single-view step.

  "mesh"       (round 6) Gv x Gs ranks, rank = v * Gs + s: the Gs ranks of row v split the K subframes of view v like
               "subframes" (their loss exchange runs inside the row's own process group), the Gv rows are a mini-batch
               like "views"; the gradient bucket is SUMMED over all ranks and divided by Gv (the sum over a row is one
               view's gradient, the mean runs over the views).  Gs = 1 is "views", Gv = 1 is "subframes".

Collective choice: one bucket of P*(11+3M) floats (152 MB at P=1M, M=9) per step, so RCCL can use
every xGMI link at once; nothing is reduced per tensor.
"""
import os

import torch
import torch.distributed as dist


# DGS_DIST_FORCE_COLLECTIVES=1: issue every collective of the product path even in a one-rank group (values unchanged).
# tools/rccl_smoke.py uses it to put the real RCCL library, init path and reduction ops under a TrainingLoop step on a
# one-GPU box.
FORCE_COLLECTIVES = os.environ.get("DGS_DIST_FORCE_COLLECTIVES", "0") == "1"


def _init_timeout():
    """Rendezvous / collective timeout: a rank that died must not leave the others waiting for torch's 10-minute default
    (DGS_DIST_TIMEOUT_S overrides the 120 s)."""
    import datetime
    return datetime.timedelta(seconds=float(os.environ.get("DGS_DIST_TIMEOUT_S", "120")))


def init_distributed(device_type="cuda"):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment.  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or os.environ.get("DGS_DIST_FORCE_INIT", "0") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # DGS_DIST_BACKEND=gloo and DGS_DIST_ONE_DEVICE=1 let the N-rank code path run on a box with ONE GPU
        # (all ranks on cuda:0, collectives staged through the host): a functional check, not a measurement
        backend = os.environ.get("DGS_DIST_BACKEND", "nccl" if device_type == "cuda" else "gloo")
        if os.environ.get("DGS_DIST_ONE_DEVICE", "0") == "1":
            local_rank = 0
        if device_type == "cuda":
            torch.cuda.set_device(local_rank)
            if backend == "nccl":
                dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=_init_timeout(),
                                        device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=_init_timeout())
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=_init_timeout())
    return rank, world, local_rank


def shard_range(K, rank, world):
    """Subframes of rank g: [floor(gK/G), floor((g+1)K/G)) -- K=15, G=8 gives seven ranks x 2 and one x 1."""
    return (rank * K) // world, ((rank + 1) * K) // world


def _shared_flat(grads):
    """If the tensors are contiguous fp32 views of ONE storage that is (almost) covered by them -- the layout
    _RasterizeCloudK.backward produces -- return the 1-D tensor spanning them, else None.  Alignment padding
    between the segments is reduced along with the data (it is never read)."""
    grads = [g for g in grads if g is not None and g.numel() > 0]
    if len(grads) == 0 or any(g.dtype != torch.float32 or not g.is_contiguous() for g in grads):
        return None
    st = grads[0].untyped_storage()
    if any(g.untyped_storage().data_ptr() != st.data_ptr() for g in grads[1:]):
        return None
    lo = min(g.storage_offset() for g in grads)
    hi = max(g.storage_offset() + g.numel() for g in grads)
    if hi - lo > sum(g.numel() for g in grads) + 4 * len(grads):
        return None
    return torch.empty(0, dtype=torch.float32, device=grads[0].device).set_(st, lo, (hi - lo,))


def _adopt_into_bucket(params):
    """The fused operator returns the six gradients as views of one bucket laid out in parameter order with 16-byte
    aligned segments, but a parameter that also receives gradient from elsewhere (the opacity hinge) ends up with a
    tensor of its own, because autograd sums the two contributions out of place.  If all other gradients still sit
    in such a bucket, copy the stray ones into their (unused) slots and re-bind .grad to the slot views, so that the
    whole bucket can be reduced in place.  Returns the bucket or None."""
    grads = [p.grad for p in params]
    if any(g is None or g.dtype != torch.float32 or not g.is_contiguous() for g in grads):
        return None
    st = grads[0].untyped_storage()
    off = grads[0].storage_offset()
    slots = []
    for g in grads:
        slots.append(off)
        off += (g.numel() + 3) // 4 * 4
    last_end = slots[-1] + grads[-1].numel()      # the operator does not pad after the final segment
    if st.nbytes() < 4 * last_end:
        return None
    inside = [g.untyped_storage().data_ptr() == st.data_ptr() and g.storage_offset() == o for g, o in zip(grads, slots)]
    if sum(g.numel() for g, i in zip(grads, inside) if i) < 0.5 * sum(g.numel() for g in grads):
        return None
    bucket = torch.empty(0, dtype=torch.float32, device=grads[0].device).set_(st, slots[0], (last_end - slots[0],))
    for p, g, o, i in zip(params, grads, slots, inside):
        if not i and g.numel() > 0:
            view = bucket[o - slots[0]:o - slots[0] + g.numel()].view_as(g)
            view.copy_(g)
            p.grad = view
    return bucket


# How large gradient buffers are summed over the ranks: "collective" = the backend's all-reduce (RCCL picks ring / tree);
# "p2p" = a direct reduce-scatter + all-gather over point-to-point sends (SURVEY 8e's fallback: on MI355X every GPU pair
# has its own xGMI link, so rank r can receive its shard from all G-1 peers at once -- seven links busy instead of the
# two a ring uses; choose it with DGS_DIST_ALLREDUCE=p2p, TrainingLoop / bench.py --allreduce p2p, if the scaling run shows
# RCCL ringing).  Small buffers always take the collective.
ALLREDUCE_MODE = os.environ.get("DGS_DIST_ALLREDUCE", "collective")
# buffers below this many elements always take the collective (DGS_DIST_P2P_MIN_NUMEL=0: every buffer takes the p2p path,
# which is how the N-rank tests exercise it on small scenes)
P2P_MIN_NUMEL = int(os.environ.get("DGS_DIST_P2P_MIN_NUMEL", str(1 << 16)))

_pinned = {}      # (slot, numel, dtype) -> pinned host staging buffer of _p2p's non-RCCL path


def _stage(slot, t):
    key = (slot, t.numel(), t.dtype)
    h = _pinned.get(key)
    if h is None:
        if len(_pinned) > 256:
            _pinned.clear()
        h = _pinned[key] = torch.empty(t.numel(), dtype=t.dtype).pin_memory()
    return h


def _p2p(sends, recvs, group=None):
    """One batch of point-to-point transfers: sends / recvs = [(contiguous tensor, peer rank within `group`)].  Blocks the
    HOST until the batch is done on every backend but RCCL.

    Backend "nccl" (= RCCL): one batch_isend_irecv = one RCCL group, ordered on the CURRENT stream like any kernel: a
    send reads what earlier work on this stream wrote, later work sees the received bytes; nothing waits on the host.

    Any other backend (gloo: the CPU tests, and the N-rank functional tests with two ranks on one GPU): such a backend
    moves bytes from the HOST thread through tensor.data_ptr(), which knows nothing about streams -- handed a device
    tensor it would read the buffer before the kernels (or the async copy_) producing it have run, and a receive could
    land while a kernel still reads the old contents.  So the current stream is drained first, device tensors travel
    through pinned host buffers (the backend never sees a device pointer), and the received bytes are copied back with
    blocking copies."""
    peer = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
    if not sends and not recvs:
        return
    if dist.get_backend(group) == "nccl":
        ops = [dist.P2POp(dist.isend, t, peer(r), group) for t, r in sends]
        ops += [dist.P2POp(dist.irecv, t, peer(r), group) for t, r in recvs]
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        return
    dev = next((t.device for t, _ in list(sends) + list(recvs) if t.is_cuda), None)
    if dev is not None:
        torch.cuda.current_stream(dev).synchronize()
    h_send, h_recv = [], []
    for i, (t, r) in enumerate(sends):
        if t.is_cuda:
            h = _stage(("s", i), t)
            h.copy_(t.reshape(-1))                 # blocking device -> pinned copy (the stream is drained)
        else:
            h = t
        h_send.append((h, r))
    for i, (t, r) in enumerate(recvs):
        h_recv.append((_stage(("r", i), t) if t.is_cuda else t, r))
    reqs = [dist.isend(h, peer(r), group=group) for h, r in h_send]
    reqs += [dist.irecv(h, peer(r), group=group) for h, r in h_recv]
    for req in reqs:
        req.wait()
    for (t, _), (h, _) in zip(recvs, h_recv):
        if t.is_cuda:
            t.reshape(-1).copy_(h)                 # blocking pinned -> device copy
    if dev is not None:
        torch.cuda.current_stream(dev).synchronize()


def _divisor(average, group=None):
    """`average` of the reductions below: False = sum, True = mean over the group, a number = sum divided by it (the
    "mesh" mode's mean over the Gv views of a sum over all Gv * Gs ranks)."""
    if average is True:
        return float(dist.get_world_size(group))
    if average is False or average is None:
        return 1.0
    return float(average)


def mesh_groups(views, subframes):
    """Process groups of a Gv x Gs mesh over the default group (rank = v * Gs + s): EVERY rank creates every row's group,
    in the same order (torch.distributed's rule).  Returns (v, s, row_group) of the calling rank; Gs == 1 -> row_group None
    (a row of one rank has nothing to exchange)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    if views * subframes != world:
        raise ValueError(f"mesh {views} x {subframes} needs {views * subframes} ranks, the process group has {world}")
    v, s = divmod(rank, subframes)
    mine = None
    if subframes > 1:
        for row in range(views):
            g = dist.new_group(ranks=list(range(row * subframes, (row + 1) * subframes)))
            if row == v:
                mine = g
    return v, s, mine


def _rank_ordered_sum_(recv, own, rank, world, average):
    """own <- sum over the ranks of (own on row `rank`, recv[s] elsewhere), added in rank order, / world if average
    (/ the number given if `average` is one: _divisor).
    Device tensors: ONE launch (dgs_rank_ordered_sum); CPU tensors (the gloo tests): the same arithmetic in torch."""
    if own.is_cuda:
        import ctypes
        from . import _lib
        stream = ctypes.c_void_p(torch.cuda.current_stream(own.device).cuda_stream)
        _lib.check(_lib.lib().dgs_rank_ordered_sum(ctypes.c_void_p(recv.data_ptr()), recv.stride(0),
                                                   ctypes.c_void_p(own.data_ptr()), own.numel(), world, rank,
                                                   float(world) if average is True else _divisor(average), stream),
                   "dgs_rank_ordered_sum")
        return
    n = own.numel()
    acc = (own if rank == 0 else recv[0, :n]).clone()
    for s in range(1, world):
        acc += own if s == rank else recv[s, :n]
    if average:
        acc /= (world if average is True else _divisor(average))
    own.copy_(acc)


_recv_blocks = {}     # (device, stream, numel) -> receive block of p2p_allreduce_multi_ (one per stream: calls on a stream are ordered)


def _recv_block(device, numel):
    key = (str(device), torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0)
    buf = _recv_blocks.get(key)
    # grown when too small, and given back when the request has dropped well below it (a prune shrinks the bucket; a
    # cached block of the old size would otherwise stay for the life of the process -- ADVICE r5)
    if buf is None or buf.numel() < numel or buf.numel() > 2 * max(numel, 1) + 4096:
        buf = _recv_blocks[key] = torch.empty(max(numel, 1), dtype=torch.float32, device=device)
    return buf


def release_buffers():
    """Drops the cached receive blocks of the point-to-point all-reduce (TrainingLoop calls this after a densification:
    the bucket has a new size, and the chunked path's per-chunk sizes change with it)."""
    _recv_blocks.clear()


def p2p_allreduce_multi_(tensors, average=False, group=None, align=256, force=False):
    """In-place sum (or mean) of several 1-D contiguous fp32 tensors over the ranks without a collective: every tensor is
    cut into one shard per rank (boundaries on multiples of `align` elements); rank r receives its shards from every peer
    -- ONE batch of point-to-point operations = one RCCL group for all tensors --, adds the G contributions IN RANK ORDER
    (one fused launch per tensor, dgs_rank_ordered_sum: the result is a fixed function of the inputs and bit-identical on
    every rank), and sends the reduced shards back to every peer (a second batch).  2 (G-1)/G of the data leave and enter
    each rank, as in a ring, but over G-1 links at once.  The receive block is kept per stream (no allocation per call).
    force: in a ONE-rank group, run both phases as a send to / receive from this rank itself (RCCL accepts a self send +
    receive inside one group; values unchanged) -- tools/rccl_smoke.py puts the RCCL point-to-point path under the product
    code on a one-GPU box this way."""
    W = dist.get_world_size(group)
    tensors = [t for t in tensors if t.numel() > 0]
    if not tensors or (W == 1 and not force):
        return
    r = dist.get_rank(group)
    if W == 1:      # (force) every tensor goes through a self send / receive and comes back as it was, twice
        for _phase in range(2):
            tmps = [torch.empty_like(t) for t in tensors]
            _p2p([(t, r) for t in tensors], [(tmp, r) for tmp in tmps], group)
            for t, tmp in zip(tensors, tmps):
                t.copy_(tmp)
        return
    plans = []
    for t in tensors:
        n = t.numel()
        per = -(-n // W)
        per = -(-per // align) * align
        plans.append([(min(i * per, n), min((i + 1) * per, n)) for i in range(W)])
    mines = [b[r][1] - b[r][0] for b in plans]
    block = _recv_block(tensors[0].device, W * sum(-(-m // 4) * 4 for m in mines))
    recvs_of, o = [], 0
    for m in mines:
        m4 = -(-m // 4) * 4                   # (rows start on 16-byte boundaries: float4 accesses in the sum kernel)
        recvs_of.append(block[o:o + W * m4].view(W, m4))
        o += W * m4
    sends, recvs = [], []
    for t, bounds, mine, rv in zip(tensors, plans, mines, recvs_of):
        for s in range(W):
            if s == r:
                continue
            s0, s1 = bounds[s]
            if s1 > s0:
                sends.append((t[s0:s1], s))
            if mine > 0:
                recvs.append((rv[s, :mine], s))
    _p2p(sends, recvs, group)
    for t, bounds, mine, rv in zip(tensors, plans, mines, recvs_of):
        if mine > 0:
            _rank_ordered_sum_(rv, t[bounds[r][0]:bounds[r][1]], r, W, average)
    sends, recvs = [], []
    for t, bounds, mine in zip(tensors, plans, mines):
        b0, b1 = bounds[r]
        for s in range(W):
            if s == r:
                continue
            s0, s1 = bounds[s]
            if mine > 0:
                sends.append((t[b0:b1], s))
            if s1 > s0:
                recvs.append((t[s0:s1], s))
    _p2p(sends, recvs, group)


def p2p_allreduce_(flat, average=False, group=None, align=256, force=False):
    """p2p_allreduce_multi_ for one 1-D fp32 tensor."""
    p2p_allreduce_multi_([flat], average, group, align, force)


def _allreduce(flat, average, group):
    if ALLREDUCE_MODE == "p2p" and flat.dim() == 1 and flat.is_contiguous() and flat.numel() >= P2P_MIN_NUMEL:
        p2p_allreduce_(flat, average, group, force=FORCE_COLLECTIVES)
        return
    if average is True and dist.get_backend(group) == "nccl":
        dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=group)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            flat /= _divisor(average, group)


def allreduce_slices(tensors, average=False, group=None):
    """One fused collective over several contiguous fp32 tensors (the six row ranges of one Gaussian-index chunk of the
    gradient bucket): RCCL gets them as one group call (coalescing manager), other backends one all-reduce each."""
    tensors = [t for t in tensors if t.numel() > 0]
    if not tensors:
        return
    if ALLREDUCE_MODE == "p2p":      # the chunk's slices in ONE pair of point-to-point batches
        big = [t.reshape(-1) for t in tensors if t.is_contiguous() and t.numel() >= P2P_MIN_NUMEL]
        p2p_allreduce_multi_(big, average, group, force=FORCE_COLLECTIVES)
        for t in tensors:
            if not (t.is_contiguous() and t.numel() >= P2P_MIN_NUMEL):
                _allreduce(t, average, group)
        return
    if dist.get_backend(group) == "nccl" and hasattr(dist, "_coalescing_manager"):
        op = dist.ReduceOp.AVG if average is True else dist.ReduceOp.SUM
        with dist._coalescing_manager(group=group, device=tensors[0].device, async_ops=False):
            for t in tensors:
                dist.all_reduce(t, op=op, group=group)
        if average and average is not True:        # "mesh": a sum over all ranks, a mean over the views
            for t in tensors:
                t /= _divisor(average, group)
        return
    for t in tensors:
        _allreduce(t, average, group)


def chunk_bounds(P, chunks, align=256):
    """Gaussian-index chunk boundaries for dgs_backward_geometry: `chunks` ranges, each starting on a multiple of `align`."""
    per = -(-P // max(int(chunks), 1))
    per = -(-per // align) * align
    b = list(range(0, P, per)) + [P] if P > 0 else [0, 0]
    return list(zip(b[:-1], b[1:]))


def allreduce_small_grads(params, average=False, group=None, force=False):
    """The trajectory parameters (curve control points, alignment) are replicated like the cloud; their gradients are a
    few KB, packed into one small buffer and reduced next to the per-Gaussian bucket.  A parameter without a gradient on
    this rank (its view was rendered elsewhere, or nothing reached it) contributes zeros, so that every rank calls the
    collective with the same layout.  force: run the collective in a one-rank group too (the RCCL smoke test)."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not (force or FORCE_COLLECTIVES)):
        return
    params = [p for p in params if p is not None and p.numel() > 0]
    if not params:
        return
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float() for p in params])
    _allreduce(flat, average, group)
    o = 0
    for p in params:
        p.grad = flat[o:o + p.numel()].view_as(p).to(p.dtype)
        o += p.numel()


def flat_allreduce_grads(params, average=False, group=None, extra=None, force=False):
    """Sum (or average) the .grad of `params` over ranks with ONE collective on a flat fp32 bucket.  Gradients that
    already are views of one flat buffer are reduced in place (no packing copies); RCCL averages inside the
    collective.  `extra`: small replicated parameters (the trajectory groups) reduced by allreduce_small_grads.
    force: run the collectives in a one-rank group too (values unchanged; proves library load, init path and op support)."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not (force or FORCE_COLLECTIVES)):
        return
    if extra:
        allreduce_small_grads(extra, average=average, group=group, force=force)
    params = [p for p in params if p is not None]
    grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in params]
    shared = None
    if all(p.grad is not None for p in params):
        shared = _shared_flat(grads)
        if shared is None:
            shared = _adopt_into_bucket(params)
    if shared is not None:
        flat = shared
    else:   # pack, every segment starting on a 16-byte boundary (the fused optimiser reads float4)
        offs, total = [], 0
        for g in grads:
            offs.append(total)
            total += (g.numel() + 3) // 4 * 4
        flat = torch.zeros(total, dtype=torch.float32, device=grads[0].device)
        for g, o in zip(grads, offs):
            flat[o:o + g.numel()].copy_(g.reshape(-1))
    _allreduce(flat, average, group)
    if shared is not None:
        return
    for p, g, o in zip(params, grads, offs):
        p.grad = flat[o:o + g.numel()].view_as(g).to(g.dtype)


def allreduce_densification_stats(cloud, prev, group=None):
    """Data-parallel runs must densify identically on every rank, so the densification statistics each rank gathered
    from ITS view since the last synchronisation are combined first (determinism caveat): the increments
    of xyz_gradient_accum and denom are summed over ranks, max_radii2D is maximised.  `prev` = (accum, denom)
    snapshots taken right after the previous call (or zeros); returns the new snapshots."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not FORCE_COLLECTIVES):
        return cloud.xyz_gradient_accum.clone(), cloud.denom.clone()
    d_acc = cloud.xyz_gradient_accum - prev[0]
    d_den = cloud.denom - prev[1]
    buf = torch.cat([d_acc.reshape(-1), d_den.reshape(-1)])
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    n = d_acc.numel()
    cloud.xyz_gradient_accum = prev[0] + buf[:n].view_as(d_acc)
    cloud.denom = prev[1] + buf[n:].view_as(d_den)
    dist.all_reduce(cloud.max_radii2D, op=dist.ReduceOp.MAX, group=group)
    return cloud.xyz_gradient_accum.clone(), cloud.denom.clone()


def _sgn(x):
    return torch.sign(x)


def subframe_sharded_loss_backward(local_subframes, gt, K, k0, lambda_t, group=None):
    """Loss + backward of one view whose K subframes are split over ranks.

    local_subframes: [k_loc,3,H,W] rendered by this rank (k0 = index of its first subframe), attached to the
    autograd graph of the rasteriser.  Computes the reference loss block (train.py:147-163 image terms:
    L1(mean_k, gt) + lambda_t * L1(sub[k+1]-sub[k])) across ranks and back-propagates dL/dsubframes through the
    local graph.  Returns (l1, smooth) as python floats (identical on every rank)."""
    dS, l1, sm = subframe_sharded_loss_grad(local_subframes, gt, K, lambda_t, group)
    if local_subframes.shape[0] > 0:
        local_subframes.backward(gradient=dS)
    return float(l1), float(sm)


def subframe_sharded_loss_grad(local_subframes, gt, K, lambda_t, group=None):
    """The loss block of subframe_sharded_loss_backward without autograd: returns (dL/dlocal_subframes, l1, smooth),
    the two values as device scalars (identical on every rank).  fused_step.FusedStep hands the gradient straight to the
    rasteriser's backward."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    S = local_subframes.detach()
    k_loc = S.shape[0]
    E = gt.numel()
    # ---- exchange 1: partial blur sum
    blur = S.sum(dim=0) if k_loc > 0 else torch.zeros_like(gt)
    every = world > 1 or (FORCE_COLLECTIVES and dist.is_initialized())
    if every:
        dist.all_reduce(blur, op=dist.ReduceOp.SUM, group=group)
    blur = blur / K
    # ---- exchange 2: boundary subframes, point to point ("send/recv of one boundary subframe"): a rank sends
    # its first frame to the nearest lower rank that holds subframes and its last frame to the nearest higher one.  Who
    # holds what follows from shard_range on the host -- no count gather, no host read of a device value.
    prev_last = None   # subframe k0-1
    next_first = None  # subframe k0+k_loc
    if world > 1 and k_loc > 0:
        holds = [shard_range(K, r, world) for r in range(world)]
        assert holds[rank][1] - holds[rank][0] == k_loc, "local_subframes does not match shard_range(K, rank, world)"
        lower = next((r for r in range(rank - 1, -1, -1) if holds[r][1] > holds[r][0]), None)
        upper = next((r for r in range(rank + 1, world) if holds[r][1] > holds[r][0]), None)
        sends, recvs = [], []
        if lower is not None:
            prev_last = torch.empty_like(gt)
            sends.append((S[0].contiguous(), lower))
            recvs.append((prev_last, lower))
        if upper is not None:
            next_first = torch.empty_like(gt)
            sends.append((S[-1].contiguous(), upper))
            recvs.append((next_first, upper))
        _p2p(sends, recvs, group)
    if S.is_cuda and 0 < k_loc <= 32 and S.dtype == torch.float32:
        # device tensors: the whole block in ONE launch (dgs_blur_loss_slice_grad: dL/dsubframes, the L1 value and this
        # rank's share of the smoothness value, deterministic totals) instead of ~20 elementwise / reduction launches
        import ctypes
        from . import _lib
        Sc, gtc, blurc = S.contiguous(), gt.contiguous().float(), blur.contiguous()
        dS = torch.empty_like(Sc)
        work = torch.empty(8, dtype=torch.float32, device=S.device)
        ptr = lambda t: None if t is None else ctypes.c_void_p(t.contiguous().data_ptr())
        prev_c = None if prev_last is None else prev_last.contiguous()
        next_c = None if next_first is None else next_first.contiguous()
        stream = ctypes.c_void_p(torch.cuda.current_stream(S.device).cuda_stream)
        _lib.check(_lib.lib().dgs_blur_loss_slice_grad(ptr(Sc), ptr(prev_c), ptr(next_c), ptr(blurc), ptr(gtc), k_loc, int(K),
                                                       int(gt.shape[0]) if gt.dim() == 3 else 1,
                                                       int(E // (gt.shape[0] if gt.dim() == 3 else 1)), float(lambda_t),
                                                       ptr(dS), ptr(work), stream), "dgs_blur_loss_slice_grad")
        l1v, sm_local = work[0], work[1:2]
        if every:
            dist.all_reduce(sm_local, op=dist.ReduceOp.SUM, group=group)
        return dS, l1v, sm_local.reshape(())
    d = blur - gt
    g_l1 = _sgn(d) / (E * K)
    dS = g_l1[None].expand_as(S).clone() if k_loc > 0 else S
    sm_local = torch.zeros((), device=gt.device)
    if K > 1 and k_loc > 0:
        ws = lambda_t / (E * (K - 1))
        ext = [S]
        if prev_last is not None:
            ext = [prev_last[None]] + ext
        if next_first is not None:
            ext = ext + [next_first[None]]
        X = torch.cat(ext, dim=0)
        diff = X[1:] - X[:-1]                      # differences touching this rank's frames
        sg = _sgn(diff)
        lo = 1 if prev_last is not None else 0
        # dL/dx_k = ws * (sign(x_k - x_{k-1}) - sign(x_{k+1} - x_k))
        left = torch.zeros_like(S)
        right = torch.zeros_like(S)
        if lo == 1:
            left += sg[0:k_loc]
        elif k_loc > 1:
            left[1:] += sg[0:k_loc - 1]
        n_right = k_loc if next_first is not None else k_loc - 1
        if n_right > 0:
            right[:n_right] += sg[lo:lo + n_right]
        dS = dS + ws * (left - right)
        # each difference is counted once: a rank owns the differences whose LEFT frame it holds
        own = diff[lo:lo + n_right]
        sm_local = own.abs().sum() / (E * (K - 1))
    if every:
        dist.all_reduce(sm_local, op=dist.ReduceOp.SUM, group=group)
    return dS, d.abs().mean(), sm_local
