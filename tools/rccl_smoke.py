"""RCCL smoke test on a ONE-GPU box: a one-rank process group on backend "nccl" (= RCCL) with every collective of the
product path forced on (DGS_DIST_FORCE_COLLECTIVES=1), under real TrainingLoop steps.

    python tools/rccl_smoke.py

Proves, before the driver's 8-GPU run does: librccl loads, init_process_group(backend="nccl", device_id=...) works,
ReduceOp.AVG on the in-place gradient bucket, ReduceOp.MAX on the int32 skip flag, broadcast of the shared draws, the
all-reduces of the "subframes" loss block, the packed small-gradient all-reduce and the densification-statistics
reduction are all supported -- and that with one rank every one of them is the identity: the trained parameters are
bit-identical to the same run with the collectives skipped.  The point-to-point fallback (sharding.p2p_allreduce_) runs
on RCCL too, as a self send + receive inside one RCCL group.
"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def train(mode, iters=12, ar_chunks=1, graph="auto", depth_tv=0.01):
    import torch
    from deblurgs_amd import synthetic
    from deblurgs_amd.cloud import GaussianCloud
    from deblurgs_amd.motion import CameraMotionModule, RefCamera
    from deblurgs_amd.training import TrainingLoop, default_optimization_params
    dev = torch.device("cuda", 0)
    K = 5
    sc = synthetic.make_scene(3000, 128, 96, K=K, seed=21, sigma_px=3.0)
    cloud = GaussianCloud.from_scene(sc, dev)
    ref = RefCamera(sc["W"], sc["H"], sc["FoVx"], sc["FoVy"], device=dev)
    torch.manual_seed(100)
    gt = torch.rand(2, 3, sc["H"], sc["W"], device=dev) * 0.5
    m = CameraMotionModule(ref, gt, curve_order=3, num_subframes=K, device=dev, curve_random_sample=True)
    with torch.no_grad():
        m._trans._control_points.copy_(torch.from_numpy(sc["ctrl_trans"])[None].to(dev).expand(2, -1, -1))
        m._rot._control_points.copy_(torch.from_numpy(sc["ctrl_rot"])[None].to(dev).expand(2, -1, -1))
    opt = default_optimization_params(iterations=iters + 10, curve_start_iter=2, densify_from_iter=3,
                                      densification_interval=4, densify_until_iter=iters - 2,
                                      densify_grad_threshold_init=2e-5, densify_grad_threshold_final=1e-5,
                                      opacity_reset_interval=1000, curve_alignment_lr=1e-3, curve_alignment_start=4,
                                      lambda_depth_tv=depth_tv)
    loop = TrainingLoop(cloud, m, opt, cameras_extent=1.0, distributed=mode, ar_chunks=ar_chunks, graph=graph)
    for it in range(1, iters + 1):
        torch.manual_seed(it)
        loop.step(it, it % 2)
    loop.flush()
    torch.cuda.synchronize()
    train.replayed = 0 if loop._fused is None else loop._fused.replayed
    return [p.detach().clone() for p in list(cloud.hot_parameters()) + list(m.parameters())]


def main():
    os.environ.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      DGS_DIST_FORCE_INIT="1", DGS_DIST_FORCE_COLLECTIVES="1")
    os.environ.setdefault("MASTER_PORT", "29731")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from deblurgs_amd import sharding
    rank, world, local = sharding.init_distributed("cuda")
    assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
    # the raw pieces first: in-place AVG on a bucket, MAX on an int32 word, broadcast, p2p-free loss block
    flat = torch.arange(64, dtype=torch.float32, device="cuda") * 0.25
    views = [torch.nn.Parameter(torch.zeros(10, 3, device="cuda")), torch.nn.Parameter(torch.zeros(7, device="cuda"))]
    views[0].grad, views[1].grad = flat[0:30].view(10, 3), flat[32:39]
    keep = flat.clone()
    small = torch.nn.Parameter(torch.ones(5, device="cuda"))
    small.grad = torch.full((5,), 3.0, device="cuda")
    sharding.flat_allreduce_grads(views, average=True, extra=[small], force=True)
    assert torch.equal(flat, keep) and views[0].grad.data_ptr() == flat.data_ptr(), "in-place AVG all-reduce"
    assert torch.equal(small.grad, torch.full((5,), 3.0, device="cuda"))
    flag = torch.tensor([0, 1], dtype=torch.int32, device="cuda")
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    assert flag.tolist() == [0, 1]
    # SURVEY 8e's fallback on RCCL's point-to-point path: in a one-rank group p2p_allreduce_(force=True) runs both of its
    # phases as ONE RCCL group holding a send to and a receive from this rank itself (batch_isend_irecv on device tensors,
    # stream-ordered) -- the values must come back unchanged, also when the producer is still running on the stream
    for n in (255, 65536 + 7, 3_000_000):
        x = torch.randn(n, device="cuda")
        y = x * 1.0                                   # (produced on the stream right before the sends)
        sharding.p2p_allreduce_(y, average=True, force=True)
        assert torch.equal(x, y), f"p2p self send/recv changed a buffer of {n} floats"
    print("rccl smoke: p2p_allreduce_ (batch_isend_irecv, self send + receive in one RCCL group) bit-identical", flush=True)
    for mode in ("views", "subframes"):
        sharding.FORCE_COLLECTIVES = False     # one rank: the gradient / loss-block / statistics collectives are skipped
        base = train(mode)
        sharding.FORCE_COLLECTIVES = True      # ... and now every one of them goes through RCCL
        got = train(mode)
        for i, (a, b) in enumerate(zip(base, got)):
            assert a.shape == b.shape and torch.equal(a, b), f"mode {mode}: parameter {i} changed by a one-rank collective"
        print(f"rccl smoke: mode {mode}: {len(got)} parameter tensors bit-identical with and without the collectives",
              flush=True)
        # the chunked, overlapped reduction of the gradient bucket (coalesced RCCL group calls on a side stream)
        got = train(mode, ar_chunks=4)
        for i, (a, b) in enumerate(zip(base, got)):
            assert a.shape == b.shape and torch.equal(a, b), f"mode {mode}, 4 chunks: parameter {i} changed"
        print(f"rccl smoke: mode {mode}: 4-chunk overlapped all-reduce bit-identical too", flush=True)
        if True:
            # the sharded step with its front replayed as a captured hipGraph from the first possible iteration on
            # (FusedStep.replay_front), RCCL reductions behind it on the side stream: same parameters, and it did replay
            # (without the depth-smoothness term, which is not captured: its own eager baseline)
            base0 = train(mode, ar_chunks=4, graph=False, depth_tv=0.0)
            got = train(mode, ar_chunks=4, graph="always", depth_tv=0.0)
            for i, (a, b) in enumerate(zip(base0, got)):
                assert a.shape == b.shape and torch.equal(a, b), f"mode {mode}, captured front: parameter {i} changed"
            assert train.replayed >= 2, train.replayed
            print(f"rccl smoke: mode {mode}: captured front + RCCL reductions bit-identical ({train.replayed} replays)", flush=True)
        # ... and the same steps with every reduction of the bucket (whole, then per chunk on the side stream) and the
        # few-KB trajectory buffer taking the point-to-point reduce-scatter path on RCCL
        sharding.ALLREDUCE_MODE, keep_min = "p2p", sharding.P2P_MIN_NUMEL
        sharding.P2P_MIN_NUMEL = 0
        for chunks in (1, 4):
            got = train(mode, ar_chunks=chunks)
            for i, (a, b) in enumerate(zip(base, got)):
                assert a.shape == b.shape and torch.equal(a, b), f"mode {mode}, p2p, {chunks} chunk(s): parameter {i} changed"
        sharding.ALLREDUCE_MODE, sharding.P2P_MIN_NUMEL = "collective", keep_min
        print(f"rccl smoke: mode {mode}: p2p reduce-scatter (whole bucket and 4 chunks) bit-identical too", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    print("rccl smoke ok: backend nccl, world 1", flush=True)


if __name__ == "__main__":
    main()
