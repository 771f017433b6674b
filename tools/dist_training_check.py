"""Functional check of the sharded training loop (TrainingLoop(distributed="views" | "subframes")) with N ranks.  Works
on a ONE-GPU box too (all ranks on cuda:0, gloo collectives staged through the host):

    DGS_DIST_BACKEND=gloo DGS_DIST_ONE_DEVICE=1 python tools/dist_training_check.py --ranks 2 --mode views
    DGS_DIST_BACKEND=gloo DGS_DIST_ONE_DEVICE=1 python tools/dist_training_check.py --ranks 2 --mode subframes

Started plainly it launches the ranks itself (fresh child processes, before any GPU call).  Checks:
  views      ranks draw DIFFERENT cam_idx from one shared CameraMotionModule, with densification; afterwards the cloud
             AND the trajectory parameters (_rot / _trans control points, _nu) are bit-identical on every rank, and the
             per-Gaussian gradient bucket was reduced in place.
  subframes  every rank takes the same cam_idx and renders its share of the K subframes; replicas bit-identical, and
             with --out the final parameters are saved so that the caller can compare them with a --ranks 1 run (the
             sharded step is the single-process step up to summation order).
"""
import argparse
import os
import socket
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=2)
    ap.add_argument("--mode", default="views", choices=["views", "subframes", "mesh"])
    ap.add_argument("--mesh-views", type=int, default=2,
                    help="--mode mesh: Gv rows of ranks / Gv (each row splits one view's subframes; rank = v * Gs + s)")
    ap.add_argument("--iters", type=int, default=33)
    ap.add_argument("--no-densify", action="store_true")
    ap.add_argument("--out", default=None)
    ap.add_argument("--curve-start", type=int, default=2, help="curve_start_iter (all K subframes from this iteration)")
    ap.add_argument("--same-seed", action="store_true",
                    help="seed every rank identically per iteration (only for comparing a 1-rank with an N-rank run)")
    ap.add_argument("--random-sample", action="store_true", help="curve_random_sample on (alignment jitter)")
    ap.add_argument("--ar-chunks", type=int, default=1,
                    help="all-reduce the gradient bucket in this many Gaussian-index chunks, overlapped with the backward")
    ap.add_argument("--graph", default="auto", choices=["auto", "always", "off"],
                    help="TrainingLoop(graph=...): 'always' replays the sharded step's front as a hipGraph whenever possible")
    ap.add_argument("--densify-interval", type=int, default=6)
    ap.add_argument("--force-overflow", type=int, default=0,
                    help="at this iteration rank 1 pretends its view used to need a quarter of the duplicates: its capacity "
                         "overflows, the MAX-reduced flag must make EVERY rank drop the step and correct its step counters")
    ap.add_argument("--config", default=None,
                    help="a BASELINE.json configuration of deblurgs_amd.synthetic (metric, cfg2, cfg3) instead of the small "
                         "3000-Gaussian scene: the sharded step at the size bench.py times (tests/test_gpu_train.py)")
    ap.add_argument("--cam-offset", type=int, default=0, help="added to every rank's view index (one-rank reference runs)")
    ap.add_argument("--depth-tv", type=float, default=0.0, help="lambda_depth_tv (one more collective per step in subframes mode)")
    ap.add_argument("--p2p-direct", action="store_true",
                    help="REPRODUCTION AID, not a product path: replace sharding._p2p by round 3's behaviour (batch_isend_irecv "
                         "handed device tensors on any backend).  Under gloo the host then reads / writes device memory "
                         "unordered with the stream: replicas diverge on some boxes (GPUTEST_r03)")
    return ap.parse_args()


def _p2p_direct(sends, recvs, group=None):
    """Round 3's point-to-point layer, kept ONLY here to reproduce its race (see --p2p-direct)."""
    import torch.distributed as dist
    ops = [dist.P2POp(dist.isend, t, r, group) for t, r in sends] + [dist.P2POp(dist.irecv, t, r, group) for t, r in recvs]
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()


def launch(args):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.ranks), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rcs = [p.wait() for p in procs]
    return 0 if all(rc == 0 for rc in rcs) else 1


def main():
    args = parse()
    if args.ranks > 1 and "RANK" not in os.environ:
        sys.exit(launch(args))
    import torch
    import torch.distributed as dist
    from deblurgs_amd import sharding, synthetic
    from deblurgs_amd.cloud import GaussianCloud
    from deblurgs_amd.motion import CameraMotionModule, RefCamera
    from deblurgs_amd.training import TrainingLoop, default_optimization_params

    rank, world, local = sharding.init_distributed("cuda")
    if args.p2p_direct:
        sharding._p2p = _p2p_direct
    dev = torch.device("cuda", 0 if os.environ.get("DGS_DIST_ONE_DEVICE", "0") == "1" else local)
    torch.cuda.set_device(dev)
    K, C = 5, 3
    if args.config:
        sc = synthetic.make_config(args.config, seed=0)
        K, C = sc["K"], synthetic.CONFIGS[args.config]["C"]
        traj = synthetic.make_trajectory(K, C, sc["projection_matrix"], seed=0)
        sc["ctrl_trans"], sc["ctrl_rot"] = traj["ctrl_trans"], traj["ctrl_rot"]
    else:
        sc = synthetic.make_scene(3000, 128, 96, K=K, seed=21, sigma_px=3.0)
    cloud = GaussianCloud.from_scene(sc, dev)
    ref = RefCamera(sc["W"], sc["H"], sc["FoVx"], sc["FoVy"], device=dev)
    gv = args.mesh_views if args.mode == "mesh" else world
    vrow = rank // max(world // max(gv, 1), 1) if args.mode == "mesh" else rank     # the view index this rank works on
    n_views = max(gv if args.mode == "mesh" else world, 2)
    torch.manual_seed(100)                        # the SAME module (ground truths, curves) on every rank
    gt = torch.rand(n_views, 3, sc["H"], sc["W"], device=dev) * 0.5
    m = CameraMotionModule(ref, gt, curve_order=C, num_subframes=K, device=dev, curve_random_sample=args.random_sample)
    with torch.no_grad():
        base = torch.from_numpy(sc["ctrl_trans"])[None].to(dev)
        m._trans._control_points.copy_(base + 0.01 * torch.arange(n_views, device=dev).reshape(-1, 1, 1))
        m._rot._control_points.copy_(torch.from_numpy(sc["ctrl_rot"])[None].to(dev).expand(n_views, -1, -1))
    far = 10 ** 9
    opt = default_optimization_params(
        iterations=args.iters + 10, curve_start_iter=args.curve_start, densify_from_iter=far if args.no_densify else 5,
        densification_interval=args.densify_interval, densify_until_iter=args.iters - 3, densify_grad_threshold_init=2e-5,
        densify_grad_threshold_final=1e-5, opacity_reset_interval=1000, curve_alignment_lr=1e-3, curve_alignment_start=4,
        lambda_depth_tv=args.depth_tv)
    loop = TrainingLoop(cloud, m, opt, cameras_extent=1.0, distributed=args.mode if world > 1 else False,
                        ar_chunks=args.ar_chunks, graph={"auto": "auto", "always": "always", "off": False}[args.graph],
                        mesh=(gv, world // gv) if args.mode == "mesh" else None)
    inplace = []
    _orig = sharding.flat_allreduce_grads

    def _spy(params, **kw):
        r = _orig(params, **kw)
        inplace.append(sharding._shared_flat([p.grad for p in params]) is not None)
        return r
    sharding.flat_allreduce_grads = _spy
    sizes = []
    snap = {}
    _step = cloud.optimizer.step

    def _step_spy(*a, **kw):          # the gradients the optimiser sees at the first iteration with all K subframes
        if snap.get("it") == args.curve_start:
            snap["grads"] = [None if p.grad is None else p.grad.detach().cpu().clone()
                             for p in list(cloud.hot_parameters()) + list(m.parameters())]
        return _step(*a, **kw)
    cloud.optimizer.step = _step_spy
    for it in range(1, args.iters + 1):
        snap["it"] = it
        torch.manual_seed(it if args.same_seed else 7919 * (rank + 1) + it)   # ranks draw DIFFERENT random numbers
        cam = ((it + vrow + args.cam_offset) % n_views if args.mode in ("views", "mesh") else
               (it + args.cam_offset) % n_views)
        if it == args.force_overflow and rank == min(1, world - 1) and loop._fused is not None:
            loop._fused._poll(block=True)
            assert loop._fused._seen, "no duplicate count learnt yet: nothing to shrink"
            loop._fused._seen = {k: [max(c // 4, 1) for c in v] for k, v in loop._fused._seen.items()}
        out = loop.step(it, cam)
        sizes.append(out["num_points"])
    loop.flush()
    tensors = list(cloud.hot_parameters()) + list(m.parameters())
    sig = torch.stack([p.detach().double().sum() for p in tensors] +
                      [p.detach().double().abs().sum() for p in tensors] +
                      [torch.tensor(float(cloud._xyz.shape[0]), device=dev, dtype=torch.float64)])
    same = True
    if world > 1:
        allsig = [torch.zeros_like(sig) for _ in range(world)]
        dist.all_gather(allsig, sig)
        same = all(torch.equal(allsig[0], s) for s in allsig)
    moved = [float((p.detach() - q).abs().max()) for p, q in
             zip(m.parameters(), [torch.zeros_like(p) for p in m.parameters()])]
    dropped = torch.tensor([float(loop.dist_dropped), float(cloud.optimizer.state[cloud._xyz]["step"])], device=dev,
                           dtype=torch.float64)
    all_dropped = [dropped.clone() for _ in range(world)]
    if world > 1:
        dist.all_gather(all_dropped, dropped)
    if rank == 0:
        print("dist_dropped per rank " + " ".join(str(int(d[0])) for d in all_dropped) + " xyz steps per rank " +
              " ".join(str(int(d[1])) for d in all_dropped), flush=True)
        import hashlib
        hsh = hashlib.sha1()
        for p_ in tensors:
            hsh.update(p_.detach().cpu().numpy().tobytes())
        print(f"params sha1 {hsh.hexdigest()[:16]} replayed {0 if loop._fused is None else loop._fused.replayed} "
              f"dropped {0 if loop._fused is None else loop._fused.dropped} dist_dropped {loop.dist_dropped}", flush=True)
        print(f"mode {args.mode} ranks {world} points {sizes[0]} -> {sizes[-1]} densified: {len(set(sizes)) > 1} "
              f"replicas (cloud + trajectory) identical: {same} bucket reduced in place: "
              f"{all(inplace) if inplace else None} curve state steps: "
              f"{float(cloud.optimizer.state[m._trans._control_points]['step'])}", flush=True)
        assert same, "replicas diverged"
        assert args.no_densify or len(set(sizes)) > 1, "densify_and_prune never changed the cloud"
        assert world == 1 or not inplace or all(inplace[1:]), "the gradient bucket must be reduced in place"
        assert args.iters < 5 or (cloud.optimizer.state[m._nu]["step"] > 0 and moved[0] > 0)
        if args.out:
            torch.save({"params": [p.detach().cpu() for p in tensors], "sizes": sizes, "grads_first": snap.get("grads"),
                        "replayed": 0 if loop._fused is None else loop._fused.replayed,
                        "captured": 0 if loop._fused is None else loop._fused.captured}, args.out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
