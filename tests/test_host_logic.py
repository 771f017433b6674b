"""CPU tests of the host-side logic: synthetic generator statistics, trajectory module, sharding partition and
the N>1 paths with world_size-2 gloo processes (a differentiable stand-in replaces the rasteriser, which needs
a GPU; what is under test is the collective logic around it)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import oracle_forward, synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_synthetic_generator_matches_survey_statistics():
    sc = synthetic.make_config("cfg1")
    assert sc["means3D"].dtype == np.float32 and sc["sh"].shape == (1000, 9, 3)
    st = oracle_forward(sc, 0, render=False)
    Pv = int((st["radii"] > 0).sum())
    assert 820 <= Pv <= 920            # SURVEY 8d dry run: ~867 visible, R ~3.9k
    assert 3000 <= st["num_rendered"] <= 5000
    # K subframe poses are small perturbations of the identity
    sc = synthetic.make_scene(10, 64, 64, K=5, curve_order=3)
    assert np.abs(sc["viewmatrix"] - np.eye(4)[None]).max() < 0.1
    assert np.allclose(sc["projmatrix"][2], sc["viewmatrix"][2] @ sc["projection_matrix"], atol=1e-6)
    a, b = synthetic.make_scene(50, 32, 32, seed=3), synthetic.make_scene(50, 32, 32, seed=3)
    assert all(np.array_equal(a[k], b[k]) for k in ("means3D", "sh", "scales", "viewmatrix"))


def test_shard_range_covers_k_exactly_once():
    from deblurgs_amd.sharding import shard_range
    for K in (1, 2, 9, 15, 31):
        for G in (1, 2, 4, 8):
            parts = [shard_range(K, g, G) for g in range(G)]
            assert parts[0][0] == 0 and parts[-1][1] == K
            assert all(parts[i][1] == parts[i + 1][0] for i in range(G - 1))
    assert [shard_range(15, g, 8)[1] - shard_range(15, g, 8)[0] for g in range(8)].count(2) == 7


def test_motion_module_trajectory_semantics():
    from deblurgs_amd.motion import CameraMotionModule, RefCamera
    torch.manual_seed(0)
    ref = RefCamera(64, 48, 1.0, 0.8, device="cpu")
    m = CameraMotionModule(ref, torch.rand(2, 3, 48, 64), curve_order=3, num_subframes=7, device="cpu")
    nu = m._sample_nu_from_alignment(0)
    assert torch.allclose(nu, torch.linspace(0, 1, 7), atol=1e-6)      # scene/motion.py:55 initialisation
    wv, fp, cc = m.get_trajectory_matrices(1)
    assert wv.shape == (7, 4, 4) and wv.dtype == torch.float32 and cc.shape == (7, 3)
    assert torch.allclose(fp, wv @ ref.projection_matrix, atol=1e-6)
    cams = m.get_trajectory(1)
    assert len(cams) == 7 and torch.allclose(cams[3].camera_center, cc[3])
    # gradients reach the control points and nu through the pose path
    (fp.sum() + wv.sum()).backward()
    assert m._trans._control_points.grad.abs().sum() > 0 and m._rot._control_points.grad.abs().sum() > 0
    assert m._nu.grad.abs().sum() > 0
    # int subframe_indice: linspace(0, f-1, n).long() (scene/motion.py:129-131); 1 -> index 0
    idx = torch.linspace(0, 6, 1).long()
    assert idx.tolist() == [0]


# ----------------------------------------------------------------------------------- world_size-2 gloo tests
def _standin_render(params, view, proj, H=6, W=5):
    """Differentiable stand-in for the fused rasteriser: [K,3,H,W] from the cloud parameters and K poses."""
    K = view.shape[0]
    base = (params[0][:, :3].sum(0)[None, :, None, None] * torch.ones(K, 3, H, W))
    wave = torch.sin(torch.arange(H * W, dtype=torch.float32).reshape(1, 1, H, W) * 0.37 + params[1].sum())
    posefac = (view[:, :3, :3].sum(dim=(1, 2)) + proj[:, 3, :3].sum(dim=1)).reshape(K, 1, 1, 1)
    return base * 0.1 + wave * posefac.float() * 0.05 + 0.3 * torch.arange(K).reshape(K, 1, 1, 1) ** 0.5


def _setup(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    from deblurgs_amd import sharding
    sharding.init_distributed("cpu")
    return sharding


def _worker_subframes(rank, world, port, K, out):
    sharding = _setup(rank, world, port)
    torch.manual_seed(0)
    params = [torch.randn(20, 3, requires_grad=True), torch.randn(7, requires_grad=True)]
    view = (torch.eye(4)[None].repeat(K, 1, 1) + 0.01 * torch.randn(K, 4, 4)).requires_grad_(True)
    proj = (torch.eye(4)[None].repeat(K, 1, 1) + 0.01 * torch.randn(K, 4, 4)).requires_grad_(True)
    gt = torch.rand(3, 6, 5)
    lam = 0.05
    k0, k1 = sharding.shard_range(K, rank, world)
    sub = _standin_render(params, view, proj)[k0:k1]
    l1, sm = sharding.subframe_sharded_loss_backward(sub, gt, K, k0, lam)
    sharding.flat_allreduce_grads(params + [view, proj], average=False)
    if rank == 0:
        torch.save(dict(l1=l1, sm=sm, g=[p.grad.clone() for p in params + [view, proj]]), out)
    dist.barrier()
    dist.destroy_process_group()


def _worker_views(rank, world, port, out):
    sharding = _setup(rank, world, port)
    torch.manual_seed(0)
    params = [torch.randn(20, 3, requires_grad=True), torch.randn(7, requires_grad=True)]
    torch.manual_seed(100 + rank)      # each rank renders its own view
    view = torch.eye(4)[None].repeat(3, 1, 1) + 0.01 * torch.randn(3, 4, 4)
    gt = torch.rand(3, 6, 5)
    sub = _standin_render(params, view, view)
    ((sub.mean(0) - gt).abs().mean()).backward()
    sharding.flat_allreduce_grads(params, average=True)
    if rank == 0:
        torch.save([p.grad.clone() for p in params], out)
    # the same gradients laid out as views of one padded flat buffer (what the fused-activation operator returns)
    # must be reduced in place, without re-binding .grad
    torch.manual_seed(200 + rank)
    flat = torch.full((64 + 4 + 8,), float("nan"))
    views = [flat[0:60].view(20, 3), flat[64:71]]
    for v_, p in zip(views, params):
        v_.copy_(torch.randn_like(p))
    mine = [v_.clone() for v_ in views]
    for v_, p in zip(views, params):
        p.grad = v_
    keep = [p.grad for p in params]
    sharding.flat_allreduce_grads(params, average=True)
    assert all(p.grad is k for p, k in zip(params, keep)), "shared-storage gradients must be reduced in place"
    assert sharding._shared_flat(keep) is not None
    gathered = [torch.zeros_like(m) for m in mine]
    for g_, m in zip(gathered, mine):
        g_.copy_(m)
        dist.all_reduce(g_)
        g_ /= world
    assert all(torch.allclose(p.grad, g_, atol=1e-7) for p, g_ in zip(params, gathered))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("K", [5, 2, 1])
def test_subframe_sharding_equals_single_process(tmp_path, K):
    """K subframes split over 2 ranks (blur all-reduce + boundary exchange + flat grad all-reduce) must give
    the single-process loss and gradients of the reference loss block."""
    from deblurgs_amd import losses
    out = str(tmp_path / "sub.pt")
    mp.spawn(_worker_subframes, args=(2, 29611 + K, K, out), nprocs=2, join=True)
    got = torch.load(out)
    torch.manual_seed(0)
    params = [torch.randn(20, 3, requires_grad=True), torch.randn(7, requires_grad=True)]
    view = (torch.eye(4)[None].repeat(K, 1, 1) + 0.01 * torch.randn(K, 4, 4)).requires_grad_(True)
    proj = (torch.eye(4)[None].repeat(K, 1, 1) + 0.01 * torch.randn(K, 4, 4)).requires_grad_(True)
    gt = torch.rand(3, 6, 5)
    sub = _standin_render(params, view, proj)
    total, blur, l1, sm = losses.blur_loss_torch(sub, gt, 0.05)
    total.backward()
    assert abs(got["l1"] - float(l1)) < 1e-6 and abs(got["sm"] - float(sm)) < 1e-6
    for a, p in zip(got["g"], params + [view, proj]):
        assert torch.allclose(a, p.grad, atol=1e-6), (a - p.grad).abs().max()


def test_view_sharding_averages_gradients(tmp_path):
    out = str(tmp_path / "views.pt")
    mp.spawn(_worker_views, args=(2, 29633, out), nprocs=2, join=True)
    got = torch.load(out)
    acc = None
    for rank in range(2):
        torch.manual_seed(0)
        params = [torch.randn(20, 3, requires_grad=True), torch.randn(7, requires_grad=True)]
        torch.manual_seed(100 + rank)
        view = torch.eye(4)[None].repeat(3, 1, 1) + 0.01 * torch.randn(3, 4, 4)
        gt = torch.rand(3, 6, 5)
        ((_standin_render(params, view, view).mean(0) - gt).abs().mean()).backward()
        g = [p.grad for p in params]
        acc = g if acc is None else [a + b for a, b in zip(acc, g)]
    for a, b in zip(got, acc):
        assert torch.allclose(a, b / 2, atol=1e-6)


def _worker_mesh(rank, world, port, K, out):
    """2 views x 2-way subframe sharding on four gloo ranks: row v splits view v's subframes (loss block inside the row's
    group), the bucket is summed over all four and divided by the two views."""
    sharding = _setup(rank, world, port)
    gv, gs = 2, 2
    v, s_, grp = sharding.mesh_groups(gv, gs)
    assert (v, s_) == divmod(rank, gs) and grp is not None
    torch.manual_seed(0)
    params = [torch.randn(20, 3, requires_grad=True), torch.randn(7, requires_grad=True)]
    torch.manual_seed(100 + v)          # the view of this ROW: both of its ranks hold the same poses and target
    view = (torch.eye(4)[None].repeat(K, 1, 1) + 0.01 * torch.randn(K, 4, 4)).requires_grad_(True)
    proj = (torch.eye(4)[None].repeat(K, 1, 1) + 0.01 * torch.randn(K, 4, 4)).requires_grad_(True)
    gt = torch.rand(3, 6, 5)
    k0, k1 = sharding.shard_range(K, s_, gs)
    sub = _standin_render(params, view, proj)[k0:k1]
    l1, sm = sharding.subframe_sharded_loss_backward(sub, gt, K, k0, 0.05, group=grp)
    sharding.flat_allreduce_grads(params, average=float(gv))       # sum over all ranks / number of views
    # both all-reduce paths give the same mean-of-sums
    y = torch.full((3000,), float(rank + 1))
    sharding.p2p_allreduce_(y, average=float(gv))
    assert torch.allclose(y, torch.full((3000,), (1 + 2 + 3 + 4) / 2.0))
    vals = torch.tensor([l1, sm])
    every = [torch.zeros(2) for _ in range(world)]
    dist.all_gather(every, vals)
    if rank == 0:
        torch.save(dict(g=[p.grad.clone() for p in params], losses=every), out)
    dist.barrier()
    dist.destroy_process_group()


def test_mesh_sharding_is_a_view_batch_of_subframe_sharded_views(tmp_path):
    """sharding.mesh_groups + the numeric `average` of the reductions: a 2 x 2 mesh gives the mean over the two views of
    each view's full single-process gradient, and every rank of a row sees its view's loss values."""
    from deblurgs_amd import losses
    K = 5
    out = str(tmp_path / "mesh.pt")
    mp.spawn(_worker_mesh, args=(4, 29671, K, out), nprocs=4, join=True)
    got = torch.load(out)
    acc, vals = None, []
    for v in range(2):
        torch.manual_seed(0)
        params = [torch.randn(20, 3, requires_grad=True), torch.randn(7, requires_grad=True)]
        torch.manual_seed(100 + v)
        view = (torch.eye(4)[None].repeat(K, 1, 1) + 0.01 * torch.randn(K, 4, 4)).requires_grad_(True)
        proj = (torch.eye(4)[None].repeat(K, 1, 1) + 0.01 * torch.randn(K, 4, 4)).requires_grad_(True)
        gt = torch.rand(3, 6, 5)
        total, blur, l1, sm = losses.blur_loss_torch(_standin_render(params, view, proj), gt, 0.05)
        total.backward()
        vals.append((float(l1), float(sm)))
        g = [p.grad for p in params]
        acc = g if acc is None else [a + b for a, b in zip(acc, g)]
    for a, b in zip(got["g"], acc):
        assert torch.allclose(a, b / 2, atol=1e-6), (a - b / 2).abs().max()
    for rank, lv in enumerate(got["losses"]):
        assert abs(float(lv[0]) - vals[rank // 2][0]) < 1e-6 and abs(float(lv[1]) - vals[rank // 2][1]) < 1e-6


def _worker_p2p_allreduce(rank, world, port, out):
    sharding = _setup(rank, world, port)
    res = {}
    for n in (1000, 65536 + 7, 3 * 256 * 5, 255):          # ragged last shard, exact multiple, fewer elements than ranks x align
        torch.manual_seed(10 * n + rank)
        x = torch.randn(n)
        for average in (False, True):
            y = x.clone()
            sharding.p2p_allreduce_(y, average=average)
            z = x.clone()
            dist.all_reduce(z)
            if average:
                z /= world
            res[(n, average)] = (y, z)
    # the switch: large 1-D buffers take the point-to-point path, small ones and the packed trajectory gradients the collective
    sharding.ALLREDUCE_MODE = "p2p"
    torch.manual_seed(rank)
    params = [torch.randn(30000, 3, requires_grad=True), torch.randn(11, requires_grad=True)]
    for p in params:
        p.grad = torch.randn_like(p)
    want = [p.grad.clone() for p in params]
    for w_ in want:
        dist.all_reduce(w_)
    sharding.flat_allreduce_grads(params, average=False)
    res["bucket"] = ([p.grad.clone() for p in params], want)
    # a chunk's six row ranges in ONE pair of point-to-point batches (allreduce_slices -> p2p_allreduce_multi_); the small
    # slice (below P2P_MIN_NUMEL) takes the collective
    torch.manual_seed(50 + rank)
    flat = torch.randn(400_000)
    cuts = [(0, 70_001), (70_004, 140_005), (140_008, 340_000), (340_000, 340_100), (340_100, 399_999)]
    ref = flat.clone()
    dist.all_reduce(ref)
    ref /= world
    sharding.allreduce_slices([flat[a:b] for a, b in cuts], average=True)
    res["slices"] = (torch.cat([flat[a:b] for a, b in cuts]), torch.cat([ref[a:b] for a, b in cuts]))
    sharding.ALLREDUCE_MODE = "collective"
    torch.save(res, out + f".{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_p2p_allreduce_equals_the_collective_and_is_identical_on_every_rank(tmp_path):
    """SURVEY 8e's fallback (direct reduce-scatter + all-gather over point-to-point sends) on 3 gloo ranks: same sums as
    all_reduce to rounding, bit-identical on every rank (replicas must not drift), any buffer length."""
    out = str(tmp_path / "p2p.pt")
    mp.spawn(_worker_p2p_allreduce, args=(3, 29677, out), nprocs=3, join=True)
    got = [torch.load(out + f".{r}") for r in range(3)]
    for key in got[0]:
        if key == "slices":
            for r in range(3):
                assert torch.allclose(got[r][key][0], got[r][key][1], atol=1e-5)
                assert torch.equal(got[r][key][0], got[0][key][0]), f"slices: rank {r} holds different bits"
            continue
        if key == "bucket":
            for r in range(3):
                for a, b in zip(*got[r]["bucket"]):
                    assert torch.allclose(a, b, atol=1e-5)
            for a, b in zip(got[0]["bucket"][0], got[2]["bucket"][0]):
                assert torch.equal(a, b)
            continue
        y0, z0 = got[0][key]
        assert torch.allclose(y0, z0, atol=1e-5), key
        for r in (1, 2):
            assert torch.equal(got[r][key][0], y0), f"{key}: rank {r} holds different bits"


def _worker_stats(rank, world, port, out):
    sharding = _setup(rank, world, port)
    import types
    cloud = types.SimpleNamespace(xyz_gradient_accum=torch.full((6, 1), 10.0), denom=torch.full((6, 1), 3.0),
                                  max_radii2D=torch.zeros(6))
    prev = (cloud.xyz_gradient_accum.clone(), cloud.denom.clone())
    # each rank accumulates the statistics of its own view
    cloud.xyz_gradient_accum += torch.arange(6.0).reshape(6, 1) * (rank + 1)
    cloud.denom[rank::2] += 1.0
    cloud.max_radii2D[rank] = 7.0 + rank
    new_prev = sharding.allreduce_densification_stats(cloud, prev)
    assert torch.allclose(cloud.xyz_gradient_accum, 10.0 + torch.arange(6.0).reshape(6, 1) * 3)
    assert torch.allclose(cloud.denom, torch.full((6, 1), 4.0))
    assert cloud.max_radii2D[0] == 7.0 and cloud.max_radii2D[1] == 8.0
    assert torch.equal(new_prev[0], cloud.xyz_gradient_accum)
    if rank == 0:
        torch.save(True, out)
    dist.barrier()
    dist.destroy_process_group()


def test_densification_stats_are_combined_across_ranks(tmp_path):
    """SURVEY 8e determinism caveat: replicas only densify identically if the statistics are all-reduced first."""
    out = str(tmp_path / "stats.pt")
    mp.spawn(_worker_stats, args=(2, 29655, out), nprocs=2, join=True)
    assert torch.load(out) is True


def _worker_subframes3(rank, world, port, K, out):
    _worker_subframes(rank, world, port, K, out)


@pytest.mark.parametrize("K", [2, 4])
def test_subframe_sharding_three_ranks_with_an_empty_one(tmp_path, K):
    """K = 2 over 3 ranks leaves rank 0 without subframes (shard_range: [0,0), [0,1), [1,2)): the point-to-point boundary
    exchange must skip it (neighbours are computed from shard_range on the host), K = 4 gives slices of 1, 1, 2."""
    from deblurgs_amd import losses
    out = str(tmp_path / "sub3.pt")
    mp.spawn(_worker_subframes3, args=(3, 29700 + K, K, out), nprocs=3, join=True)
    got = torch.load(out)
    torch.manual_seed(0)
    params = [torch.randn(20, 3, requires_grad=True), torch.randn(7, requires_grad=True)]
    view = (torch.eye(4)[None].repeat(K, 1, 1) + 0.01 * torch.randn(K, 4, 4)).requires_grad_(True)
    proj = (torch.eye(4)[None].repeat(K, 1, 1) + 0.01 * torch.randn(K, 4, 4)).requires_grad_(True)
    gt = torch.rand(3, 6, 5)
    total, blur, l1, sm = losses.blur_loss_torch(_standin_render(params, view, proj), gt, 0.05)
    total.backward()
    assert abs(got["l1"] - float(l1)) < 1e-6 and abs(got["sm"] - float(sm)) < 1e-6
    for a, p in zip(got["g"], params + [view, proj]):
        assert torch.allclose(a, p.grad, atol=1e-6), (a - p.grad).abs().max()


@pytest.mark.parametrize("how", ["die:1", "timeout"])
def test_bench_launcher_does_not_wait_for_a_dead_rank(how):
    """bench.py --gpus N started plainly: when a rank exits non-zero (or the overall timeout passes) the launcher ends the
    other ranks and returns non-zero within seconds -- it never sits in communicate() until a collective times out."""
    import subprocess
    import time
    env = dict(os.environ, DGS_BENCH_SELFTEST=how if how != "timeout" else "hang:0", DGS_DIST_ONE_DEVICE="1",
               DGS_BENCH_TIMEOUT_S="3")
    env.pop("RANK", None)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 1, (r.returncode, r.stderr[-500:])
    assert time.time() - t0 < 60
    assert ("exited non-zero" in r.stderr) if how != "timeout" else ("timeout" in r.stderr)


def test_poll_charges_every_overflow_to_its_own_step():
    """ADVICE r4 (high), host side only: FusedStep._poll with several views in flight.  Every pending entry owns its
    pinned words, so the order [A1 ok, B1 overflowed, A2 ok, B2 overflowed, A3 ok] -- all complete at one poll -- yields
    exactly two retries, both view B's, and each view learns only its own counts.  (The round-4 code read a running
    drop counter from per-GRAPH blocks: this sequence charged B's drop to A, and a counter that seemed to run backwards
    became 2^32 - 1 queued retries.)"""
    from deblurgs_amd.fused_step import FusedStep, _Pending

    class Cloud:
        fused_activations = True

    class Done:
        def query(self):
            return True

        def synchronize(self):
            pass

    fs = FusedStep(Cloud(), motion=None)
    seq = [("A", 100, 0), ("B", 900, 1), ("A", 101, 0), ("B", 901, 1), ("A", 102, 0)]
    for view, count, over in seq:
        p = _Pending()
        p.host = torch.tensor([count, 0, over, 0 if over else count, 0, 0, 0, 0], dtype=torch.int32)
        p.event, p.capacity, p.speculative = Done(), 500, True
        p.key, p.generation, p.request = (view, 5, 0), fs._generation, (view, "all")
        fs._pending.append(p)
    fs._poll()
    assert fs.dropped == 2 and fs.retry == [("B", "all"), ("B", "all")]
    assert fs._seen[("A", 5, 0)] == [100, 101, 102] and fs._seen[("B", 5, 0)] == [900, 901]
    assert not fs._pending and len(fs._free_hosts) == 5


def test_rccl_log_parser_reads_version_algorithm_and_topology(tmp_path):
    """bench.py --gpus N writes every rank's RCCL log to a file (NCCL_DEBUG=INFO, set before init) and reports what the
    library chose ("verify which algorithm it picks").  The parser on a log in NCCL's / RCCL's INFO format."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("dgs_bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    log = tmp_path / "rank0.log"
    log.write_text(
        "host:1:1 [0] NCCL INFO RCCL version 2.22.3+hip6.4 HEAD:abc\n"
        "host:1:9 [0] NCCL INFO Channel 00/16 :    0   1   2   3   4   5   6   7\n"
        "host:1:9 [0] NCCL INFO Trees [0] 1/-1/-1->0->-1 [1] 1/-1/-1->0->-1\n"
        "host:1:9 [0] NCCL INFO Connected all rings\n"
        "host:1:9 [0] NCCL INFO comm 0x1 rank 0 nRanks 8 nNodes 1 localRanks 8 localRank 0 MNNVL 0\n"
        "host:1:9 [0] NCCL INFO ncclCommInitRankConfig comm 0x1 rank 0 nranks 8 cudaDev 0 - Init COMPLETE\n"
        "host:1:1 [0] NCCL INFO AllReduce: opCount 5 sendbuff 0x1 recvbuff 0x1 count 38000000 datatype 7 op 4 root 0\n"
        "host:1:1 [0] NCCL INFO 152000000 Bytes -> Algo 1 proto 2 time 901.5\n"
        "host:1:1 [0] NCCL INFO AllReduce: opCount 6 sendbuff 0x1 recvbuff 0x1 count 1 datatype 2 op 2 root 0\n"
        "host:1:1 [0] NCCL INFO 4 Bytes -> Algo 0 proto 0 time 8.1\n"
        "host:1:1 [0] NCCL INFO 152000000 Bytes -> Algo 1 proto 2 time 901.5\n")
    got = bench.parse_rccl_log(str(log))
    assert got["version"].startswith("RCCL 2.22.3")
    assert got["algo_proto_by_message_bytes"]["152000000"] == {"Ring/Simple": 2}
    assert got["algo_proto_by_message_bytes"]["4"] == {"Tree/LL": 1}
    assert got["calls_logged"] == {"AllReduce": 2}
    assert any("Init COMPLETE" in ln for ln in got["init_lines"]) and got["channels"] == 16
    # the image's RCCL 2.26 writes the banner as "RCCL version : <x>" and (newer NCCL) the tuner's choice by name
    log.write_text("runc:227:227 [0] NCCL INFO RCCL version : 2.26.6-HEAD:64f48b6\nHIP version  : 7.0\n"
                   "runc:227:227 [0] NCCL INFO AllReduce: 38000000 Bytes -> Algo RING proto SIMPLE channel{Lo..Hi}={0..15}\n"
                   "runc:227:227 [0] NCCL INFO AllReduce: opCount 0 sendbuff 0x7 recvbuff 0x7 count 75264 datatype 7 op 4 root 0 "
                   "comm 0x65 [nranks=8] stream 0x65 task 0 globalrank 0\n")
    got2 = bench.parse_rccl_log(str(log))
    assert got2["version"] == "RCCL 2.26.6-HEAD:64f48b6" and got2["calls_logged"] == {"AllReduce": 1}
    assert got2["algo_proto_by_message_bytes"] == {"38000000": {"Ring/Simple": 1}}
    assert bench.parse_rccl_log(str(tmp_path / "missing.log")) is None
    # a rank only redirects its log for the real backend
    keep = {k: os.environ.get(k) for k in ("DGS_DIST_BACKEND", "NCCL_DEBUG", "NCCL_DEBUG_FILE", "NCCL_DEBUG_SUBSYS")}
    try:
        os.environ["DGS_DIST_BACKEND"] = "gloo"
        assert bench.rccl_log_setup(0) is None and os.environ.get("NCCL_DEBUG") == keep["NCCL_DEBUG"]
    finally:
        for k, v in keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
