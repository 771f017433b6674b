#!/usr/bin/env python3
"""bench.py -- subframe-renders/sec (fwd+bwd) of the blur-integration hot path on synthetic Gaussian clouds.

    python bench.py --gpus N --steps K --warmup W

N > 1: started plainly, this process launches N ranks itself (one fresh child process per GPU, spawned BEFORE anything
here touches the GPU, rendezvous on 127.0.0.1) and exits non-zero unless all N report; started under
`python -m torch.distributed.run --nproc-per-node N ...` it is one of the ranks (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*
from the environment).  Either way: one process per GPU, torch.distributed backend "nccl" (= RCCL over xGMI).

One "step" = one query()-equivalent of the reference's training iteration (train.py:126-165) for one blurry view:
pose path (Bezier -> se3_exp_map -> K cameras), ONE fused K-subframe rasterisation, the fused loss-gradient
image (L1 of the pixel-averaged blur + temporal smoothness) + opacity hinge, the full backward to the
per-Gaussian and trajectory gradients, and the iteration's tail (train.py:188-208): densification statistics and
ONE fused Adam launch over all parameter groups (--no-optimizer leaves the tail out; it costs ~0.3 ms).
Densification itself (every 200 iterations in the reference) and data loading are excluded.
Default workload = BASELINE.json's metric configuration: 1M Gaussians, 1920x1080, K=15, curve order 3, SH degree 2.

Sharding over N GPUs:
  --shard views      (default; weak scaling) every rank renders all K subframes of its own view; per-Gaussian and
                     trajectory gradients are averaged with one flat RCCL all-reduce per step.
                     value = N * K * steps / seconds.
  --shard subframes  (strong scaling; the reference's single-view step) the K subframes of ONE view are split over the
                     ranks; partial blur sum + boundary subframes are exchanged before the backward, gradients are
                     summed after it.  value = K * steps / seconds.

Prints ONE JSON line on rank 0 (metric contract + "roofline" for the dominant kernel + "cpu_baseline").
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

EXIT_EXTRAS_HUNG = 3    # a rank's exit code when the headline line was printed but an extra region behind it never returned
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="metric", help="metric | cfg2 | cfg3 | cfg5 | cfg1")
    ap.add_argument("--shard", default="views", choices=["views", "subframes", "mesh"])
    ap.add_argument("--mesh-subframes", type=int, default=2,
                    help="--shard mesh: Gs, the ranks that split ONE view's subframes (rank = v * Gs + s; Gv = gpus / Gs rows "
                         "are a mini-batch of views)")
    ap.add_argument("--K", type=int, default=None)
    ap.add_argument("--sh-degree", type=int, default=2)
    ap.add_argument("--P", type=int, default=None, help="override the number of Gaussians (stress variants)")
    ap.add_argument("--sigma-px", type=float, default=None, help="override the splat size of the generator (default 1.5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-lists", action="store_true",
                    help="skip the second timed region with tile_cull=0 (the reference's duplicate lists)")
    ap.add_argument("--lambda-t", type=float, default=1e-3)
    ap.add_argument("--autograd-path", action="store_true",
                    help="run the step through CameraMotionModule.query + torch autograd instead of the fused step")
    ap.add_argument("--ar-chunks", type=int, default=4,
                    help="N > 1 GPUs: all-reduce the per-Gaussian gradient bucket in this many Gaussian-index chunks on a side "
                         "stream, each as soon as the backward has produced it (1 = one collective after the backward)")
    ap.add_argument("--allreduce", default=None, choices=["collective", "p2p"],
                    help="N > 1 GPUs: how the gradient bucket is summed -- the backend's all-reduce (default) or a direct "
                         "reduce-scatter + all-gather over point-to-point sends (deblurgs_amd.sharding.p2p_allreduce_)")
    ap.add_argument("--no-graph", action="store_true",
                    help="enqueue every step eagerly instead of replaying the captured hipGraph")
    ap.add_argument("--graph-always", action="store_true",
                    help="replay the captured hipGraph also for large views (default: TrainingLoop's own policy, graph='auto': "
                         "views whose compositing backward the library runs in parts when enqueued eagerly stay eager)")
    ap.add_argument("--emulate-shard", default=None, metavar="r/G",
                    help="ONE GPU: time rank r's share of a G-GPU step of --shard mode with every collective degenerate "
                         "(one-rank process group): 'subframes' = its slice of the view's K subframes, 'views' = a whole "
                         "view through the sharded code path.  Prints an 'emulated_shard' line, not the metric line; "
                         "tools/predict_scaling.py turns these into DESIGN.md's predicted scaling table")
    ap.add_argument("--extras-mesh", action="store_true",
                    help="N >= 4 GPUs: also time the (N/2) x 2 mesh among the extra regions (off by default: its row groups "
                         "have never met RCCL on real hardware, and an extra region that hangs costs the run its exit code 0)")
    ap.add_argument("--no-extras", action="store_true",
                    help="N > 1 GPUs: skip the extra regions behind the headline one (other sharding mode, collective vs "
                         "point-to-point all-reduce A/B)")
    ap.add_argument("--no-optimizer", action="store_true",
                    help="time query + loss + backward (+ all-reduce) only, without densification stats and Adam")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ N-rank launcher
def count_gpus_without_hip():
    """GPUs of this node from the KFD topology in sysfs -- the parent of an N-rank run never calls into the HIP runtime
    (a process that initialised the GPU must not be the one that spawns or re-execs).  None if sysfs has no answer."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        if os.environ.get(var, "").strip():
            return len([x for x in os.environ[var].split(",") if x.strip()])
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for d in os.listdir(base):
            with open(os.path.join(base, d, "properties")) as f:
                props = dict(ln.split()[:2] for ln in f if len(ln.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:      # CPU nodes have simd_count 0
                n += 1
        return n
    except (OSError, ValueError):
        return None


def launch_ranks(args):
    """Parent of an N-GPU run: spawns N fresh rank processes and relays rank 0's JSON line.  Makes NO GPU call itself
    (devices are counted through sysfs), never re-execs.  All children are watched: the first non-zero exit or the
    overall timeout (DGS_BENCH_TIMEOUT_S, default 1500 s) terminates the others and the launcher returns non-zero."""
    n = args.gpus
    one_device = os.environ.get("DGS_DIST_ONE_DEVICE", "0") == "1"
    have = count_gpus_without_hip()
    if have is not None and have < n and not one_device:
        print(f"bench.py: --gpus {n} requested but only {have} GPU(s) are visible; refusing to report a smaller run as "
              f"n_gpus={n}", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    import tempfile
    procs = []
    out0 = tempfile.TemporaryFile(mode="w+")       # rank 0's stdout: a file, so that no pipe can fill up while we poll
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        env.setdefault("DGS_BENCH_JOB", str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL, text=True))
    deadline = time.time() + float(os.environ.get("DGS_BENCH_TIMEOUT_S", "1500"))
    failed = None
    while True:
        rcs = [p.poll() for p in procs]
        if all(rc is not None for rc in rcs):
            break
        # (EXIT_EXTRAS_HUNG: that rank's headline figures are valid and its peers will give up on the same region by themselves)
        bad = [i for i, rc in enumerate(rcs) if rc not in (None, 0, EXIT_EXTRAS_HUNG)]
        if bad or time.time() > deadline:
            failed = f"rank(s) {bad} exited non-zero" if bad else "timeout"
            for p in procs:                     # our own children, by handle (never by pattern)
                if p.poll() is None:
                    p.terminate()
            t_kill = time.time() + 10
            for p in procs:
                try:
                    p.wait(timeout=max(0.1, t_kill - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
            rcs = [p.returncode for p in procs]
            break
        time.sleep(0.2)
    out0.seek(0)
    line = None
    for ln in out0.read().splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if failed is not None or any(rc not in (0, EXIT_EXTRAS_HUNG) for rc in rcs) or line is None:
        why = failed or ("no result line" if line is None else "non-zero exit")
        print(f"bench.py: {why}; rank exit codes {rcs}", file=sys.stderr)
        return 1
    if json.loads(line)["n_gpus"] != n:
        print(f"bench.py: the ranks saw {json.loads(line)['n_gpus']} peers, not {n}", file=sys.stderr)
        return 1
    print(line, flush=True)
    if any(rc == EXIT_EXTRAS_HUNG for rc in rcs):    # headline valid, an extra region hung: the line is out, the code says so
        print(f"bench.py: an extra region hung on rank(s) {[i for i, rc in enumerate(rcs) if rc]}; headline line printed",
              file=sys.stderr)
        return EXIT_EXTRAS_HUNG
    return 0


# ------------------------------------------------------------------------------------------------ byte model
def stage_bytes(P, Pv_tot, R_tot, N, K, s):
    """Algorithmic bytes per K-fused launch of each stage (SURVEY.md 8d byte model, summed over the K subframes;
    K-fused: the P*(44+12s) input read and the per-Gaussian gradient write are paid once per launch)."""
    return {
        "preprocess": P * (44 + 12 * s) + 84 * Pv_tot,
        "scan": 8 * P * K,
        "duplicate": 20 * P * K + 12 * R_tot,
        "sort": 24 * R_tot,
        # tile ranges by binary search over the sorted keys: per tile (N / 256 of them per subframe) one 8-byte result
        # and ~log2(R) 8-byte probes; the reference's one-thread-per-key sweep reads 8 R bytes (DGS_RANGES_SWEEP)
        "ranges": (N * K / 256.0) * (8 + 8 * max(math.log2(max(R_tot, 2)), 1.0)),
        "composite_fwd": (28 + 16) * R_tot + 24 * N * K,
        "composite_bwd": 44 * R_tot + 24 * N * K + 2 * 48 * Pv_tot,
        "geometry_bwd": Pv_tot * (100 + 12 * s + 48) + P * (40 + 12 * s),
        # second stage of the atomics-free reduction (48-byte contribution rows -> per-pair totals): work the reference does
        # with atomics inside its render backward, so SURVEY 8d's model gives it no bytes of its own (the 2 * 48 Pv term of
        # composite_bwd is the algorithmic cost of those sums); the stage's own traffic is 48 R + 64 Pv
        "contrib_reduce": 0,
        "depth_order": 24 * P * K,   # ideal one-read-one-write sort of the K*P (key, index) pairs
        # tile_cull: the per-slot test in natural order (tiles_touched in, 16-byte record + count out per pair, 36 B of
        # each visible pair's row).  K * P <= 2^24: the counts ride through the depth sort and only their scan follows
        # (timed as "scan"); beyond that they are gathered into depth order (order, flag, count in, count out) and scanned
        # inside this stage
        "tile_cull": (4 + 20) * P * K + 36 * Pv_tot + (0 if P * K <= (1 << 24) else 16 * P * K + 8 * P * K),
    }


# ------------------------------------------------------------------------------------------------ CPU baselines
def cpu_baseline_port(scene, k):
    """The CPU oracle (OpenMP build, all host cores) timed on ONE of the K subframes of the same workload."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import oracle
    from helpers import oracle_forward
    threads = oracle.use_openmp(True)
    try:
        g = np.random.default_rng(0).normal(size=(3, scene["H"], scene["W"])).astype(np.float32)
        t0 = time.time()
        st = oracle_forward(scene, k)
        oracle.backward(st, g)
        dt = time.time() - t0
    finally:
        oracle.use_openmp(False)
    return {"value": 1.0 / dt, "unit": "subframe-renders/sec (fwd+bwd)", "cores": int(threads), "kind": "port",
            "sample": f"1 of the K={scene['K']} subframes (k={k}) of this workload, forward+backward, "
                      f"oracle/dgs_oracle.cpp with OpenMP; {dt:.1f} s", "seconds": dt}


def cpu_baseline_torch_naive(scene, k, side=96):
    """The baseline north_star names: a naive dense PyTorch CPU rasteriser (oracle/torch_naive.py: every pixel against
    every candidate Gaussian, autograd backward) on the box's own host cores.  Dense N x P is out of reach at these
    sizes, so the sample is a side x side pixel window at the image centre of subframe k with the Gaussians whose
    tile rectangles meet it; the measured figure is windows/sec, and the full-frame equivalent divides by the stated
    pixel ratio (an extrapolation, labelled as such)."""
    import numpy as np
    import torch
    from oracle import torch_naive
    # torch's CPU kernels on these small dense tiles get SLOWER beyond a few dozen threads (143 s with 128 threads
    # against 9 s with 8 for the same sample, tools/naive_threads.py); the count used is stated in `cores`
    threads = int(os.environ.get("DGS_NAIVE_THREADS", min(os.cpu_count() or 1, 16)))
    torch.set_num_threads(threads)
    W, H = scene["W"], scene["H"]
    x0, y0 = (W - side) // 2 // 16 * 16, (H - side) // 2 // 16 * 16
    window = (x0, y0, x0 + side, y0 + side)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    leaves = {n: t(scene[n]).requires_grad_(True) for n in ("means3D", "opacities", "sh", "scales", "rotations")}
    g = torch.from_numpy(np.random.default_rng(0).normal(size=(3, side, side)).astype(np.float32))
    t0 = time.time()
    color, depth, radii = torch_naive.rasterize(
        leaves["means3D"], leaves["opacities"], t(scene["viewmatrix"][k]), t(scene["projmatrix"][k]),
        t(scene["campos"][k]), t(scene["bg"]), W, H, scene["tanfovx"], scene["tanfovy"], sh=leaves["sh"],
        scales=leaves["scales"], rotations=leaves["rotations"], sh_degree=scene["sh_degree"], pixel_chunk=2048,
        window=window)
    (color * g).sum().backward()
    dt = time.time() - t0
    ratio = (W * H) / float(side * side)
    return {"value": 1.0 / dt, "unit": f"{side}x{side}-pixel windows/sec (fwd+bwd)", "cores": int(threads),
            "kind": "port", "implementation": "oracle/torch_naive.py (dense PyTorch CPU rasteriser, autograd backward)",
            "sample": f"pixel window {window} of subframe k={k} of this workload (1/{ratio:.0f} of the frame's pixels; "
                      f"all {scene['P']} Gaussians preprocessed, those meeting the window composited); {dt:.1f} s",
            "seconds": dt,
            "full_frame_equivalent_renders_per_sec_extrapolated": 1.0 / (dt * ratio)}


# ------------------------------------------------------------------------------------------------ RCCL's own log
NCCL_ALGOS = {0: "Tree", 1: "Ring", 2: "CollNetDirect", 3: "CollNetChain", 4: "NVLS", 5: "NVLSTree", 6: "PAT"}
NCCL_PROTOS = {0: "LL", 1: "LL128", 2: "Simple"}


def rccl_log_setup(rank):
    """"verify which algorithm RCCL picks".  BEFORE init_process_group (environment only, no exec): this rank's
    RCCL log (INFO; init, collective calls, the tuner's algorithm / protocol choices) goes to a file of its own, which
    rank 0 parses after the run (parse_rccl_log).  A log level, subsystem list or file the caller already set wins."""
    if os.environ.get("DGS_DIST_BACKEND", "nccl") != "nccl" or os.environ.get("DGS_BENCH_NO_RCCL_LOG", "0") == "1":
        return None
    import tempfile
    job = os.environ.get("DGS_BENCH_JOB") or os.environ.get("MASTER_PORT", "0")
    d = os.path.join(tempfile.gettempdir(), f"dgs_rccl_{job}")
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, f"rank{rank}.log")
    if os.environ.get("NCCL_DEBUG", "").upper() not in ("INFO", "TRACE"):    # (the pool's image presets NCCL_DEBUG=VERSION)
        os.environ["NCCL_DEBUG"] = "INFO"
    os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,COLL,TUNING")
    os.environ.setdefault("NCCL_DEBUG_FILE", path)
    return os.environ["NCCL_DEBUG_FILE"]


def parse_rccl_log(path, max_lines=10):
    """What the library said about itself: version, the rings / trees / channels it built, and -- per message size -- the
    algorithm and protocol its tuner chose ("<n> Bytes -> Algo <a> proto <p> ...", numbers or names depending on the
    release).  Checked against a log of this image's RCCL 2.26 (one rank: no tuning lines) and, for the tuning lines, a
    log in NCCL's format (tests/test_host_logic.py)."""
    import re
    if not path or not os.path.exists(path):
        return None
    try:
        text = open(path, errors="replace").read()
    except OSError:
        return None
    out = {"file": path, "bytes_logged": len(text)}
    m = re.search(r"\b(RCCL|NCCL) version\s*:?\s*([^\n]+)", text)
    if m:
        out["version"] = (m.group(1) + " " + m.group(2)).strip()[:120]
    chosen = {}
    for nb, a, pr in re.findall(r"(\d+) Bytes -> Algo (\w+) proto (\w+)", text):
        key = int(nb)
        an = NCCL_ALGOS.get(int(a), a) if a.isdigit() else a.capitalize()
        pn = NCCL_PROTOS.get(int(pr), pr) if pr.isdigit() else {"SIMPLE": "Simple"}.get(pr.upper(), pr.upper())
        name = f"{an}/{pn}"
        chosen.setdefault(key, {})
        chosen[key][name] = chosen[key].get(name, 0) + 1
    if chosen:      # the largest messages are the gradient bucket (or its chunks)
        out["algo_proto_by_message_bytes"] = {str(k): chosen[k] for k in sorted(chosen, reverse=True)[:6]}
    calls = {}
    for name in re.findall(r"\b(AllReduce|Broadcast|AllGather|ReduceScatter|Send|Recv): opCount", text):
        calls[name] = calls.get(name, 0) + 1
    if calls:
        out["calls_logged"] = calls
    lines = text.splitlines()
    pick = lambda rx, n: [ln.strip()[-200:] for ln in lines if re.search(rx, ln)][:n]
    chan = re.findall(r"Channel \d+/(\d+) *:", text)
    if chan:
        out["channels"] = int(chan[0])
    topo = (pick(r"nRanks \d+", 1) + pick(r"Init COMPLETE", 1) + pick(r"Connected all (rings|trees)", 2) +
            pick(r"\bRing \d+ *:", 2) + pick(r"Trees? \[", 1) + pick(r"Channel \d+/\d+ *:", 2) + pick(r"via P2P|P2P/IPC|P2P/direct", 2))
    if topo:
        out["init_lines"] = topo[:max_lines]
    return out


# ------------------------------------------------------------------------------------------------ one rank
def run_rank(args):
    import numpy as np
    import torch
    import torch.distributed as dist
    from deblurgs_amd import _lib, losses, sharding, synthetic
    from deblurgs_amd import diff_gaussian_rasterization as dgr
    from deblurgs_amd.cloud import GaussianCloud
    from deblurgs_amd.motion import CameraMotionModule, RefCamera
    from deblurgs_amd.training import default_optimization_params

    if args.allreduce is not None:
        sharding.ALLREDUCE_MODE = args.allreduce
    emu = None
    if args.emulate_shard:
        # one GPU, a ONE-rank process group (the real backend), the sharded step of rank r of G with degenerate collectives
        er, eg = (int(x) for x in args.emulate_shard.split("/"))
        if not (0 <= er < eg) or args.gpus != 1 or "RANK" in os.environ:
            raise SystemExit("bench.py: --emulate-shard r/G needs 0 <= r < G, --gpus 1 and a plain (single-process) start")
        emu = (er, eg)
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            free_port = sk.getsockname()[1]
        os.environ.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port),
                          DGS_DIST_FORCE_INIT="1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rccl_log = None
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 or emu is not None:
        rccl_log = rccl_log_setup(int(os.environ.get("RANK", "0")))
    rank, world_env, local_rank = sharding.init_distributed("cuda")
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world != max(args.gpus, 1):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but {world} rank(s) joined the process group")
    if os.environ.get("DGS_DIST_ONE_DEVICE", "0") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:      # count the peers through the collective library itself
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        assert int(ones.item()) == world, "all-reduce did not reach every rank"
    # "mesh" (round 6): Gv x Gs ranks, rank = v * Gs + s -- row v's Gs ranks split view v's subframes, the rows are a batch
    mesh_dims = None
    if args.shard == "mesh" and world > 1 and emu is None:
        gs_ = max(1, min(args.mesh_subframes, world))
        if world % gs_ != 0:
            raise SystemExit(f"bench.py: --shard mesh needs --gpus divisible by --mesh-subframes ({world} / {gs_})")
        mesh_dims = (world // gs_, gs_)
    if args.shard == "mesh" and mesh_dims is None:
        raise SystemExit("bench.py: --shard mesh needs --gpus > 1 (and no --emulate-shard)")
    # slices of ONE view per rank ("subframes", and the rows of a mesh): what the per-rank byte model and the probe see
    subframes_mode = (world > 1 or emu is not None) and args.shard in ("subframes", "mesh")
    shard_id = emu if emu is not None else ((rank % mesh_dims[1], mesh_dims[1]) if mesh_dims else (rank, world))
    views_per_step = (mesh_dims[0] if mesh_dims else (1 if subframes_mode else world))     # views rendered per step, whole job

    over = {} if args.K is None else {"K": args.K}
    if args.P is not None:
        over["P"] = args.P
    if args.sigma_px is not None:
        over["sigma_px"] = args.sigma_px
    scene = synthetic.make_config(args.config, seed=0, sh_degree=args.sh_degree, **over)
    P, W, H, K = scene["P"], scene["W"], scene["H"], scene["K"]
    C = synthetic.CONFIGS[args.config]["C"]
    cloud = GaussianCloud.from_scene(scene, dev)
    ref_cam = RefCamera(W, H, scene["FoVx"], scene["FoVy"], device=dev)
    def make_view(view_seed):
        gen = torch.Generator(device="cpu").manual_seed(1234 + view_seed)
        gt_ = torch.rand((1, 3, H, W), generator=gen).to(dev)
        mo = CameraMotionModule(ref_cam, gt_, curve_order=C, num_subframes=K, device=dev)
        traj = synthetic.make_trajectory(K, C, scene["projection_matrix"], seed=view_seed)
        with torch.no_grad():
            mo._trans._control_points.copy_(torch.from_numpy(traj["ctrl_trans"])[None].to(dev))
            mo._rot._control_points.copy_(torch.from_numpy(traj["ctrl_rot"])[None].to(dev))
        mo.link_gaussian(cloud)
        return mo

    # "subframes": every rank works on the SAME view; "views": rank g on its own
    motion = make_view((rank // mesh_dims[1]) if mesh_dims else (0 if subframes_mode else (rank if emu is None else emu[0])))
    params = cloud.hot_parameters()
    # The timed step is the PRODUCT's training iteration, deblurgs_amd.training.TrainingLoop.step (train.py:104-208):
    # scheduled hyper-parameters, the view's K subframes rendered and back-propagated (by default through
    # deblurgs_amd.fused_step.FusedStep: no autograd graph, duplicate arrays sized ahead, no host synchronisation;
    # --autograd-path selects CameraMotionModule.query + losses + loss.backward()), the opacity hinge, the gradient
    # all-reduce, densification statistics and ONE fused Adam launch over the six per-Gaussian groups and the trajectory
    # groups, reference learning rates (arguments/__init__.py:84-123).  densify_and_prune itself never fires
    # (densify_from_iter is out of reach) and --no-optimizer drops statistics + Adam.
    from deblurgs_amd.training import TrainingLoop
    far = 10 ** 9
    opt = default_optimization_params(iterations=far if not args.no_optimizer else 0, lambda_t_smooth_init=args.lambda_t,
                                      lambda_t_smooth_final=args.lambda_t, lambda_hinge=0.1, curve_start_iter=1,
                                      curve_end_iter=far, densify_from_iter=far,
                                      densify_until_iter=far if not args.no_optimizer else 0,
                                      opacity_reset_interval=far, curve_rotation_lr=1e-3, curve_controlpoints_lr=1e-2,
                                      curve_alignment_lr=0.0)
    LR_SCALE = 1e-6

    def make_loop(mode_, motion_, mesh_=None):
        lp = TrainingLoop(cloud, motion_, opt, cameras_extent=1.0, spatial_lr_scale=1.0, distributed=mode_, mesh=mesh_,
                          fused_step=False if args.autograd_path else "auto", log_losses=False,
                          graph=False if args.no_graph else ("always" if args.graph_always else "auto"),
                          ar_chunks=args.ar_chunks,
                          emulate_shard=emu if mode_ else None)
        # The ground truth is noise, so real learning rates would pull the cloud away from the configured workload within
        # the timed region (opacities collapse and the step gets ~5 % cheaper).  The Adam kernel does the same work for
        # any learning rate; scale the rates down so that every timed step renders the workload BASELINE.json names.
        for group in cloud.optimizer.param_groups:
            group["lr"] *= LR_SCALE
        cloud.xyz_scheduler_args = lambda it: 0.00016 * LR_SCALE
        if (world > 1 or emu is not None) and lp._fused is not None:
            lp._fused.time_allreduce = True
        return lp

    mode = args.shard if (world > 1 or emu is not None) else False
    loop = make_loop(mode, motion, mesh_dims)

    stats = {"it": 0}
    ar_events = []
    if world > 1:      # time the gradient all-reduce inside the product's step
        _flat = sharding.flat_allreduce_grads

        def timed_allreduce(*a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = _flat(*a, **kw)
            e1.record()
            ar_events.append((e0, e1))
            return r
        sharding.flat_allreduce_grads = timed_allreduce

    def step():
        stats["it"] += 1
        loop.step(stats["it"], 0)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(nsteps, profile, step_fn=None):
        step_fn = step_fn or step
        sync()
        if profile:
            _lib.profile_reset()
            _lib.profile_enable(True)
        ar_events.clear()
        sync()
        t0 = time.time()
        for _ in range(nsteps):
            step_fn()
        torch.cuda.synchronize()
        stats["dt_local"] = time.time() - t0       # this rank's own time, before it waits for the others
        sync()
        dt = time.time() - t0
        if profile:
            _lib.profile_enable(False)
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    for _ in range(args.warmup):
        step()
    sync()
    with torch.no_grad():       # measured Pv of this rank's workload (an output of the forward)
        probe = motion.query(0, "all", compute_blurred=False, shard=shard_id if subframes_mode else None)
        Pv_tot = int((probe["radii_all"] > 0).sum().item())
        del probe
    # (N ranks: the step up to its first collective is replayed -- FusedStep.replay_front)
    replayed_before = loop._fused.replayed if loop._fused is not None else 0
    # the timed region never carries the stage timers (their event pairs, and -- csrc/api.hip -- the compositing backward
    # as one launch instead of parts); the per-stage durations come from a second region right after it
    # the region is run three times back to back (one region of 20 steps is 0.2 s, and box-to-box noise is as
    # large as a round's gain); every region is exactly args.steps steps between barriers; the headline is the MEDIAN region
    region_dts = [timed(args.steps, profile=False) for _ in range(1 if emu is not None else 3)]
    dt = sorted(region_dts)[len(region_dts) // 2]
    per_rank = None
    if world > 1:       # every rank's own time for the region (the headline divides by the slowest)
        mine = torch.tensor([stats["dt_local"]], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        ms = [float(x.item()) / args.steps * 1e3 for x in every]
        per_rank = {"min_ms_per_step": round(min(ms), 3), "max_ms_per_step": round(max(ms), 3),
                    "ms_per_step_by_rank": [round(x, 3) for x in ms]}
    allreduce_ms = None
    if ar_events:
        allreduce_ms = sum(a.elapsed_time(b) for a, b in ar_events) / len(ar_events)
    elif world > 1 and loop._fused is not None and loop._fused.ar_events:
        # chunked reduction: the span on the side stream from "first chunk ready" to "last chunk reduced" (it overlaps
        # the backward's tail, so it is not additional step time)
        evs = loop._fused.ar_events[-args.steps:]
        allreduce_ms = sum(a.elapsed_time(b) for a, b in evs) / len(evs)
    graph_info = None
    # did the timed region replay the captured step?  (TrainingLoop(graph="auto") leaves large views to the eager fused step)
    replaying = loop._fused is not None and loop._fused.replayed > replayed_before
    if replaying:
        # The timed region replayed the captured step (one hipGraph launch per iteration): HIP events cannot be recorded
        # between the kernels of a graph, so the per-stage durations come from a second, EAGER region of the same step
        # (same kernels, same arguments) right after it.
        graph_info = {"captured": loop._fused.captured, "replayed": loop._fused.replayed}
        loop.graph = False
        for _ in range(2):
            step()
        n_prof = max(10, min(args.steps, 30))
        dt_eager = timed(n_prof, profile=True)
        stats["profiled_steps"] = n_prof
        graph_info["eager_ms_per_step"] = round(dt_eager / n_prof * 1e3, 3)
        loop.graph = True
    else:
        n_prof = max(10, min(args.steps, 30))
        dt_prof = timed(n_prof, profile=True)
        stats["profiled_steps"] = n_prof
        stats["profiled_ms_per_step"] = round(dt_prof / n_prof * 1e3, 3)
    prof = _lib.profile_read()

    # the same step on the reference's duplicate lists (tile_cull = 0: sort keys / point lists bit-identical to the
    # reference's), reported beside the headline value
    # ---- N > 1 only: what makes ONE multi-GPU run decisive -- the other sharding mode's value and a
    # collective-vs-point-to-point A/B of the gradient bucket's all-reduce, in the same invocation on the same ranks.
    # They run AFTER rank 0 has assembled the headline's result and under a watchdog: should an extra region never return
    # (a collective path that has never met this node's RCCL), the line still goes out with the headline in it.
    def run_extras(extras):
        if rank == 0:   # for the record, should an extra region never return: the headline region's figures, on stderr
            print(json.dumps({"headline_before_extras": {
                "n_gpus": world, "sharding": args.shard, "ms_per_step": round(dt / args.steps * 1e3, 3),
                "value": round(views_per_step * K * args.steps / dt, 2)}}), file=sys.stderr, flush=True)
        n_bucket = sum(p.numel() for p in params)

        def ab_allreduce(kind, iters=8):
            buf = torch.full((n_bucket,), 1.0 / 1024.0, dtype=torch.float32, device=dev)
            keep_mode = sharding.ALLREDUCE_MODE
            sharding.ALLREDUCE_MODE = kind
            try:
                for _ in range(2):
                    sharding._allreduce(buf, True, None)
                sync()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    sharding._allreduce(buf, True, None)      # (a mean: the values stay put however often it runs)
                e1.record()
                torch.cuda.synchronize()
                t = torch.tensor([e0.elapsed_time(e1) / iters], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                ok = bool(torch.all(buf == buf[0]).item()) and abs(float(buf[0].item()) - 1.0 / 1024.0) < 1e-9
            finally:
                sharding.ALLREDUCE_MODE = keep_mode
            ms_ = float(t.item())
            nbytes = 4 * n_bucket
            return {"ms": round(ms_, 4), "bytes": nbytes, "algbw_GBps": round(nbytes / (ms_ * 1e-3) / 1e9, 1),
                    "busbw_GBps": round(2.0 * (world - 1) / world * nbytes / (ms_ * 1e-3) / 1e9, 1), "values_ok": ok}
        try:
            extras["allreduce_ab"] = {"collective": ab_allreduce("collective"), "p2p": ab_allreduce("p2p"),
                                      "note": "the whole gradient bucket (P * (11 + 3 M) floats), in-place mean, alone on the "
                                              "stream: HIP events around 8 back-to-back calls, max over ranks; busbw = "
                                              "2 (G-1)/G * bytes / time"}
        except Exception as ex:            # (never lose the headline line to a secondary region)
            extras["allreduce_ab"] = {"error": repr(ex)}
        # every sharding mode but the headline's, timed on the same ranks (the mesh: Gv x 2 when the ranks allow it)
        others = [m_ for m_ in ("views", "subframes") if m_ != args.shard]
        if args.extras_mesh and world >= 4 and world % 2 == 0 and args.shard != "mesh":
            others.append("mesh")
        extras["other_modes"] = {}
        for other in others:
            try:
                if loop._fused is not None:
                    loop._fused._poll(block=True)
                    loop._fused.invalidate()           # the headline loop's graphs and pool go back to the driver
                dims2 = (world // 2, 2) if other == "mesh" else None
                motion2 = make_view(0 if other == "subframes" else (rank // 2 if other == "mesh" else rank))
                loop2 = make_loop(other, motion2, dims2)
                it2 = {"it": 10 ** 6}

                def step2():
                    it2["it"] += 1
                    loop2.step(it2["it"], 0)
                for _ in range(max(args.warmup, 4)):
                    step2()
                n2 = max(10, args.steps // 4)
                dt2 = timed(n2, profile=False, step_fn=step2)
                views2 = {"subframes": 1, "views": world, "mesh": world // 2}[other]
                extras["other_modes"][other] = {
                    "sharding": other if dims2 is None else f"mesh {dims2[0]} views x {dims2[1]} subframe slices",
                    "scaling": {"subframes": "strong", "views": "weak", "mesh": "mesh"}[other], "steps": n2,
                    "value": round(views2 * K * n2 / dt2, 2), "views_per_step": views2,
                    "ms_per_step": round(dt2 / n2 * 1e3, 3),
                    "graph": None if loop2._fused is None else {"captured": loop2._fused.captured,
                                                                "replayed": loop2._fused.replayed},
                    "note": {"subframes": "strong scaling: ONE view per step, its K subframes split over the GPUs "
                                          "(BASELINE.json cfg4's wording; the reference's single-view step exactly)",
                             "views": "weak scaling: one view per GPU per step (a G-view mini-batch)",
                             "mesh": "G/2 views per step, each view's K subframes split over two GPUs (loss exchange inside "
                                     "the pair, gradient bucket summed over all GPUs / number of views)"}[other]}
                del loop2, motion2
            except Exception as ex:
                extras["other_modes"][other] = {"sharding": other, "error": repr(ex)}
        sync()

    # ---- the same step doing the REFERENCE's render work: the headline step leaves out what the
    # training loss never reads -- the depth image (the reference composites it with every render, forward.cu:373,390, and
    # query() returns it, scene/motion.py:145-158) -- and builds tile-culled lists.  Three more regions put the values with
    # that work next to the headline: with depth, on the reference's lists, and both (the worst case).
    def side_region(with_depth, reference_lists, note):
        fs = loop._fused
        if fs is None:
            return None
        keep_cull, keep_depth = dgr.TILE_CULL, fs.always_depth
        fs._poll(block=True)
        fs.invalidate()                     # other lists / other outputs: learn the counts afresh, drop captured steps
        dgr.TILE_CULL = keep_cull and not reference_lists
        fs.always_depth = bool(with_depth)
        try:
            for _ in range(5):
                step()
            n0 = max(10, args.steps // 4)
            # two regions, the faster one reported: the first region after a switch has been seen to carry a one-off stall
            # of tens of milliseconds (a kernel variant's first launch, the allocator re-growing after the invalidate)
            dt0 = min(timed(n0, profile=False), timed(n0, profile=False))
            out = {"value": round(K * n0 / dt0, 2), "ms_per_step": round(dt0 / n0 * 1e3, 3), "steps": n0,
                   "depth_output": bool(with_depth), "tile_cull": bool(dgr.TILE_CULL), "note": note,
                   "dropped_steps": fs.dropped, "retried_steps": loop.retried}
            if os.environ.get("DGS_BENCH_DEBUG_STEPS"):
                ts = []
                for _ in range(8):
                    torch.cuda.synchronize()
                    t_ = time.time()
                    step()
                    torch.cuda.synchronize()
                    ts.append(round((time.time() - t_) * 1e3, 2))
                out["debug_synced_step_ms"] = ts
            return out
        finally:
            dgr.TILE_CULL, fs.always_depth = keep_cull, keep_depth
            fs._poll(block=True)
            fs.invalidate()

    ref_lists = with_depth = ref_lists_depth = None
    if world == 1 and emu is None and not args.no_reference_lists and dgr.TILE_CULL and not args.autograd_path:
        with_depth = side_region(True, False, "the headline step + the K depth images composited and stored "
                                              "(composite_fwd_kernel<true, true>), as every render of the reference does")
        ref_lists = side_region(False, True, "same step with DgsProblem.tile_cull = 0: the duplicate lists, sort keys and "
                                             "tile ranges are the reference's bit for bit; images and gradients are bitwise "
                                             "equal to the headline run's (tests/test_gpu_configs.py)")
        ref_lists_depth = side_region(True, True, "the reference's lists AND the depth images: every piece of render work "
                                                  "the reference does per subframe (the worst case for this line)")

    # ---- the reference's second use of the operator: inference (test.py:117, render_spiral.py:29 call render() under
    # no_grad).  The K subframes of the view through gaussian_renderer.render_subframes with nothing that can receive a
    # gradient: DgsProblem.forward_only = 1 (no final_T / n_contrib / cov3D / mask stores), the exact two-phase forward
    # with its one host read per call, depth images included (render() returns them).
    fwd_only = None
    if world == 1 and emu is None and not args.autograd_path:
        from deblurgs_amd import gaussian_renderer
        with torch.no_grad():
            wv_, fp_, cc_ = (t.contiguous() for t in motion.get_trajectory_matrices(0))
            bg_ = torch.zeros(3, device=dev)
            for _ in range(3):
                gaussian_renderer.render_subframes(wv_, fp_, cc_, ref_cam, cloud, bg_)
            nf = max(10, args.steps // 4)
            sync()
            t0 = time.time()
            for _ in range(nf):
                gaussian_renderer.render_subframes(wv_, fp_, cc_, ref_cam, cloud, bg_)
            torch.cuda.synchronize()
            dtf = time.time() - t0
            # ... and the reference's own inference call shape: render() on ONE camera (K = 1), one call per frame
            from deblurgs_amd.pose import MiniCam
            cams1 = [MiniCam(ref_cam.image_width, ref_cam.image_height, ref_cam.FoVy, ref_cam.FoVx, ref_cam.znear,
                             ref_cam.zfar, wv_[k_], fp_[k_], cc_[k_]) for k_ in range(K)]
            for k_ in range(min(3, K)):
                gaussian_renderer.render(cams1[k_], cloud, bg_)
            n1 = max(15, args.steps // 2)
            sync()
            t0 = time.time()
            for i_ in range(n1):
                gaussian_renderer.render(cams1[i_ % K], cloud, bg_)
            torch.cuda.synchronize()
            dt1 = time.time() - t0
        fwd_only = {"value": round(K * nf / dtf, 2), "unit": "subframe-renders/sec (forward only)",
                    "ms_per_call": round(dtf / nf * 1e3, 3), "calls": nf,
                    "note": "K subframes per call, colour + depth images, DgsProblem.forward_only = 1, tile culling as the "
                            "headline, the exact two-phase forward (one host read of the duplicate count per call)",
                    "single_camera": {"value": round(n1 / dt1, 2), "unit": "render() calls / s (K = 1, forward only)",
                                      "ms_per_call": round(dt1 / n1 * 1e3, 3), "calls": n1,
                                      "note": "gaussian_renderer.render(camera, cloud, bg) under no_grad, the call of the "
                                              "reference's test.py:117 / render_spiral.py:29, once per frame: the cloud's "
                                              "raw parameters go to the kernels (no getter launches, no SH concat), "
                                              "two-phase forward with its host read; GPU-bound (about 30 small launches)"}}

    if rank == 0:
        # R from a state-level forward (the operator keeps it in its autograd ctx)
        with torch.no_grad():
            wv, fp, cc = motion.get_trajectory_matrices(0)
            rs = dgr.GaussianRasterizationSettings(H, W, scene["tanfovx"], scene["tanfovy"],
                                                   torch.from_numpy(scene["bg"]).to(dev), 1.0, 0.2, 100.0, False,
                                                   args.sh_degree, cc, False, False)
            R_tot = dgr._forward_impl(K, cloud.get_xyz.detach(), cloud.get_features.detach().contiguous(), None,
                                      cloud.get_opacity.detach().reshape(-1), cloud.get_scaling.detach(),
                                      cloud.get_rotation.detach(), None, wv.contiguous(), fp.contiguous(),
                                      cc.contiguous(), rs)[0]
        N = W * H
        s = (args.sh_degree + 1) ** 2
        frac_k = 1.0
        if subframes_mode:       # this rank's share of the K subframes
            k0, k1 = sharding.shard_range(K, shard_id[0], shard_id[1])
            frac_k = (k1 - k0) / K
        bytes_by_stage = stage_bytes(P, Pv_tot, R_tot * frac_k, N, K * frac_k, s)
        stages = {}
        for name, (ms, calls) in prof.items():
            if calls == 0:
                continue
            avg_ms = ms / calls
            gbs = bytes_by_stage[name] / (avg_ms * 1e-3) / 1e9
            # model_GBps: the byte MODEL of SURVEY 8d over the measured time -- not achieved bandwidth (for geometry_bwd the
            # model counts the K-summed Pv reads the fused kernel does not repeat); traffic_* (below) is counter-based
            stages[name] = {"avg_ms": round(avg_ms, 4), "launches": calls, "alg_bytes": int(bytes_by_stage[name]),
                            "model_GBps": round(gbs, 1), "ms_per_step": round(ms / max(stats["profiled_steps"], 1), 4),
                            "traffic_bytes": None, "traffic_GBps": None}
            if name == "contrib_reduce":
                moved = 48 * R_tot * frac_k + 64 * Pv_tot
                stages[name]["moved_bytes"] = int(moved)
                stages[name]["moved_GBps"] = round(moved / (avg_ms * 1e-3) / 1e9, 1)
        # counter-based HBM bytes per step and stage: the committed FETCH_SIZE / WRITE_SIZE passes of this same command
        # (profiles/traffic_rNN.json, `_per_step`; PMC counters cannot be collected from inside this process), over THIS
        # run's stage times -- same provenance rule as roofline.traffic
        traffic_doc, traffic_file = None, None
        import glob as _glob
        for path in sorted(_glob.glob(os.path.join(ROOT, "profiles", "traffic_r[0-9][0-9].json")))[::-1]:
            try:
                doc = json.load(open(path))
            except Exception:
                continue
            if doc.get("_config") == args.config and "_per_step" in doc and world == 1 and emu is None \
                    and args.sh_degree == doc.get("_sh_degree", 2):
                traffic_doc, traffic_file = doc, os.path.basename(path)
                break
        if traffic_doc is not None:
            for name, nbytes in traffic_doc["_per_step"].items():
                if name in stages and stages[name]["ms_per_step"] > 0:
                    stages[name]["traffic_bytes"] = int(nbytes)
                    stages[name]["traffic_GBps"] = round(nbytes / (stages[name]["ms_per_step"] * 1e-3) / 1e9, 1)
        dom = max(stages, key=lambda n: stages[n]["avg_ms"])
        total_bytes = sum(bytes_by_stage.values())
        ms_per_step = dt / args.steps * 1e3
        value = views_per_step * K * args.steps / dt
        result = {
            "metric": "subframe-renders/sec (fwd+bwd), 1M Gaussians, K=15, 1080p, 1/2/4/8 GPU",
            "value": round(value, 2),
            "unit": "subframe-renders/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": ("mesh" if mesh_dims else ("strong" if subframes_mode else "weak")),
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.config}: P={P} Gaussians, {W}x{H}, K={K} subframes fused, curve_order={C}, "
                                   f"SH degree {args.sh_degree} (M={s}); " +
                                   (f"{mesh_dims[0]} blurry views per step, each view's K subframes split over {mesh_dims[1]} GPUs"
                                    if mesh_dims else
                                    "one blurry view per step, its K subframes split over the GPUs" if subframes_mode
                                    else "one blurry view per GPU per step"),
                       "step": ("query + fused loss + backward" + (" + grad all-reduce" if world > 1 else "") +
                                ("" if args.no_optimizer else " + densification stats + fused Adam (all groups; learning "
                                 "rates x1e-6 so the synthetic workload stays stationary)")),
                       "step_path": ("deblurgs_amd.training.TrainingLoop.step via " +
                                     ("fused_step.FusedStep (C ABI, no autograd, duplicate arrays sized ahead" +
                                      (", the iteration replayed as one captured hipGraph)" if graph_info else ")")
                                      if loop._fused is not None else "CameraMotionModule.query + torch autograd")),
                       "graph": graph_info,
                       "graph_policy": ("off (--no-graph)" if args.no_graph else "always (--graph-always)" if args.graph_always
                                        else "auto (TrainingLoop's default): a view whose compositing backward the library "
                                             "runs in parts when it is enqueued eagerly -- tile culling, K >= 6, >= 4 M "
                                             "duplicates -- is enqueued eagerly, smaller views are replayed as one hipGraph"),
                       "eager_preferred_steps": (loop._fused.eager_preferred if loop._fused is not None else None),
                       "dropped_steps": (loop._fused.dropped if loop._fused is not None else 0),
                       "retried_steps": loop.retried,
                       "tile_cull": bool(dgr.TILE_CULL),
                       # what the timed step renders: the K colour images; the depth images only if something reads them
                       # (the reference always composites them: see value_with_depth)
                       "depth_output": bool(loop._fused is not None and loop._fused.always_depth) or args.autograd_path,
                       # csrc/api.hip: how many parts the timed region's compositing backward ran in (1 = one launch; > 1: a
                       # large view enqueued eagerly, each part's row totals on a side stream beside the next part's
                       # compositing, bit-identical -- DESIGN.md 7); the per-stage averages below are taken with the stage
                       # timers on, i.e. with ONE launch per step
                       # and only for eagerly enqueued steps: the library does not fork inside a stream capture (a forked
                       # executable graph does not give its memory back on this runtime, tools/graph_fork_leak.hip)
                       "backward_in_parts": (int(_lib.lib().dgs_backward_parts(_lib.context(), int(round(K * frac_k)), int(R_tot * frac_k),
                                                                               int(bool(dgr.TILE_CULL)))) if
                                             (not graph_info or _lib.context_overlap_mode() == 3) else 1),
                       "profiled_region_ms_per_step": stats.get("profiled_ms_per_step"),
                       "sharding": (args.shard if world > 1 else "none"), "ranks_in_process_group": world,
                       "ar_chunks": (args.ar_chunks if world > 1 and loop._fused is not None else None),
                       "allreduce": (sharding.ALLREDUCE_MODE if world > 1 else None),
                       "allreduce_ms_per_step": None if allreduce_ms is None else round(allreduce_ms, 3),
                       "Pv_total": Pv_tot, "R_total": int(R_tot),
                       "pixel_gaussian_evals_upper_bound_per_step": int(256 * R_tot)},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": stages[dom]["model_GBps"], "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(stages[dom]["model_GBps"] / HBM_PEAK_GBS, 5), "traffic": None,
                         "alg_bytes_per_launch": stages[dom]["alg_bytes"], "avg_launch_ms": stages[dom]["avg_ms"],
                         "avg_launch_ms_source": "HIP events on the launch stream inside this run (dgs_profile_*)" +
                                                 (", recorded in the eager region that follows the replayed one" if graph_info else "") +
                                                 "; with the stage timers on the compositing backward runs as ONE launch "
                                                 "per step (as it does inside a captured graph; an eagerly enqueued step "
                                                 "runs it in parts, see config.backward_in_parts), and "
                                                 "profiles/r06_kernel_stats.csv is taken the same way (DGS_BWD_OVERLAP=0)",
                         "note": "the dominant kernel (compositing) is bound by VALU issue, not by HBM (SURVEY 8d): `frac` is "
                                 "the honest HBM fraction of its byte model, `valu` (when the round's PMC profile of this "
                                 "config is committed) the fraction of the VALU issue peak it reaches"},
            "pipeline_hbm": {"alg_bytes_per_step": int(total_bytes),
                             "model_GBps": round(total_bytes / (ms_per_step * 1e-3) / 1e9, 1),
                             "frac": round(total_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                             # counter-based: every kernel of the step (loss image and Adam included), committed passes
                             "traffic_bytes_per_step": (int(traffic_doc["_per_step_total"]) if traffic_doc is not None
                                                        else None),
                             "traffic_GBps": (round(traffic_doc["_per_step_total"] / (ms_per_step * 1e-3) / 1e9, 1)
                                              if traffic_doc is not None else None),
                             "traffic_source": (f"profiles/{traffic_file}: {traffic_doc.get('_source', '')}"
                                                if traffic_doc is not None else None)},
            "regions": {"ms_per_step": [round(x / args.steps * 1e3, 3) for x in region_dts],
                        "min_ms_per_step": round(min(region_dts) / args.steps * 1e3, 3),
                        "median_ms_per_step": round(ms_per_step, 3),
                        "max_ms_per_step": round(max(region_dts) / args.steps * 1e3, 3),
                        "value_min": round(views_per_step * K * args.steps / max(region_dts), 2),
                        "value_max": round(views_per_step * K * args.steps / min(region_dts), 2),
                        "note": "the timed region (exactly `steps` steps between barriers) run three times back to back; "
                                "value / ms_per_step are the median region's"},
            "build_id": _lib.build_id(),
            "stages": stages,
        }
        if fwd_only is not None:
            result["forward_only_renders_per_s"] = fwd_only["value"]
            result["forward_only"] = fwd_only
        if with_depth is not None:
            result["value_with_depth"] = with_depth
        if ref_lists is not None:
            result["value_reference_lists"] = ref_lists
        if ref_lists_depth is not None:
            result["value_reference_lists_with_depth"] = ref_lists_depth
        if per_rank is not None:
            result["config"]["per_rank"] = per_rank
        if world > 1:
            result["config"]["rccl"] = parse_rccl_log(rccl_log)
            result["config"]["backend"] = dist.get_backend()
        if emu is not None:
            # not the metric line: one rank's share of a G-GPU step, measured alone on one GPU with degenerate collectives
            k0e, k1e = sharding.shard_range(K, emu[0], emu[1]) if subframes_mode else (0, K)
            result = {"emulated_shard": {"rank": emu[0], "world": emu[1], "sharding": args.shard,
                                         "subframes_of_this_rank": k1e - k0e, "K": K,
                                         "ms_per_step": round(ms_per_step, 3), "steps": args.steps,
                                         "graph": graph_info,
                                         "eager_ms_per_step": None if graph_info is None else graph_info.get("eager_ms_per_step"),
                                         "rccl": parse_rccl_log(rccl_log),
                                         "bucket_bytes": 4 * sum(p.numel() for p in params),
                                         "blur_bytes": 12 * H * W,
                                         "note": "one rank's launches between its exchanges (one-rank process group on the "
                                                 "real backend: every collective is degenerate); NOT a throughput figure"},
                      "config": result["config"], "stages": stages}
        # PMC counters cannot be collected inside this run (rocprofv3 wraps the process): traffic / VALU figures are the
        # committed measurements of the same command, with their provenance, or null
        import glob
        for stem, key in (("traffic", "traffic"), ("valu", "valu")):
            found = sorted(glob.glob(os.path.join(ROOT, "profiles", f"{stem}_r[0-9][0-9].json")))    # the newest round's
            if not found or world != 1 or emu is not None:
                continue
            path = found[-1]
            fname = os.path.basename(path)
            try:
                doc = json.load(open(path))
                if doc.get("_config") != args.config:
                    continue
                if key == "traffic" and dom in doc:
                    result["roofline"]["traffic"] = doc[dom]
                    result["roofline"]["traffic_source"] = f"profiles/{fname}: {doc.get('_source', '')}"
                if key == "valu":
                    hit = [v_ for k_, v_ in doc.items() if k_.startswith(dom + "_kernel")]
                    if hit:
                        h = hit[0]
                        # the bound that actually limits the dominant kernel: VALU issue.  frac = measured wave64 VALU
                        # instructions per cycle and SIMD / 0.5 (one fp32 instruction per 2 cycles on a SIMD-32);
                        # issue_weighted_frac prices every instruction class at its own PMC-measured issue cost (<= 1)
                        result["roofline"]["valu"] = {
                            "bound": "valu", "achieved": h.get("ipc_per_simd"), "peak": 0.5,
                            "unit": "wave64 VALU instructions / cycle / SIMD", "frac": h.get("frac_of_peak_issue"),
                            "issue_weighted_frac": h.get("issue_weighted_frac"), "lds_busy_frac": h.get("lds_busy_frac"),
                            "valu_insts_per_launch": h.get("valu_insts"), "cycles_per_launch": h.get("cycles_per_xcd"),
                            "passes_per_entry": h.get("passes_per_entry"),
                            "source": f"profiles/{fname}: {doc.get('_source', '')}"}
            except Exception:
                pass
        if world == 1 and emu is None and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline_port(scene, K // 2)
            try:
                result["cpu_baseline_torch_naive"] = cpu_baseline_torch_naive(scene, K // 2)
            except Exception as ex:     # never lose the headline line to the secondary baseline
                result["cpu_baseline_torch_naive"] = {"error": repr(ex)}
    else:
        result = None
    hung = False
    if world > 1 and not args.no_extras and not args.autograd_path:
        import threading
        extras = {}

        def guarded():
            torch.cuda.set_device(local_rank)
            run_extras(extras)
        th = threading.Thread(target=guarded, daemon=True)
        th.start()
        th.join(timeout=float(os.environ.get("DGS_BENCH_EXTRAS_TIMEOUT_S", "300")))
        hung = th.is_alive()
        if hung:
            extras = dict(extras, error="an extra region did not return within DGS_BENCH_EXTRAS_TIMEOUT_S; the headline "
                                        "figures above were measured before it started")
        if result is not None:
            # both sharding modes as first-class values keyed by mode: whichever the one multi-GPU run
            # favours can be read as the headline; `value` above is the --shard mode's
            result["by_mode"] = {args.shard: {"sharding": args.shard, "scaling": result["scaling"], "steps": args.steps,
                                              "value": result["value"], "ms_per_step": result["ms_per_step"],
                                              "views_per_step": views_per_step, "regions": result.get("regions")}}
            for name_, res_ in (extras.pop("other_modes", None) or {}).items():
                result["by_mode"][name_] = res_
            result["extras"] = extras
            result["config"]["rccl"] = parse_rccl_log(rccl_log)      # (again: now with the extra regions' collectives)
    final_line = json.dumps(result) if result is not None else None
    if hung:    # the process group is wedged: no barrier, no destroy -- say what was measured and leave
        sys.stderr.flush()
        if final_line is not None:
            print(final_line, flush=True)
        os._exit(EXIT_EXTRAS_HUNG)   # non-zero: a hung collective is not a success; the headline line is out
    if dist.is_initialized():
        if world > 1:
            dist.barrier()
        dist.destroy_process_group()
    # The one JSON line goes out LAST: the collective library prints a version banner of its own through C stdio (block-
    # buffered when stdout is a pipe, flushed at exit), which would otherwise land behind the line in rank 0's output.
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stderr.flush()
    if final_line is not None:
        print(final_line, flush=True)


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args))
    selftest = os.environ.get("DGS_BENCH_SELFTEST")      # launcher test hook (tests/test_host_logic.py): no GPU work
    if selftest:
        kind, _, who = selftest.partition(":")
        if kind == "die" and os.environ.get("RANK") == who:
            sys.exit(5)                # (any code but 0 and EXIT_EXTRAS_HUNG)
        time.sleep(3600)
    run_rank(args)


if __name__ == "__main__":
    main()
