"""Runs the two-rank point-to-point all-reduce leg of the N-rank functional check N times on this box and counts the
runs whose replicas stayed bit-identical.

    python tools/p2p_repeat.py --repeat 20 --mode subframes            # the product path: must be 20 / 20
    python tools/p2p_repeat.py --repeat 20 --mode subframes --direct   # round 3's racy layer: expected to fail on some boxes

Two ranks share cuda:0 over gloo (DGS_DIST_ONE_DEVICE=1), DGS_DIST_ALLREDUCE=p2p, DGS_DIST_P2P_MIN_NUMEL=0 so that every
slice of every chunk takes the reduce-scatter from iteration 1 on, --ar-chunks 4 (the reduction runs on the side stream
behind the backward's chunks: the configuration GPUTEST_r03 failed in)."""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--repeat", type=int, default=20)
    ap.add_argument("--mode", default="subframes", choices=["views", "subframes"])
    ap.add_argument("--direct", action="store_true")
    ap.add_argument("--iters", type=int, default=33)
    a = ap.parse_args()
    env = dict(os.environ, DGS_DIST_BACKEND="gloo", DGS_DIST_ONE_DEVICE="1", DGS_DIST_ALLREDUCE="p2p",
               DGS_DIST_P2P_MIN_NUMEL="0", PYTHONPATH=os.path.abspath(ROOT))
    cmd = [sys.executable, os.path.join(ROOT, "tools", "dist_training_check.py"), "--ranks", "2", "--mode", a.mode,
           "--ar-chunks", "4", "--iters", str(a.iters)] + (["--p2p-direct"] if a.direct else [])
    ok = 0
    t0 = time.time()
    for i in range(a.repeat):
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        good = r.returncode == 0 and "identical: True" in r.stdout
        ok += good
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("mode ")]
        print(f"run {i + 1:2d}: {'ok  ' if good else 'FAIL'} {line[-1] if line else r.stderr.strip().splitlines()[-1:]}", flush=True)
    print(f"p2p repeat: layer {'round-3 direct (racy)' if a.direct else 'product (_p2p)'} mode {a.mode}: {ok} / {a.repeat} "
          f"runs with bit-identical replicas, {time.time() - t0:.0f} s", flush=True)
    sys.exit(0 if (ok == a.repeat or a.direct) else 1)


if __name__ == "__main__":
    main()
