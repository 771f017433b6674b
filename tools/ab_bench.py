"""A/B of library builds on the GPU box: runs bench.py once per library (DGS_LIB_PATH) and prints the step time and the
per-stage averages side by side.  usage: python tools/ab_bench.py [--steps N] [--config metric] lib_a.so lib_b.so ...
('default' = the in-tree build)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    steps, config, extra = "30", "metric", []
    libs = []
    i = 0
    while i < len(args):
        if args[i] == "--steps":
            steps = args[i + 1]; i += 2
        elif args[i] == "--config":
            config = args[i + 1]; i += 2
        elif args[i] == "--extra":
            extra = args[i + 1].split(); i += 2
        else:
            libs.append(args[i]); i += 1
    rows = {}
    for rep in range(2):            # two rounds, interleaved, to see the run-to-run noise
        for lib in libs:
            env = dict(os.environ)
            if lib != "default":
                env["DGS_LIB_PATH"] = os.path.abspath(lib)
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", steps, "--warmup", "5",
                                "--config", config, "--no-cpu-baseline", "--no-reference-lists"] + extra, env=env,
                               capture_output=True, text=True)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if not line:
                print(lib, "FAILED", r.stderr[-800:])
                continue
            d = json.loads(line[-1])
            rows.setdefault(lib, []).append(d)
    names = ["preprocess", "depth_order", "tile_cull", "duplicate", "sort", "ranges", "composite_fwd", "composite_bwd",
             "contrib_reduce", "geometry_bwd", "scan"]
    print(f"{'lib':28s} {'ms/step':>8s} " + " ".join(f"{n[:9]:>9s}" for n in names))
    for lib, ds in rows.items():
        for d in ds:
            st = d["stages"]
            print(f"{os.path.basename(lib):28s} {d['ms_per_step']:8.3f} " +
                  " ".join(f"{st[n]['avg_ms']:9.4f}" if n in st else f"{'-':>9s}" for n in names), flush=True)


if __name__ == "__main__":
    main()
