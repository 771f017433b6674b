import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from deblurgs_amd import synthetic
sc = synthetic.make_config("metric")
for th in (8, 16, 32, 128):
    os.environ["DGS_NAIVE_THREADS"] = str(th)
    t0 = time.time(); r = bench.cpu_baseline_torch_naive(sc, 7); print(th, r["seconds"], r["cores"], flush=True)
