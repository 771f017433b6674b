"""Oracle throughput on the GPU box's host: K metric subframes, W host threads x T OpenMP threads.
usage: oracle_par.py W T   (environment: OMP_WAIT_POLICY etc. as given)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
W, T = int(sys.argv[1]), int(sys.argv[2])
os.environ["DGS_ORACLE_THREADS"] = str(T)
from helpers import oracle_forward, synthetic   # noqa: E402
from oracle import oracle   # noqa: E402

oracle.parallel_calls = lambda: W
sc = synthetic.make_config("metric")
K = 8
oracle.use_openmp(True)
t = time.time()
st = oracle.map_subframes(lambda k: oracle_forward(sc, k), range(K))
t1 = time.time()
un = oracle.map_subframes(oracle.unstable, st)
t2 = time.time()
g = np.random.default_rng(0).normal(size=(K, 3, sc["H"], sc["W"])).astype(np.float32)
gr = oracle.map_subframes(lambda k: oracle.backward(st[k], g[k]), range(K))
t3 = time.time()
print(f"W={W} T={T} wait={os.environ.get('OMP_WAIT_POLICY', '-')} proc_bind={os.environ.get('OMP_PROC_BIND', '-')}: "
      f"forward {t1 - t:.2f}  unstable {t2 - t1:.2f}  backward {t3 - t2:.2f}   (x{K} subframes)", flush=True)
