"""Where the time of a full-size oracle check goes on the GPU box's host (128 cores)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import OracleRun, oracle_forward, synthetic   # noqa: E402
from oracle import oracle   # noqa: E402

sc = synthetic.make_config("metric")
K, H, W = 3, sc["H"], sc["W"]
t = time.time()
oracle.use_openmp(True)
print("threads", oracle.lib().dgs_oracle_threads(), "cpu_count", os.cpu_count(), flush=True)
st = [oracle_forward(sc, k) for k in range(K)]
print(f"forward x{K}: {time.time() - t:.2f} s", flush=True)
t = time.time()
un = [oracle.unstable(s) for s in st]
print(f"unstable x{K}: {time.time() - t:.2f} s", flush=True)
g = np.random.default_rng(0).normal(size=(K, 3, H, W)).astype(np.float32)
for mode in ("double", "f32", "fma"):
    oracle.set_accum_f32(mode == "f32")
    oracle.use_fma(mode == "fma")
    t = time.time()
    for k in range(K):
        oracle.backward(st[k], g[k])
    print(f"backward[{mode}] x{K}: {time.time() - t:.2f} s", flush=True)
oracle.use_fma(False)
oracle.set_accum_f32(False)
# thread scaling of one backward
import ctypes
omp = ctypes.CDLL("libgomp.so.1")
for n in (128, 64, 32, 16):
    omp.omp_set_num_threads(n)
    t = time.time()
    oracle.backward(st[0], g[0])
    tb = time.time() - t
    t = time.time()
    oracle_forward(sc, 0)
    print(f"threads {n}: backward {tb:.2f} s, forward {time.time() - t:.2f} s", flush=True)
