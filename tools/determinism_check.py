"""Bitwise run-to-run reproducibility of the fused forward + backward at the metric configuration."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from helpers import synthetic, hip_forward_backward, tile_cull
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
sc = synthetic.make_config("metric", K=K)
rng = np.random.default_rng(3)
gC = rng.normal(size=(K, 3, sc["H"], sc["W"])).astype(np.float32)
ref = None
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6):
    with tile_cull(True):
        out = hip_forward_backward(sc, K, gC, None)
    if ref is None:
        ref = out
        continue
    bad = []
    for k, v in out.items():
        if isinstance(v, np.ndarray) and not np.array_equal(v, ref[k], equal_nan=True):
            d = np.abs(v.astype(np.float64) - ref[k].astype(np.float64))
            bad.append((k, int((d > 0).sum()), float(d.max()), float(np.abs(ref[k]).max())))
    print("run", it, "differences:", bad)
