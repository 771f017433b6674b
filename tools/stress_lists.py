"""One-off stress run: random small problems (sizes that are multiples of nothing, K from 1 to 31, tiny to huge splats):
reference duplicate lists against the oracle bit for bit, tile-culled results bit-equal to them, capacity mode equal to
the two-phase path, both record storages (DgsProblem.wide_records) equal.  Prints one line per case; exits non-zero on the
first mismatch.    python tools/stress_lists.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import synthetic, hip_forward_state, hip_forward_backward, oracle_forward, tile_cull, wide_records

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
GRADS = ("dL_dmeans3D", "dL_dsh", "dL_dopacities", "dL_dscales", "dL_drotations", "dL_dmeans2D", "dL_dviewmatrix",
         "dL_dprojmatrix")
for case in range(n_cases):
    P = int(rng.choice([1, 2, 7, 63, 64, 65, 255, 257, 1000, 4097, 20000]))
    W, H = int(rng.integers(17, 400)), int(rng.integers(17, 300))
    K = int(rng.choice([1, 2, 3, 5, 8, 15, 16, 17, 31]))
    sigma = float(rng.choice([0.3, 1.0, 2.5, 8.0, 30.0]))
    sc = synthetic.make_scene(P, W, H, K=K, seed=int(rng.integers(1 << 30)), sigma_px=sigma)
    a = hip_forward_state(sc, K, cull=False)
    # the reference's lists, subframe by subframe
    off = 0
    for k in range(K):
        o = oracle_forward(sc, k)
        R = o["num_rendered"]
        T = a["T"]
        assert np.array_equal(a["point_list"][off:off + R], o["point_list"]), (case, k, "point_list")
        assert np.array_equal(a["keys"][off:off + R] - (np.uint64(k * T) << np.uint64(32)), o["keys"]), (case, k, "keys")
        off += R
    assert off == a["R"]
    b = hip_forward_state(sc, K, cull=True)
    for key in ("color", "depth", "final_T", "radii"):   # (n_contrib counts list positions: differs by construction)
        assert np.array_equal(a[key], b[key]), (case, key)
    for cap in (b["R"] + 5, max(b["R"], 1)):
        c = hip_forward_state(sc, K, cull=True, capacity=cap)
        assert c["R"] == b["R"] and not c["overflow"]
        assert np.array_equal(b["keys"], c["keys"][:b["R"]]) and np.array_equal(b["point_list"], c["point_list"][:b["R"]])
        assert np.array_equal(b["ranges"], c["ranges"]) and np.array_equal(b["color"], c["color"])
    gC = rng.normal(size=(K, 3, H, W)).astype(np.float32)
    with wide_records(True):
        d = hip_forward_state(sc, K, cull=True)
        with tile_cull(True):
            g0 = hip_forward_backward(sc, K, gC)
    with tile_cull(True):
        g1 = hip_forward_backward(sc, K, gC)
    with tile_cull(False):
        g2 = hip_forward_backward(sc, K, gC)
    assert not d["compact_keys"]
    for key in ("keys", "point_list", "ranges", "color"):
        assert np.array_equal(b[key], d[key]), (case, key, "storage")
    for key in GRADS:
        assert np.array_equal(g0[key], g1[key]) and np.array_equal(g1[key], g2[key]), (case, key)
    print(f"case {case:3d}  P={P:6d} {W}x{H} K={K:2d} sigma={sigma:5.1f}  R={a['R']:8d} culled={b['R']:8d} "
          f"compact={b['compact_keys']}", flush=True)
print("stress ok")
