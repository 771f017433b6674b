"""Prints the HIP-vs-oracle error statistics that the parity bars in tests/ are set from.  For each gradient:
whole-tensor relerr (round-1 metric), per-column and per-Gaussian errors (helpers.grad_errors) against the OpenMP oracle
(double accumulation = the centre of the reference's order-dependent fp32 results), and beside them the same metrics
for the oracle's own fp32 rounding noise (emulated-fp32 accumulation vs double).  The upstream gradient is zero on the
pixels whose oracle traversal sits on one of the reference's thresholds (oracle.unstable), as in the tests.
Usage: python tools/parity_report.py [small|metric2]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import (grad_errors, hip_forward_backward, oracle_forward, relerr, synthetic)   # noqa: E402
from oracle import oracle   # noqa: E402

KEYS = ["dL_dmeans3D", "dL_dopacities", "dL_dsh", "dL_dscales", "dL_drotations", "dL_dmeans2D", "dL_dviewmatrix",
        "dL_dprojmatrix"]
OKEY = {"dL_dopacities": "dL_dopacity"}


def report(sc, K, tag, depth=True):
    rng = np.random.default_rng(5)
    gC = rng.normal(size=(K, 3, sc["H"], sc["W"])).astype(np.float32)
    gD = (rng.normal(size=(K, 1, sc["H"], sc["W"])) * 0.05).astype(np.float32) if depth else None
    oracle.use_openmp(True)
    try:
        sts = [oracle_forward(sc, k) for k in range(K)]
        frac = []
        for k, st in enumerate(sts):
            un = oracle.unstable(st)
            frac.append(un.mean())
            gC[k][:, un] = 0.0
            if gD is not None:
                gD[k][:, un] = 0.0
        print(f"{tag}: unstable pixel fraction per subframe: max {max(frac):.2e}")
        res = {}
        for mode in ("double", "f32"):
            oracle.set_accum_f32(mode == "f32")
            gs = [oracle.backward(st, gC[k], None if gD is None else gD[k]) for k, st in enumerate(sts)]
            res[mode] = {}
            for key in KEYS:
                ok = OKEY.get(key, key)
                if key in ("dL_dmeans2D", "dL_dviewmatrix", "dL_dprojmatrix"):
                    res[mode][key] = np.stack([g[ok] for g in gs])
                else:
                    res[mode][key] = sum(g[ok].astype(np.float64) for g in gs)
        oracle.set_accum_f32(False)
    finally:
        oracle.use_openmp(False)
    hip = hip_forward_backward(sc, K, gC, gD)
    for key in KEYS:
        b, n = res["double"][key], res["f32"][key]
        a = hip[key].reshape(b.shape)
        if key in ("dL_dviewmatrix", "dL_dprojmatrix"):
            print(f"{tag} {key:16s} relerr {relerr(a, b):.2e}   (oracle fp32 noise {relerr(n, b):.2e})")
            continue
        rows = a.shape[0] * a.shape[1] if key == "dL_dmeans2D" else a.shape[0]
        if key == "dL_dmeans2D":
            a, b, n = a[..., :2], b[..., :2], n[..., :2]
        e, en = grad_errors(a, b, rows=rows), grad_errors(n, b, rows=rows)
        print(f"{tag} {key:16s} relerr {relerr(a, b):.2e}  col {e['col']:.2e} (noise {en['col']:.2e})  "
              f"row {e['row']:.2e} (noise {en['row']:.2e})")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "small"
    if what == "small":
        report(synthetic.make_scene(2500, 160, 120, K=3, seed=2), 3, "small")
        report(synthetic.make_scene(3000, 200, 136, K=3, seed=1, sigma_px=5.0), 3, "big-splats")
        report(synthetic.make_config("cfg2"), 9, "cfg2")
    else:
        report(synthetic.make_config("metric", K=2), 2, "metric-K2")
