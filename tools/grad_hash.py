"""SHA-256 of every output of one forward + backward through the benchmarked operator path (raw cloud parameters, tile
culling) at a named configuration: two builds / environment settings agree bit for bit iff the printed lines are equal.
usage: python tools/grad_hash.py [cfg2|metric|...]"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def hashes(name="cfg2"):
    """One line per output: name, SHA-256 prefix, shape."""
    from deblurgs_amd import synthetic
    import helpers
    sc = synthetic.make_config(name, seed=0)
    K = sc["K"]
    rng = np.random.default_rng(5)
    dL = rng.standard_normal((K, 3, sc["H"], sc["W"])).astype(np.float32)
    out = helpers.hip_cloud_forward_backward(sc, K, dL, keep_on_device=True)
    lines = []
    for key in sorted(out):
        a = out[key].cpu().numpy()
        lines.append(f"{key:16s} {hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:24]}  {a.shape}")
    return lines


def main():
    for ln in hashes(sys.argv[1] if len(sys.argv) > 1 else "cfg2"):
        print(ln)


if __name__ == "__main__":
    main()
