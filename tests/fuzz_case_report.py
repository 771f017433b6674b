"""Not a test: prints where the HIP path and the oracle disagree on one case of test_fuzz_shapes_against_oracle's seeded
sweep (DGS_FUZZ_SWEEP).  usage: python tests/fuzz_case_report.py P W H K seed sigma deg [use_sigmoid] [sh_degree=N]"""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 1)[0])
import test_gpu_parity as T  # noqa: E402
from helpers import exempt_pixels, hip_forward_backward, oracle_forward_backward, synthetic  # noqa: E402

P, W, H, K, seed = (int(x) for x in sys.argv[1:6])
sigma, deg = float(sys.argv[6]), int(sys.argv[7])
kw = {}
for a in sys.argv[8:]:
    if a == "use_sigmoid":
        kw["use_sigmoid"] = True
    elif a.startswith("sh_degree="):
        kw["sh_degree"] = int(a.split("=")[1])
sc = synthetic.make_scene(P, W, H, K=K, seed=seed, sigma_px=sigma, sh_degree=deg)
rng = np.random.default_rng(seed)
sc["opacities"][:20] = 1.0
sc["opacities"][20:40] = 0.0
sc["opacities"][40:60] = 1.0 / 255.0
sc["means3D"][60:80, 2] = rng.uniform(0.15, 0.25, 20)
sc["scales"][80:90] = 1e-9
sc["scales"][90:100, 0] *= 12.0
sc["means3D"][100] = [50.0, -40.0, 2.0]
sc["scales"][100] = 5.0
gC, gD = T._grads(sc, K, seed=seed)
hip = hip_forward_backward(sc, K, gC, gD, **kw)
ora = oracle_forward_backward(sc, K, gC, gD, **kw)
ex = exempt_pixels(sc, K, ora["states"], **kw)
print("exempt pixels per subframe:", [int(e.sum()) for e in ex], "of", W * H)
for k in range(K):
    d = np.abs(hip["color"][k] - ora["color"][k]).max(axis=0)
    print(f"subframe {k}: image max diff off the exempt set {d[~ex[k]].max():.3e}, on it {d[ex[k]].max() if ex[k].any() else 0:.3e}")
for key in T.GRAD_KEYS:
    a, b = np.asarray(hip[key], np.float64), np.asarray(ora[key], np.float64)
    a = a.reshape(b.shape)
    e = np.abs(a - b)
    i = np.unravel_index(e.argmax(), e.shape)
    print(f"{key:16s} rel {e.max() / (np.abs(b).max() + 1e-30):.3e}  worst at {i}: hip {a[i]:.6e} oracle {b[i]:.6e}  max|oracle| {np.abs(b).max():.3e}")
    if key in ("dL_drotations", "dL_dscales", "dL_dmeans3D"):
        g = i[0]
        print(f"    Gaussian {g}: scales {sc['scales'][g]}, opacity {sc['opacities'][g]}, mean {sc['means3D'][g]}, "
              f"radii per subframe {[int(hip['radii'][k][g]) for k in range(K)]}")
        print(f"    hip row {a[g]}\n    ora row {b[g]}")
