"""Measures how many (tile, Gaussian) duplicates of the reference's 3-sigma rectangle rule can contribute at all
(exact ellipse-vs-tile test at the alpha >= 1/255 level set), on the metric scene.  Measurement tool only."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from helpers import synthetic, hip_forward_state

cfg = dict(synthetic.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "metric"])
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2
sc = synthetic.make_scene(cfg["P"], cfg["W"], cfg["H"], K=K, seed=0, sh_degree=cfg.get("sh_degree", 2))
st = hip_forward_state(sc, K)
R, T = st["R"], st["T"]
W, H = sc["W"], sc["H"]
gx = (W + 15) // 16
keys = st["keys"]
kt = (keys >> np.uint64(32)).astype(np.int64)
k, tile = kt // T, kt % T
tx, ty = tile % gx, tile // gx
g = st["point_list"].astype(np.int64)
rows = st["rows"][k, g]            # [R, 12]
x, y, a, b, c, op = (rows[:, i].astype(np.float64) for i in range(6))
r2 = 2 * np.log(255 * op)
# d = mean - pixel over pixel centres [t*16, t*16+15] clipped to the image
x_lo, x_hi = x - np.minimum(tx * 16 + 15, W - 1), x - tx * 16
y_lo, y_hi = y - np.minimum(ty * 16 + 15, H - 1), y - ty * 16
def emin_x(e, lo, hi):
    t = np.clip(-b * e / c, lo, hi)
    return a * e * e + (2 * b * e + c * t) * t
def emin_y(e, lo, hi):
    t = np.clip(-b * e / a, lo, hi)
    return c * e * e + (2 * b * e + a * t) * t
qm = np.minimum(np.minimum(emin_x(x_lo, y_lo, y_hi), emin_x(x_hi, y_lo, y_hi)),
                np.minimum(emin_y(y_lo, x_lo, x_hi), emin_y(y_hi, x_lo, x_hi)))
inside = (x_lo <= 0) & (x_hi >= 0) & (y_lo <= 0) & (y_hi >= 0)
qm = np.where(inside, 0.0, qm)
hit = ~(qm > r2)
print("R", R, "hit fraction", hit.mean())
tt = st["tiles_touched"].reshape(-1)
vis = tt > 0
print("visible", vis.sum(), "mean tiles/visible", tt[vis].mean(), "tight", hit.sum() / vis.sum())
cnt = np.bincount(k * sc["P"] + g, weights=hit, minlength=K * sc["P"])
print("gaussians with zero tight tiles among visible:", ((cnt == 0) & vis).mean() / vis.mean())
for q in (1, 2, 4, 6, 9, 16):
    print("rect tiles <=", q, (tt[vis] <= q).mean(), " tight <=", q, (cnt[vis] <= q).mean())
