"""What a finer compositing granularity could buy: for the metric scene, compares the number of (Gaussian, 8x8
quadrant) passes the kernels run today with the number of lock-step rounds if each 16-lane row of the wave walked
its own list of Gaussians hitting its 4x4 (or 8x2 ...) sub-block.  Measurement tool only."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from helpers import synthetic, hip_forward_state

cfg = sys.argv[1] if len(sys.argv) > 1 else "metric"
sc = synthetic.make_config(cfg, K=1)
st = hip_forward_state(sc, 1, cull=True)
W, H, T = sc["W"], sc["H"], st["T"]
gx = (W + 15) // 16
keys = st["keys"]; tile = (keys >> np.uint64(32)).astype(np.int64)
g = st["point_list"].astype(np.int64)
rows = st["rows"][0, g].astype(np.float64)
x, y, a, b, c, op = (rows[:, i] for i in range(6))
r2 = 2 * np.log(255 * op)
tx, ty = tile % gx, tile // gx

def hit_box(x0, y0, w, h):
    """exact ellipse-vs-box of pixel centres [x0, x0+w-1] x [y0, y0+h-1] (absolute pixel coords per dup)"""
    x_lo, x_hi = x - (x0 + w - 1), x - x0
    y_lo, y_hi = y - (y0 + h - 1), y - y0
    def emx(e, lo, hi):
        t = np.clip(-b * e / c, lo, hi); return a * e * e + (2 * b * e + c * t) * t
    def emy(e, lo, hi):
        t = np.clip(-b * e / a, lo, hi); return c * e * e + (2 * b * e + a * t) * t
    qm = np.minimum(np.minimum(emx(x_lo, y_lo, y_hi), emx(x_hi, y_lo, y_hi)), np.minimum(emy(y_lo, x_lo, x_hi), emy(y_hi, x_lo, x_hi)))
    inside = (x_lo <= 0) & (x_hi >= 0) & (y_lo <= 0) & (y_hi >= 0)
    return ~(np.where(inside, 0.0, qm) > r2)

nt = int(tile.max()) + 1
quad = [hit_box(tx * 16 + (q & 1) * 8, ty * 16 + (q >> 1) * 8, 8, 8) for q in range(4)]
passes_now = sum(int(h.sum()) for h in quad)
print("dups", len(tile), "quadrant passes", passes_now, "per dup", passes_now / len(tile))
for name, (sw, sh) in {"4x4": (4, 4), "8x2": (8, 2), "2x8": (2, 8), "8x4 (32 lanes)": (8, 4), "4x8 (32 lanes)": (4, 8)}.items():
    rounds = 0
    for q in range(4):
        qx, qy = tx * 16 + (q & 1) * 8, ty * 16 + (q >> 1) * 8
        cnts = []
        for sy in range(0, 8, sh):
            for sx in range(0, 8, sw):
                h = hit_box(qx + sx, qy + sy, sw, sh) & quad[q]
                cnts.append(np.bincount(tile[h], minlength=nt))
        rounds += int(np.max(np.stack(cnts), axis=0).sum())
    print(name, "rounds", rounds, "ratio to passes", rounds / passes_now)
