"""Localises the one-pixel discrepancy behind dL_dconic[k=8, g=683110] at the metric size (HIP -2488.466 vs oracle
-2487.980): subframe 8 alone, upstream gradient as in tests/test_gpu_configs.py, bisected over pixel windows."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch   # noqa: E402
from helpers import OracleRun, hip_cloud_forward_backward, hip_state_on_device, synthetic   # noqa: E402
from test_gpu_configs import kernel_activated_scene   # noqa: E402
from oracle import oracle   # noqa: E402

KSEL, G = int(os.environ.get("DBG_K", "8")), int(os.environ.get("DBG_G", "683110"))
sc = synthetic.make_config("metric")
K, H, W = sc["K"], sc["H"], sc["W"]
act = kernel_activated_scene(sc)
sub_raw, sub_act = dict(sc), dict(act)
for name in ("viewmatrix", "projmatrix", "campos"):
    sub_raw[name] = sc[name][KSEL:KSEL + 1]
    sub_act[name] = act[name][KSEL:KSEL + 1]
sub_raw["K"] = sub_act["K"] = 1
rng = np.random.default_rng(3)
g_all = rng.normal(size=(K, 3, H, W)).astype(np.float32)[KSEL:KSEL + 1]
run = OracleRun(sub_act, 1, margin_masks=False)
st0 = hip_state_on_device(sub_raw, 1, cull=False, raw=True, checksum=True)
print("exempt pixels", run.use_exact_masks(st0["contrib_checksum"].cpu().numpy(), st0["n_contrib"].cpu().numpy()))
hip_nc = st0["n_contrib"].cpu().numpy().view(np.uint32).reshape(H, W)
hip_ft = st0["final_T"].cpu().numpy().reshape(H, W)
del st0
g_all, _ = run.mask(g_all)
st = run.states[0]
mx, my = st["means2D"][G]
rad = int(st["radii"][G])
print("Gaussian", G, "mean2D", mx, my, "radius", rad, "conic_opacity", st["conic_opacity"][G].tolist())


def diff(gC):
    hip = hip_cloud_forward_backward(sub_raw, 1, gC, cull=True)
    oracle.use_openmp(True)
    try:
        ob = oracle.backward(st, gC[0])
    finally:
        oracle.use_openmp(False)
    a = hip["dL_dconic"][0][G]
    b = ob["dL_dconic"][G][[0, 1, 3]].astype(np.float64)
    return a, b


a, b = diff(g_all)
print("full frame: hip", a.tolist(), "oracle", b.tolist(), "diff", (a - b).tolist())
x0, x1 = max(int(mx) - rad - 16, 0), min(int(mx) + rad + 17, W)
y0, y1 = max(int(my) - rad - 16, 0), min(int(my) + rad + 17, H)
while (x1 - x0) * (y1 - y0) > 1:
    if x1 - x0 >= y1 - y0:
        xm = (x0 + x1) // 2
        boxes = [(x0, xm, y0, y1), (xm, x1, y0, y1)]
    else:
        ym = (y0 + y1) // 2
        boxes = [(x0, x1, y0, ym), (x0, x1, ym, y1)]
    best = None
    for bx in boxes:
        gC = np.zeros_like(g_all)
        gC[0][:, bx[2]:bx[3], bx[0]:bx[1]] = g_all[0][:, bx[2]:bx[3], bx[0]:bx[1]]
        a, b = diff(gC)
        d = float(np.abs(a - b).max())
        print("window x", bx[0], bx[1], "y", bx[2], bx[3], "max |diff|", d, flush=True)
        if best is None or d > best[0]:
            best = (d, bx)
    x0, x1, y0, y1 = best[1]
px, py = x0, y0
print("pixel", px, py, "hip n_contrib", int(hip_nc[py, px]), "oracle n_contrib", int(st["n_contrib"][py * W + px]),
      "hip final_T", float(hip_ft[py, px]), "oracle final_T", float(st["final_T"][py * W + px]),
      "upstream", g_all[0][:, py, px].tolist())
# the oracle's traversal of that pixel (float32, the reference's expressions)
gx = (W + 15) // 16
tile = (py // 16) * gx + (px // 16)
r0, r1 = (int(v) for v in st["ranges"][tile])
T = np.float32(1.0)
f32 = np.float32
for s in range(r0, r1):
    g = int(st["point_list"][s])
    dx, dy = f32(st["means2D"][g][0]) - f32(px), f32(st["means2D"][g][1]) - f32(py)
    co = st["conic_opacity"][g]
    power = f32(-0.5) * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy
    if power > 0:
        continue
    alpha = min(f32(0.99), co[3] * np.exp(power, dtype=np.float32))
    if g == G or abs(float(alpha) - 1 / 255) < 2e-6:
        print(f"   pos {s - r0 + 1} g {g} power {float(power):.7f} alpha {float(alpha):.9f} (1/255 = {1 / 255:.9f}) T before {float(T):.6g}"
              f"{'  <-- the Gaussian' if g == G else ''}")
    if alpha < f32(1.0 / 255.0):
        continue
    tT = T * (f32(1) - alpha)
    if tT < f32(0.0001):
        print(f"   stop at pos {s - r0 + 1} (T {float(T):.6g} -> {float(tT):.6g})")
        break
    T = tT
