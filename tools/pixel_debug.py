"""Diagnoses the worst 'stable' pixel of one subframe of a config: prints the oracle's traversal around its thresholds."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import *
from test_gpu_configs import kernel_activated_scene
from oracle import oracle
cfg, k = sys.argv[1], int(sys.argv[2])
sc = synthetic.make_config(cfg); K = sc["K"]
act = kernel_activated_scene(sc)
st = hip_state_on_device(sc, K, cull=False, raw=True)
oracle.use_openmp(True)
o = oracle_forward(act, k)
un = oracle.unstable(o)
oracle.use_openmp(False)
col = st["color"][k].cpu().numpy()
dc = np.abs(col - o["color"]).max(axis=0)
dc_s = np.where(un, 0, dc)
bad = np.argwhere(dc_s > 1e-4)
print("pixels beyond 1e-4 among stable:", len(bad), "unstable fraction", un.mean())
W, H = sc["W"], sc["H"]
gx = (W + 15) // 16
for (py, px) in bad[:4]:
    pid = py * W + px
    print("pixel", px, py, "err", dc[py, px], "hip n_contrib", int(st["n_contrib"][k][pid]), "oracle", int(o["n_contrib"][pid]),
          "hip final_T", float(st["final_T"][k][pid]), "oracle", float(o["final_T"][pid]))
    tile = (py // 16) * gx + px // 16
    r0, r1 = o["ranges"][tile]
    T = np.float32(1.0)
    for s in range(r0, r1):
        g = o["point_list"][s]
        dx = np.float32(o["means2D"][g, 0] - np.float32(px)); dy = np.float32(o["means2D"][g, 1] - np.float32(py))
        c = o["conic_opacity"][g]
        power = np.float32(np.float32(-0.5) * (c[0] * dx * dx + c[2] * dy * dy) - c[1] * dx * dy)
        if power > 0:
            if abs(power) < 1e-3: print("   pos", s - r0, "power>0", power)
            continue
        alpha = min(np.float32(0.99), np.float32(c[3] * np.exp(power)))
        if alpha < 1 / 255:
            if abs(alpha - 1 / 255) < 2e-5: print("   pos", s - r0, "alpha just below", alpha - 1 / 255)
            continue
        if abs(alpha - 1 / 255) < 2e-5: print("   pos", s - r0, "alpha just above", alpha - 1 / 255, "T", T)
        if alpha >= 0.99: print("   pos", s - r0, "alpha clamped 0.99, raw", c[3] * np.exp(power), "T", T)
        tT = np.float32(T * (1 - alpha))
        if tT < 1e-4:
            print("   pos", s - r0, "terminates: test_T", tT, "T", T, "alpha", alpha)
            break
        if abs(tT - 1e-4) < 2e-6: print("   pos", s - r0, "test_T near 1e-4:", tT)
        T = tT
