import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import *
from test_gpu_configs import kernel_activated_scene
np.set_printoptions(linewidth=200, precision=5, suppress=False)
sc = synthetic.make_config("cfg2"); K = sc["K"]
act = kernel_activated_scene(sc)
run = OracleRun(act, K)
rng = np.random.default_rng(3)
gC = rng.normal(size=(K, 3, sc["H"], sc["W"])).astype(np.float32)
gC, _ = run.mask(gC)
hip = hip_cloud_forward_backward(sc, K, gC, cull=True)
hip2 = hip_forward_backward(act, K, gC)
ora = run.backward(gC)
for k in (0, 4, 8):
    b = ora["double"]["dL_dviewmatrix"][k]; n = ora["f32"]["dL_dviewmatrix"][k]
    print("k", k, "oracle\n", b, "\nhip-ora\n", hip["dL_dviewmatrix"][k] - b, "\nhip(activated path)-ora\n", hip2["dL_dviewmatrix"][k] - b, "\nnoise\n", n - b)
    print("proj hip-ora\n", hip["dL_dprojmatrix"][k] - ora["double"]["dL_dprojmatrix"][k])
