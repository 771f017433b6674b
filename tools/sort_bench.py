"""Micro-benchmark of dgs_sort_pairs (A/B of library builds: select one with DGS_LIB_PATH; the rejected sort modes live in variants/binning_sort_modes_and_ranges_sweep.patch): python tools/sort_bench.py [n] [bits]"""
import ctypes, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deblurgs_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 63_000_000
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 49
L = _lib.lib()
g = torch.Generator(device="cuda").manual_seed(0)
keys = torch.randint(0, 2 ** 62, (n,), dtype=torch.int64, device="cuda", generator=g) & ((1 << bits) - 1)
vals = torch.arange(n, dtype=torch.int32, device="cuda")
k0, v0, k1, v1 = keys.clone(), vals.clone(), torch.empty_like(keys), torch.empty_like(vals)
tmp = torch.empty(L.dgs_sort_tmp_bytes(n) + 16, dtype=torch.uint8, device="cuda")
alt = ctypes.c_int32(0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def run():
    k0.copy_(keys); v0.copy_(vals)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(L.dgs_sort_pairs(k0.data_ptr(), v0.data_ptr(), k1.data_ptr(), v1.data_ptr(), n, 0, bits, tmp.data_ptr(), ctypes.byref(alt), st), "sort")
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
ts = [run() for _ in range(6)]
ko = k1 if alt.value else k0
ok = bool((ko[1:] >= ko[:-1]).all().item())
print(f"lib={os.environ.get('DGS_LIB_PATH','default')} n={n} bits={bits} ms={min(ts[1:]):.3f} sorted={ok}")
