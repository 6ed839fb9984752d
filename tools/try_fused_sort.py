"""Self-test of the fused radix pass (vbx_sort.hpp, the default) against std::stable_sort."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from voxblox_amd import capi
gm = capi.Map(0.05, 16, max_blocks=64)
cases = [(1, 0, 20), (100, 0, 20), (8192, 0, 20), (8193, 32, 52), (300000, 32, 52), (1000003, 44, 64), (1000003, 0, 8),
         (3000001, 32, 57), (70000, 3, 33), (500000, 0, 4)]
for n, b, e in cases:
    for seed in (1, 2):
        t = time.time()
        gm.selftest_sort(n, b, e, seed, with_vals=True)
        gm.selftest_sort(n, b, e, seed + 10, with_vals=False)
        print("ok", n, b, e, seed, round(time.time() - t, 2), flush=True)
print("fused sort self-test passed")
