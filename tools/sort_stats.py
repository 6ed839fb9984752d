"""Where a tile of k_rsort_fused spends its time (library built with -DVBX_SORT_STATS, passed in VBX_LIB): the stable
sort self-test on n keys, 20-bit field = two passes; 100 MHz ticks per stage, mean and max over the tiles."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from voxblox_amd import capi
capi.LIB_PATH = os.environ["VBX_LIB"]
L = capi.lib()
L.vbx_debug_sort_stats.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
gm = capi.Map(0.1, 16, max_blocks=64)
out = (C.c_ulonglong * 16)()
names = ["load + count", "publish", "wait", "row sums", "scatter"]
for n in [int(a) for a in sys.argv[1:]] or [300_000, 1_000_000, 2_000_000, 4_000_000, 16_000_000]:
    gm.selftest_sort(n, 44, 64, 2, with_vals=False)   # warm
    L.vbx_debug_sort_stats(out, 1)
    gm.selftest_sort(n, 44, 64, 2, with_vals=False)
    L.vbx_debug_sort_stats(out, 1)
    o = [int(x) for x in out]
    t = max(o[7], 1)
    print("n = %9d: %5d tiles (2 passes);" % (n, t), "  ".join("%s %.1f / %.1f us" % (names[i], o[i] / t / 100, o[8 + i] / 100) for i in range(5)))
