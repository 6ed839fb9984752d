"""Per-kernel table (vbx_profile_*) of the Fast integrator over frames [SKIP, SKIP+N) of the room stream.
usage: python tools/kernel_table.py VOXEL N [SKIP] [fast|merged|simple]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from voxblox_amd import capi, scenes
if "VBX_LIB" in os.environ:
    capi.LIB_PATH = os.environ["VBX_LIB"]   # A/B builds
voxel = float(sys.argv[1]); nf = int(sys.argv[2]); skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
kind = {"fast": capi.TSDF_FAST, "merged": capi.TSDF_MERGED, "simple": capi.TSDF_SIMPLE}[sys.argv[4] if len(sys.argv) > 4 else "fast"]
gm = capi.Map(voxel, 16, max_blocks=int(8192 * max(1.0, (0.05 / voxel) ** 3)))
cfg = capi.tsdf_cfg(default_truncation_distance=4 * voxel)
gm.enable_timing(True)
tot = {}
for k in range(skip + nf):
    pose, pts, col = scenes.room_frame(k, 100)
    if k == skip:
        gm.profile(True, reset=True)
    gm.integrate(kind, cfg, pose[0], pose[1], pts, col)
    if k >= skip:
        for a, b in gm.timing().items():
            tot[a] = tot.get(a, 0.0) + b
        c = gm.counters()
        print(f"frame {k}: total {gm.timing()['total_ms']:.3f} ms replay {gm.timing()['replay_ms']:.3f} rounds {c['replay_rounds']} "
              f"(in blocks {c['replay_block_rounds']}) updates {c['voxel_updates']}")
tab, calls = gm.profile_table()
print({a: round(b / nf, 4) for a, b in tot.items()})
print(f"{'kernel':44s} {'launches/frame':>14s} {'us/launch':>10s} {'us/frame':>10s}")
s = 0.0
for name, (n, ms) in sorted(tab.items(), key=lambda kv: -kv[1][1]):
    print(f"{name:44s} {n / calls:14.1f} {1e3 * ms / n:10.2f} {1e3 * ms / calls:10.1f}")
    s += ms
print(f"{'sum':44s} {sum(n for n, _ in tab.values()) / calls:14.1f} {'':10s} {1e3 * s / calls:10.1f}")
