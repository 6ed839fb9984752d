"""Which kernels wait for a load right after issuing it?  Compiles the device code to assembly (no GPU needed)
and counts, per kernel, the global loads that are followed within three instructions by `s_waitcnt vmcnt(0)` —
a load inside an `if` together with its first use compiles to load / wait / use per item instead of N loads in
flight (found in the radix kernels: eight dependent round trips per thread).
usage: python tools/isa_load_waits.py [min_count]"""
import os, re, subprocess, sys, tempfile
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
out = os.path.join(tempfile.mkdtemp(), "vbx.s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                       "-fhip-fp32-correctly-rounded-divide-sqrt", "--cuda-device-only", "-S", "-o", out,
                       os.path.join(root, "voxblox_amd", "csrc", "vbx_hip.hip")], stderr=subprocess.DEVNULL)
funcs, cur = {}, None
for line in open(out):
    m = re.match(r'^(_ZN\S+):', line)
    if m and 'GLOBAL__N_1' in m.group(1):
        cur = m.group(1); funcs[cur] = []
    elif line.startswith('.Lfunc_end'):
        cur = None
    elif cur:
        funcs[cur].append(line.strip())
is_load = lambda l: re.match(r'(global_load|flat_load|buffer_load)', l) is not None
rows = {}
for name, body in funcs.items():
    m = re.search(r'\d+(k_[a-z_0-9]+)', name)
    short = m.group(1) if m else name[:40]
    loads = [i for i, l in enumerate(body) if is_load(l)]
    waited = 0
    for i in loads:
        for j in range(i + 1, min(i + 4, len(body))):
            if body[j].startswith('s_waitcnt') and 'vmcnt(0)' in body[j]:
                waited += 1
                break
            if is_load(body[j]):
                break
    if short not in rows or rows[short][1] < waited:
        rows[short] = (len(loads), waited, len(body))
lo = int(sys.argv[1]) if len(sys.argv) > 1 else 3
print(f"{'kernel':34s} {'loads':>6s} {'waited at once':>15s} {'instructions':>13s}")
for k, (n, w, sz) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    if w >= lo:
        print(f"{k:34s} {n:6d} {w:15d} {sz:13d}")
