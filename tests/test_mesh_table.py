"""The packed marching-cubes case table (voxblox_amd/csrc/vbx_mc_table.hpp for the kernels,
oracle/vbx_mc_table.hpp for the oracle): both copies identical, structurally sane, and — where
the reference tree exists — equal to MarchingCubes::kTriangleTable (marching_cubes.cc:32-288)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/voxblox/src/mesh/marching_cubes.cc"


def _words(path):
    src = open(path).read()
    body = src[src.index("kMcTriTable[256]"):]
    return [int(x, 16) for x in re.findall(r"0x([0-9a-f]{16})ull", body)[:256]]


def _unpack(w):
    n = w >> 60
    return [(w >> (4 * k)) & 15 for k in range(3 * n)]


def test_both_copies_identical_and_sane():
    a = _words(os.path.join(ROOT, "voxblox_amd", "csrc", "vbx_mc_table.hpp"))
    b = _words(os.path.join(ROOT, "oracle", "vbx_mc_table.hpp"))
    assert len(a) == 256 and a == b
    edge_corners = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]
    for case, w in enumerate(a):
        edges = _unpack(w)
        assert (w >> 60) <= 5 and all(e < 12 for e in edges)
        assert (w & ((1 << 60) - 1)) >> (12 * (w >> 60)) == 0           # nothing after the list
        # every referenced edge crosses the surface: exactly one end inside
        for e in edges:
            c0, c1 = edge_corners[e]
            assert ((case >> c0) & 1) != ((case >> c1) & 1), (case, e)
        assert (len(edges) == 0) == (case in (0, 255))
    # complementary cases use the same edges
    for case in range(256):
        assert set(_unpack(a[case])) == set(_unpack(a[255 - case]))


@pytest.mark.skipif(not os.path.exists(REF), reason="no reference tree")
def test_equals_reference_table():
    src = open(REF).read()
    body = src[src.index("kTriangleTable[256][16]"):]
    body = body[body.index("{") + 1:]
    rows = []
    for m in re.finditer(r"\{([^{}]*)\}", body):
        vals = [int(x) for x in m.group(1).split(",") if x.strip()]
        if len(vals) != 16:
            break
        rows.append(vals)
    assert len(rows) >= 256
    words = _words(os.path.join(ROOT, "voxblox_amd", "csrc", "vbx_mc_table.hpp"))
    for case in range(256):
        assert _unpack(words[case]) == [e for e in rows[case] if e >= 0], case
