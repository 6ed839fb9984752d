"""Helpers shared by the parity tests: run the same frames through the CPU oracle and the
HIP path and compare the resulting Layer<TsdfVoxel> block by block."""
import numpy as np


def compare_tsdf(gpu, ref, exact=True, tol=1e-4):
    """gpu/ref: {(bx,by,bz): (dist, weight, rgba, updated_bits)}.  Returns a stats dict and
    raises AssertionError on mismatch."""
    gk, rk = set(gpu.keys()), set(ref.keys())
    assert gk == rk, (f"allocated block sets differ: only-gpu={sorted(gk - rk)[:5]} "
                      f"only-ref={sorted(rk - gk)[:5]} ({len(gk)} vs {len(rk)})")
    max_dd = max_dw = 0.0
    ncol = nvox = 0
    for k in rk:
        gd, gw, gc, gu = gpu[k]
        rd, rw, rc, ru = ref[k]
        assert gu == ru, f"updated bits differ at block {k}: {gu} vs {ru}"
        # identical sets of observed voxels ("voxel indices bit-exact")
        assert np.array_equal(gw > 0, rw > 0), f"observed-voxel mask differs in block {k}"
        if exact:
            assert np.array_equal(gd.view(np.uint32), rd.view(np.uint32)), \
                f"distance bits differ in block {k}: max |d|={np.abs(gd - rd).max()}"
            assert np.array_equal(gw.view(np.uint32), rw.view(np.uint32)), \
                f"weight bits differ in block {k}: max |d|={np.abs(gw - rw).max()}"
            assert np.array_equal(gc, rc), f"colours differ in block {k}"
        else:
            max_dd = max(max_dd, float(np.abs(gd - rd).max()))
            max_dw = max(max_dw, float(np.abs(gw - rw).max()))
            ncol += int((gc != rc).any(axis=1).sum())
        nvox += int((rw > 0).sum())
    if not exact:
        assert max_dd <= tol and max_dw <= tol * max(1.0, float(max(np.abs(v[1]).max() for v in ref.values()))), \
            f"max |dd|={max_dd} max |dw|={max_dw}"
    return dict(blocks=len(rk), observed_voxels=nvox, max_dd=max_dd, max_dw=max_dw, color_mismatch=ncol)


def layer_stats(a, b, obs_eps=1e-6):
    """evaluateLayersRmse-style statistics (utils/evaluation_utils.h:73-185) between two
    TSDF dicts: rmse over voxels observed in both, and overlap counts."""
    se = 0.0
    n_both = n_a_only = n_b_only = 0
    max_err = 0.0
    for k in set(a.keys()) | set(b.keys()):
        da, wa = (a[k][0], a[k][1]) if k in a else (None, None)
        db, wb = (b[k][0], b[k][1]) if k in b else (None, None)
        oa = wa > obs_eps if wa is not None else None
        ob = wb > obs_eps if wb is not None else None
        if oa is None:
            n_b_only += int(ob.sum())
            continue
        if ob is None:
            n_a_only += int(oa.sum())
            continue
        both = oa & ob
        n_both += int(both.sum())
        n_a_only += int((oa & ~ob).sum())
        n_b_only += int((ob & ~oa).sum())
        if both.any():
            e = (da[both] - db[both]).astype(np.float64)
            se += float((e * e).sum())
            max_err = max(max_err, float(np.abs(e).max()))
    rmse = (se / n_both) ** 0.5 if n_both else 0.0
    return dict(rmse=rmse, max_err=max_err, both=n_both, a_only=n_a_only, b_only=n_b_only)


def clear_sphere_assertions(tsdf, esdf, min_distance_m):
    """The assertions of test_clear_spheres.cc:175-203 over {block: (d, w, rgba, upd)} /
    {block: (d, flags, parent, upd)} dicts.  Returns (#band voxels, #hallucinated voxels)."""
    import numpy as np
    n_band = n_hall = 0
    for b, (td, tw, _, _) in tsdf.items():
        assert b in esdf, f"ESDF lacks TSDF block {b}"                    # ASSERT_TRUE(esdf_layer.hasBlock)
        ed, ef, _, _ = esdf[b]
        obs = (ef & 1).astype(bool)
        hall = (ef & 2).astype(bool)
        unobserved_tsdf = tw < 1e-6
        assert np.all(hall[unobserved_tsdf & obs]), "observed ESDF voxel without TSDF data must be hallucinated"
        band = (tw > 1e-6) & (np.abs(td) <= np.float32(min_distance_m))
        assert np.all(obs[band]) and not np.any(hall[band])
        assert np.array_equal(np.sign(td[band]), np.sign(ed[band]))
        assert np.abs(td[band] - ed[band]).max(initial=0.0) <= 1e-3
        n_band += int(band.sum()); n_hall += int(hall.sum())
    return n_band, n_hall
