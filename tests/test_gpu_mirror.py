"""-m gpu: host <-> HBM coherence entry points: the bulk block mirror
must return exactly what the per-block call returns, for both layers, and report missing blocks."""
import numpy as np
import pytest

from voxblox_amd import scenes

pytestmark = pytest.mark.gpu


def test_blocks_download_equals_per_block():
    from voxblox_amd import capi
    gm = capi.Map(0.05, 16, max_blocks=2048)
    cfg = capi.tsdf_cfg(default_truncation_distance=0.2)
    for k in (0, 5):
        pose, pts, col = scenes.room_frame(k, 100, f=80.0, width=160, height=120)
        gm.integrate(capi.TSDF_MERGED, cfg, pose[0], pose[1], pts, col)
    ecfg = capi.esdf_cfg(min_distance_m=0.1)
    gm.esdf_update(ecfg, batch=False, clear_updated_flag=False)
    gm.esdf_add_new_robot_position(capi.esdf_cfg(min_distance_m=0.1, clear_sphere_radius=0.5, occupied_sphere_radius=1.0),
                                   np.zeros(3, np.float32))
    for layer in (capi.LAYER_TSDF, capi.LAYER_ESDF):
        idx = gm.block_indices(layer)
        assert len(idx) > 50
        v, u, hd = gm.blocks_download(idx, layer)
        assert v.shape == (len(idx), 4096)
        for k in range(0, len(idx), 7):
            v1, u1, hd1 = gm.block_download(idx[k], layer)
            assert v[k].tobytes() == v1.tobytes()
            assert (u[k], hd[k]) == (u1, hd1)
    # page-locked staging buffer
    st = gm.pinned_voxels(len(idx), capi.LAYER_ESDF)
    v2, u2, _ = gm.blocks_download(idx, capi.LAYER_ESDF, out=st)
    assert v2.tobytes() == v.tobytes() and np.array_equal(u2, u)
    # updated-block listing + bulk download is the per-frame mirror of INTEGRATION.md
    upd = gm.blocks_updated(capi.UPDATE_MAP)
    v, u, _ = gm.blocks_download(upd)
    assert len(upd) == gm.num_blocks() and (u & 1).all()
    gm.clear_updated(capi.UPDATE_MAP)
    assert len(gm.blocks_updated(capi.UPDATE_MAP)) == 0
    assert len(gm.blocks_updated(capi.UPDATE_MESH)) == gm.num_blocks()   # other bits untouched
    # a block that is not in the layer is an error, like the per-block call
    with pytest.raises(Exception):
        gm.blocks_download(np.array([[1000, 1000, 1000]], np.int32))
    assert gm.blocks_download(np.zeros((0, 3), np.int32))[0].shape[0] == 0
