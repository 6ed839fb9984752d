"""Host-side mirror of the reference's integrator classes over the C-ABI.

Same names, argument meaning and error behaviour as
/root/reference/voxblox/include/voxblox/integrator/tsdf_integrator.h (TsdfIntegratorBase,
SimpleTsdfIntegrator, MergedTsdfIntegrator, FastTsdfIntegrator, TsdfIntegratorFactory) and
esdf_integrator.h (EsdfIntegrator), mesh/mesh_integrator.h, mesh_layer.h, mesh.h (MeshIntegrator,
MeshLayer, Mesh), so parity tests read like the reference's own tests
(test/test_sdf_integrators.cc).  All numerics happen in libvbx_hip.so on the GPU.
"""
import numpy as np

from . import capi


class Transformation:
    """kindr::minimal::QuatTransformationTemplate<float>: translation + unit quaternion (w,x,y,z)."""

    def __init__(self, position=(0, 0, 0), quat_wxyz=(1, 0, 0, 0)):
        self.position = tuple(float(v) for v in position)
        self.quat_wxyz = tuple(float(v) for v in quat_wxyz)

    def getPosition(self):
        return self.position


class TsdfIntegratorBase:
    """tsdf_integrator.h:51-198.  `layer` is a voxblox_amd.capi.Map (the HBM-resident Layer)."""
    Config = staticmethod(capi.tsdf_cfg)
    kind = None

    def __init__(self, config, layer):
        if layer is None:
            raise ValueError("layer must not be null")  # CHECK_NOTNULL, tsdf_integrator.cc:69
        self.config_ = config
        self.setLayer(layer)

    def setLayer(self, layer):
        if layer is None:
            raise ValueError("layer must not be null")
        self.layer_ = layer

    def getConfig(self):
        return self.config_

    def integratePointCloud(self, T_G_C, points_C, colors, freespace_points=False):
        """tsdf_integrator.h:100-103."""
        if len(points_C) != len(colors):
            raise ValueError("points_C.size() != colors.size()")  # CHECK_EQ, tsdf_integrator.cc:247
        self.layer_.integrate(self.kind, self.config_, T_G_C.position, T_G_C.quat_wxyz, points_C,
                              colors, freespace_points)


class SimpleTsdfIntegrator(TsdfIntegratorBase):
    kind = capi.TSDF_SIMPLE


class MergedTsdfIntegrator(TsdfIntegratorBase):
    kind = capi.TSDF_MERGED


class FastTsdfIntegrator(TsdfIntegratorBase):
    kind = capi.TSDF_FAST


class TsdfIntegratorFactory:
    """tsdf_integrator.h:201-209, tsdf_integrator.cc:8-46."""
    _names = {"simple": SimpleTsdfIntegrator, "merged": MergedTsdfIntegrator,
              "fast": FastTsdfIntegrator}
    _types = {1: SimpleTsdfIntegrator, 2: MergedTsdfIntegrator, 3: FastTsdfIntegrator}

    @staticmethod
    def create(integrator_type, config, layer):
        if layer is None:
            raise ValueError("layer must not be null")
        if isinstance(integrator_type, str):
            if not integrator_type:
                raise ValueError("empty integrator type name")
            cls = TsdfIntegratorFactory._names.get(integrator_type)
            if cls is None:
                raise ValueError(f"Unknown TSDF integrator type: {integrator_type}")
        else:
            cls = TsdfIntegratorFactory._types.get(int(integrator_type))
            if cls is None:
                raise ValueError(f"Unknown TSDF integrator type: {integrator_type}")
        return cls(config, layer)


class EsdfIntegrator:
    """esdf_integrator.h:24-179.  tsdf_layer and esdf_layer are the same capi.Map handle."""
    Config = staticmethod(capi.esdf_cfg)

    def __init__(self, config, tsdf_layer, esdf_layer=None):
        if tsdf_layer is None:
            raise ValueError("tsdf_layer must not be null")
        self.config_ = config
        self.map_ = tsdf_layer

    def updateFromTsdfLayer(self, clear_updated_flag):
        self.map_.esdf_update(self.config_, batch=False, clear_updated_flag=clear_updated_flag)

    def updateFromTsdfLayerBatch(self):
        self.map_.esdf_update(self.config_, batch=True, clear_updated_flag=False)

    def updateFromTsdfBlocks(self, tsdf_blocks, incremental=False):
        self.map_.esdf_update_blocks(self.config_, tsdf_blocks, incremental)

    def clear(self):
        self.map_.esdf_integrator_clear()

    def addNewRobotPosition(self, position):
        self.map_.esdf_add_new_robot_position(self.config_, position)


class Mesh:
    """mesh/mesh.h:35-162: the arrays MeshIntegrator fills (numpy instead of AlignedVector)."""

    def __init__(self, block_size, origin):
        self.block_size = block_size
        self.origin = np.asarray(origin, np.float32)
        self.vertices = np.zeros((0, 3), np.float32)
        self.normals = np.zeros((0, 3), np.float32)
        self.colors = np.zeros((0, 4), np.uint8)
        self.indices = np.zeros(0, np.uint64)
        self.updated = False

    def clear(self):
        self.vertices = np.zeros((0, 3), np.float32)
        self.normals = np.zeros((0, 3), np.float32)
        self.colors = np.zeros((0, 4), np.uint8)
        self.indices = np.zeros(0, np.uint64)

    def size(self):
        return self.vertices.shape[0]


class MeshLayer:
    """mesh/mesh_layer.h:23-312, the members MeshIntegrator and the publishers use."""

    def __init__(self, block_size):
        self.block_size_ = np.float32(block_size)
        self.mesh_map_ = {}

    def allocateMeshPtrByIndex(self, index):
        key = tuple(int(x) for x in index)
        m = self.mesh_map_.get(key)
        if m is None:
            m = Mesh(self.block_size_, np.asarray(key, np.float32) * self.block_size_)   # mesh_layer.h:112-122
            self.mesh_map_[key] = m
        return m

    def getMeshPtrByIndex(self, index):
        return self.mesh_map_.get(tuple(int(x) for x in index))

    def getAllAllocatedMeshes(self):
        return list(self.mesh_map_.keys())

    def getAllUpdatedMeshes(self):
        return [k for k, m in self.mesh_map_.items() if m.updated]

    def getNumberOfAllocatedMeshes(self):
        return len(self.mesh_map_)

    def clear(self):
        self.mesh_map_.clear()


class MeshIntegrator:
    """mesh/mesh_integrator.h:72-412 for TsdfVoxel: generateMesh() runs vbx_mesh_generate and stores
    every re-meshed block in the (host) MeshLayer, like updateMeshForBlock (:250-270)."""
    Config = staticmethod(capi.mesh_cfg)

    def __init__(self, config, sdf_layer, mesh_layer):
        if sdf_layer is None or mesh_layer is None:
            raise ValueError("sdf_layer and mesh_layer must not be null")   # CHECK_NOTNULL, :96-97
        self.config_ = config
        self.map_ = sdf_layer
        self.mesh_layer_ = mesh_layer

    def generateMesh(self, only_mesh_updated_blocks, clear_updated_flag):
        idx, off, v, n, c = self.map_.mesh_generate(self.config_, only_mesh_updated_blocks, clear_updated_flag)
        for b, key in enumerate(idx):
            a, e = int(off[b]), int(off[b + 1])
            mesh = self.mesh_layer_.allocateMeshPtrByIndex(key)
            mesh.clear()
            mesh.vertices = v[a:e].copy()
            mesh.normals = n[a:e].copy()
            if c is not None:
                mesh.colors = c[a:e].copy()
            mesh.indices = np.arange(e - a, dtype=np.uint64)   # marching_cubes.h:94-96
            mesh.updated = True
