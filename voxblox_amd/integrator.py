"""Host-side mirror of the reference's integrator classes over the C-ABI.

Same names, argument meaning and error behaviour as
/root/reference/voxblox/include/voxblox/integrator/tsdf_integrator.h (TsdfIntegratorBase,
SimpleTsdfIntegrator, MergedTsdfIntegrator, FastTsdfIntegrator, TsdfIntegratorFactory) and
esdf_integrator.h (EsdfIntegrator), so parity tests read like the reference's own tests
(test/test_sdf_integrators.cc).  All numerics happen in libvbx_hip.so on the GPU.
"""
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
