"""voxblox_amd — MI355X-native TSDF/ESDF integration hot path of voxblox.

Only what the hot path needs lives here:
  csrc/      hand-written HIP kernels for gfx950 + the C-ABI (include/vbx_hip.h)
  capi.py    ctypes binding of the C-ABI shared library (fails loudly if it is missing)
  integrator.py  host-side mirror of the reference's integrator classes
  scenes.py  deterministic synthetic depth-camera inputs for the BASELINE configs
"""
__all__ = ["capi", "integrator", "scenes"]
