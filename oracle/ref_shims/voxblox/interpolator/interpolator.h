// DEPENDENCY SHIM (oracle/_ref build only): mesh_integrator.h:39 includes the interpolator but
// never uses it; the real header needs Eigen features (colwise, 8x8 products) outside the
// stand-in in ref_shims/Eigen/Core, so the include resolves to this empty file instead.
#pragma once
