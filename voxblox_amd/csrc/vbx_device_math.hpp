// Device-side arithmetic of the TSDF hot path for gfx950.
//
// Every function here must produce bit-identical results to the reference's CPU code
// (compiled without FMA contraction), so: this translation unit is built with
// -ffp-contract=off and correctly-rounded fp32 divide/sqrt, min/max are spelled as the
// ternaries std::min/std::max expand to (NaN behaviour!), 3-element reductions associate
// as c0 + (c1 + c2) like Eigen's unrolled redux, and the voxel centre goes through double
// exactly where the reference's `+ 0.5` literal forces it.  Reference file:line is cited
// per function (paths relative to /root/reference/voxblox).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vbx {

struct f3 {
  float x, y, z;
};
struct l3 {
  long long x, y, z;
};
struct i3 {
  int x, y, z;
};

__host__ __device__ inline f3 f3_sub(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__host__ __device__ inline f3 f3_add(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__host__ __device__ inline f3 f3_mul(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__host__ __device__ inline f3 f3_div(f3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
__host__ __device__ inline float f3_sqnorm(f3 a) { return a.x * a.x + (a.y * a.y + a.z * a.z); }
__host__ __device__ inline float f3_dot(f3 a, f3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
__host__ __device__ inline f3 f3_cross(f3 a, f3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__host__ __device__ inline float f3_norm(f3 a) { return sqrtf(f3_sqnorm(a)); }
// Eigen 3.3 normalized(): guarded by squaredNorm > 0.
__host__ __device__ inline f3 f3_normalized(f3 a) {
  const float z = f3_sqnorm(a);
  if (z > 0.0f) return f3_div(a, sqrtf(z));
  return a;
}
// std::min / std::max semantics (second argument wins only on strict comparison).
__host__ __device__ inline float std_min(float a, float b) { return (b < a) ? b : a; }
__host__ __device__ inline float std_max(float a, float b) { return (a < b) ? b : a; }

// common.h:248
__host__ __device__ inline int signum(float x) { return (x == 0) ? 0 : x < 0 ? -1 : 1; }

// minkindr QuatTransformation::transform = q.rotate(v) + t with Eigen's
// Quaternion::_transformVector (uv = q.vec x v; uv += uv; v + w*uv + q.vec x uv).
struct Pose {
  f3 t;
  float qw, qx, qy, qz;
};
__host__ __device__ inline f3 pose_transform(const Pose& T, f3 v) {
  const f3 qv{T.qx, T.qy, T.qz};
  f3 uv = f3_cross(qv, v);
  uv = f3_add(uv, uv);
  const f3 rot = f3_add(f3_add(v, f3_mul(uv, T.qw)), f3_cross(qv, uv));
  return f3_add(rot, T.t);
}

// common.h:153-159 — floor(p * inv + 1e-6f) in fp32.
__host__ __device__ inline l3 grid_index_from_point(f3 p, float inv) {
  return {(long long)floorf(p.x * inv + 1e-6f), (long long)floorf(p.y * inv + 1e-6f),
          (long long)floorf(p.z * inv + 1e-6f)};
}
// common.h:166-171
__host__ __device__ inline l3 grid_index_from_scaled_point(f3 p) {
  return {(long long)floorf(p.x + 1e-6f), (long long)floorf(p.y + 1e-6f),
          (long long)floorf(p.z + 1e-6f)};
}
// common.h:187-193 — evaluated in double because of the 0.5 literal, rounded once.
__host__ __device__ inline f3 center_point_from_grid_index(l3 idx, float grid_size) {
  return {(float)(((double)(float)idx.x + 0.5) * (double)grid_size),
          (float)(((double)(float)idx.y + 0.5) * (double)grid_size),
          (float)(((double)(float)idx.z + 0.5) * (double)grid_size)};
}
// common.h:215-224
__host__ __device__ inline i3 block_index_from_global(l3 g, float vps_inv) {
  return {(int)floorf((float)g.x * vps_inv), (int)floorf((float)g.y * vps_inv),
          (int)floorf((float)g.z * vps_inv)};
}
// common.h:233-243
__host__ __device__ inline i3 local_from_global(l3 g, int vps) {
  const long long off = (long long)INT32_MIN;
  return {(int)((g.x + off) & (vps - 1)), (int)((g.y + off) & (vps - 1)),
          (int)((g.z + off) & (vps - 1))};
}
// block_hash.h:54-64 — wraps in 64 bit, truncates to 32.
__host__ __device__ inline uint32_t long_index_hash(l3 i) {
  const unsigned long long sl = 17191ull, sl2 = sl * sl;
  return (uint32_t)((unsigned long long)i.x + (unsigned long long)i.y * sl +
                    (unsigned long long)i.z * sl2);
}

// common.h:105-125 — Color::blendTwoColors on packed r | g<<8 | b<<16 | a<<24.
__host__ __device__ inline uint32_t blend_two_colors(uint32_t c1, float w1, uint32_t c2, float w2) {
  const float total = w1 + w2;
  w1 /= total;
  w2 /= total;
  uint32_t out = 0;
#pragma unroll
  for (int ch = 0; ch < 4; ++ch) {
    const float a = (float)(int)((c1 >> (8 * ch)) & 0xFF);
    const float b = (float)(int)((c2 >> (8 * ch)) & 0xFF);
    const float v = roundf(a * w1 + b * w2);
    out |= ((uint32_t)(int)v & 0xFFu) << (8 * ch);
  }
  return out;
}

// MixedThreadSafeIndex, integrator_utils.cc:54-63, and its inverse (point -> sequence).
__host__ __device__ inline size_t mixed_index(size_t seq, size_t n) {
  const size_t step = 1024, groups = n / step;
  if (groups * step <= seq) return seq;
  return (seq % groups) * step + seq / groups;
}
__host__ __device__ inline size_t mixed_index_inverse(size_t p, size_t n) {
  const size_t step = 1024, groups = n / step;
  if (groups * step <= p) return p;
  return (p % step) * groups + p / step;
}

// ---------------------------------------------------------------------------
// RayCaster, integrator_utils.cc:72-179.  A plain struct stepped by next().
// ---------------------------------------------------------------------------
struct RayCaster {
  float tx, ty, tz;  // t_to_next_boundary_
  float dx, dy, dz;  // t_step_size_
  long long cx, cy, cz;
  int sx, sy, sz;
  unsigned steps;  // ray_length_in_steps_
  unsigned cur;    // current_step_

  // integrator_utils.cc:72-104 (endpoints) + :127-179 (setup).
  __device__ inline void init(f3 origin, f3 point_G, bool clearing, bool carving,
                              float max_ray_length_m, float voxel_size_inv, float trunc,
                              bool cast_from_origin) {
    const f3 d = f3_sub(point_G, origin);
    const f3 unit = f3_normalized(d);
    f3 ray_start, ray_end;
    if (clearing) {
      float len = f3_norm(d);
      len = std_min(std_max(len - trunc, 0.0f), max_ray_length_m);
      ray_end = f3_add(origin, f3_mul(unit, len));
      ray_start = carving ? origin : ray_end;
    } else {
      ray_end = f3_add(point_G, f3_mul(unit, trunc));
      ray_start = carving ? origin : f3_sub(point_G, f3_mul(unit, trunc));
    }
    const f3 a = f3_mul(ray_start, voxel_size_inv);
    const f3 b = f3_mul(ray_end, voxel_size_inv);
    if (cast_from_origin) setup(a, b); else setup(b, a);
  }

  __device__ inline void setup(f3 s, f3 e) {
    if (isnan(s.x) || isnan(s.y) || isnan(s.z) || isnan(e.x) || isnan(e.y) || isnan(e.z)) {
      steps = 0;
      cur = 1;  // emit nothing (the reference's behaviour is undefined here)
      tx = ty = tz = dx = dy = dz = 0.f;
      cx = cy = cz = 0;
      sx = sy = sz = 0;
      return;
    }
    const l3 c = grid_index_from_scaled_point(s);
    const l3 en = grid_index_from_scaled_point(e);
    cx = c.x; cy = c.y; cz = c.z;
    cur = 0;
    const long long ddx = en.x - c.x, ddy = en.y - c.y, ddz = en.z - c.z;
    steps = (unsigned)((ddx < 0 ? -ddx : ddx) + (ddy < 0 ? -ddy : ddy) + (ddz < 0 ? -ddz : ddz));
    const f3 r = f3_sub(e, s);
    sx = signum(r.x); sy = signum(r.y); sz = signum(r.z);
    const float shx = s.x - (float)c.x, shy = s.y - (float)c.y, shz = s.z - (float)c.z;
    const float bx = (float)(sx > 0 ? sx : 0) - shx;
    const float by = (float)(sy > 0 ? sy : 0) - shy;
    const float bz = (float)(sz > 0 ? sz : 0) - shz;
    tx = bx / r.x; ty = by / r.y; tz = bz / r.z;  // no zero guard
    dx = (float)sx / r.x; dy = (float)sy / r.y; dz = (float)sz / r.z;
  }

  // integrator_utils.cc:111-125 — Eigen minCoeff: strict '<' against the running minimum.
  __device__ inline bool next(l3* out) {
    if (cur++ > steps) return false;
    out->x = cx; out->y = cy; out->z = cz;
    int a = 0;
    float m = tx;
    if (ty < m) { m = ty; a = 1; }
    if (tz < m) { a = 2; }
    if (a == 0) { cx += sx; tx += dx; }
    else if (a == 1) { cy += sy; ty += dy; }
    else { cz += sz; tz += dz; }
    return true;
  }
  // Same step, reporting which axis moved (0/1/2) instead of the voxel it left.
  __device__ inline bool next_axis(int* axis) {
    if (cur++ > steps) return false;
    int a = 0;
    float m = tx;
    if (ty < m) { m = ty; a = 1; }
    if (tz < m) { a = 2; }
    if (a == 0) { cx += sx; tx += dx; }
    else if (a == 1) { cy += sy; ty += dy; }
    else { cz += sz; tz += dz; }
    *axis = a;
    return true;
  }
};

// The DDA walk of a RayCaster as (BlockIndex, linear VoxelIndex) with branch-free steps.
// getBlockIndexFromGlobalVoxelIndex (common.h:215-224) is floor(float(g) * (1/vps)) and the
// local index (g + INT_MIN) & (vps - 1) (:233-243); for a power-of-two vps and |g| < 2^24 both
// are exact integer floor division / remainder, so stepping them by +-1 gives the same values as
// re-deriving them from the int64 index at every step.  The axis choice is RayCaster::next's
// (Eigen minCoeff: strict '<' against the running minimum, first index wins ties), the t updates
// are the same single float additions; everything is selects, because with one ray per lane a
// branchy step costs the wave every path (measured: ~270 instructions per step before, ~60 now).
struct BlockWalk {
  float tx, ty, tz, dx, dy, dz;
  int sx, sy, sz;
  int bx, by, bz;
  uint32_t lin;   // lx + vps * (ly + lz * vps), block_inl.h:15-17
  uint32_t h;     // LongIndexHash of the current voxel (block_hash.h:54-64): x + 17191 y + 17191^2 z mod 2^32, so a
                  // step along one axis adds or subtracts that axis' multiplier
  bool entered;   // the current voxel is the walk's first or lies in another block than the previous one

  __device__ inline void start(const RayCaster& rc, int vps, float vps_inv) {
    tx = rc.tx; ty = rc.ty; tz = rc.tz; dx = rc.dx; dy = rc.dy; dz = rc.dz;
    sx = rc.sx; sy = rc.sy; sz = rc.sz;
    const l3 g{rc.cx, rc.cy, rc.cz};
    const i3 b = block_index_from_global(g, vps_inv);
    const i3 l = local_from_global(g, vps);
    bx = b.x; by = b.y; bz = b.z;
    lin = (uint32_t)(l.x + vps * (l.y + l.z * vps));
    h = long_index_hash(g);
    entered = true;
  }
  // One DDA step (integrator_utils.cc:111-125) from the current voxel to the next.
  __device__ inline void step(int vps, int vps_log2) {
    const bool y_lt = ty < tx;
    const float m = y_lt ? ty : tx;
    const bool z_lt = tz < m;
    const int a = z_lt ? 2 : (y_lt ? 1 : 0);
    const float ntx = tx + dx, nty = ty + dy, ntz = tz + dz;
    tx = (a == 0) ? ntx : tx;
    ty = (a == 1) ? nty : ty;
    tz = (a == 2) ? ntz : tz;
    const int s = (a == 0) ? sx : ((a == 1) ? sy : sz);
    h += (uint32_t)s * ((a == 0) ? 1u : ((a == 1) ? 17191u : 295530481u));
    const int shift = a * vps_log2;
    const int la = (int)((lin >> shift) & (uint32_t)(vps - 1)) + s;
    const bool lo = la < 0, hi = la >= vps;
    const int wrap = (lo ? vps : 0) - (hi ? vps : 0);  // local coordinate re-enters from the other face
    lin += (uint32_t)((s + wrap) * (1 << shift));
    const int db = (hi ? 1 : 0) - (lo ? 1 : 0);
    bx += (a == 0) ? db : 0;
    by += (a == 1) ? db : 0;
    bz += (a == 2) ? db : 0;
    entered = lo || hi;
  }
};

}  // namespace vbx
