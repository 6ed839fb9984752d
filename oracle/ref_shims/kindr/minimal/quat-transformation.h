// DEPENDENCY SHIM (oracle/_ref build only): the subset of minkindr's QuatTransformation used
// on the integration path.  transform(v) = q.rotate(v) + t, rotate = Eigen Quaternion * Vector3
// (restated from minkindr's published sources; written for this repo).
#pragma once
#include <Eigen/Core>
namespace kindr {
namespace minimal {
template <typename S>
class RotationQuaternionTemplate {
 public:
  typedef Eigen::Quaternion<S> Implementation;
  typedef Eigen::Matrix<S, 3, 1> Vector3;
  RotationQuaternionTemplate() {}
  explicit RotationQuaternionTemplate(const Implementation& q) : q_(q) {}
  RotationQuaternionTemplate(S w, S x, S y, S z) : q_(w, x, y, z) {}
  Vector3 rotate(const Vector3& v) const { return q_ * v; }
  const Implementation& toImplementation() const { return q_; }
  RotationQuaternionTemplate inverse() const { return RotationQuaternionTemplate(q_.conjugate()); }
 private:
  Implementation q_;
};
template <typename S>
class QuatTransformationTemplate {
 public:
  typedef Eigen::Matrix<S, 3, 1> Position;
  typedef RotationQuaternionTemplate<S> Rotation;
  QuatTransformationTemplate() {}
  QuatTransformationTemplate(const Rotation& q, const Position& t) : q_(q), t_(t) {}
  const Position& getPosition() const { return t_; }
  const Rotation& getRotation() const { return q_; }
  Position transform(const Position& v) const { return q_.rotate(v) + t_; }
  Position operator*(const Position& v) const { return transform(v); }
  QuatTransformationTemplate inverse() const {
    return QuatTransformationTemplate(q_.inverse(), -q_.inverse().rotate(t_));
  }
 private:
  Rotation q_;
  Position t_;
};
}  // namespace minimal
}  // namespace kindr
