// include/fiesta/vec3.h -- the two vector types the fiesta::ESDFMap surface is written against.
// With Eigen on the include path these are Eigen::Vector3d / Eigen::Vector3i (as in the reference,
// include/ESDFMap.h:8); without it (this build image ships no Eigen) a minimal stand-in with the members
// the facade and its callers use: (i) element access, 3-argument constructor, + - scalar * /.
#pragma once
#if defined(FIESTA_USE_EIGEN) || (__has_include(<Eigen/Eigen>) && !defined(FIESTA_NO_EIGEN))
#include <Eigen/Eigen>
#else
#include <cmath>
namespace Eigen {
template <typename T>
struct FiestaVec3 {
  T v[3];
  FiestaVec3() : v{0, 0, 0} {}
  FiestaVec3(T x, T y, T z) : v{x, y, z} {}
  T &operator()(int i) { return v[i]; }
  const T &operator()(int i) const { return v[i]; }
  T &operator[](int i) { return v[i]; }
  const T &operator[](int i) const { return v[i]; }
  T &x() { return v[0]; }
  T &y() { return v[1]; }
  T &z() { return v[2]; }
  const T &x() const { return v[0]; }
  const T &y() const { return v[1]; }
  const T &z() const { return v[2]; }
  FiestaVec3 operator+(const FiestaVec3 &o) const { return {T(v[0] + o.v[0]), T(v[1] + o.v[1]), T(v[2] + o.v[2])}; }
  FiestaVec3 operator-(const FiestaVec3 &o) const { return {T(v[0] - o.v[0]), T(v[1] - o.v[1]), T(v[2] - o.v[2])}; }
  FiestaVec3 operator*(T s) const { return {T(v[0] * s), T(v[1] * s), T(v[2] * s)}; }
  FiestaVec3 operator/(T s) const { return {T(v[0] / s), T(v[1] / s), T(v[2] / s)}; }
  bool operator==(const FiestaVec3 &o) const { return v[0] == o.v[0] && v[1] == o.v[1] && v[2] == o.v[2]; }
  double norm() const { return std::sqrt((double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2]); }
};
typedef FiestaVec3<double> Vector3d;
typedef FiestaVec3<int> Vector3i;
}  // namespace Eigen
#endif
