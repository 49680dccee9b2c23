// geom.hpp -- small vector/matrix/spherical-angle toolkit for the HOST model
// builder (cell construction, coordinate mapping, seismometer axes, moment
// tensors).  The device hot path has its own math (csrc/r3d_math.h).
//
// Class and method names follow the reference's R3::XYZ / R3::Matrix /
// S2::ThetaPhi (reference geom_r3.hpp:51-330, geom_s2.hpp:88-150) so that
// model-definition code written against the reference reads the same here.
#ifndef R3DH_GEOM_HPP_
#define R3DH_GEOM_HPP_

#include <cmath>
#include <ostream>
#include <vector>

#include "typedefs.hpp"

namespace Geometry {
constexpr Real Pi = 3.14159265358979323846;
constexpr Real Pi45 = Pi * 0.25;
constexpr Real Pi90 = Pi * 0.5;
constexpr Real Pi180 = Pi;
constexpr Real Pi270 = Pi * 1.5;
constexpr Real Pi360 = Pi * 2.0;
constexpr Real RtoD = 180.0 / Pi;
constexpr Real DtoR = Pi / 180.0;
}  // namespace Geometry

namespace S2 {
struct ThetaPhi {
  Real theta = 0, phi = 0;
  ThetaPhi() = default;
  ThetaPhi(Real t, Real p) : theta(t), phi(p) {}
  Real Theta() const { return theta; }
  Real Phi() const { return phi; }
  Real x() const { return std::sin(theta) * std::cos(phi); }
  Real y() const { return std::sin(theta) * std::sin(phi); }
  Real z() const { return std::cos(theta); }
};

// Take-off-angle set: centre directions of a recursively quadrisected
// icosahedron, 20*4^degree points (reference geom_s2.cpp:60-130, 251-292).
// Midpoints are projected onto the unit sphere at every level; the centre of
// a leaf triangle is the normalised vertex sum.  Order differs from the
// reference's (depth-first per face here as well, but nothing statistical
// depends on it).
std::vector<ThetaPhi> TesselSphereIco(int degree);
}  // namespace S2

namespace R3 {

class Matrix;

class XYZ {
 protected:
  Real mX = 0, mY = 0, mZ = 0;

 public:
  XYZ() = default;
  XYZ(Real x, Real y, Real z) : mX(x), mY(y), mZ(z) {}
  XYZ(const S2::ThetaPhi& a) : mX(a.x()), mY(a.y()), mZ(a.z()) {}

  Real x() const { return mX; }
  Real y() const { return mY; }
  Real z() const { return mZ; }
  void SetXYZ(Real x, Real y, Real z) { mX = x, mY = y, mZ = z; }

  Real MagSquared() const { return mX * mX + mY * mY + mZ * mZ; }
  Real Mag() const { return std::sqrt(MagSquared()); }
  bool IsZero() const { return mX == 0 && mY == 0 && mZ == 0; }
  bool IsSquaredZero() const { return MagSquared() == 0; }
  Real Theta() const { return IsSquaredZero() ? 0 : std::acos(mZ / Mag()); }
  Real Phi() const { return std::atan2(mY, mX); }

  XYZ Unit() const {
    Real s = 1.0 / Mag();
    return {mX * s, mY * s, mZ * s};
  }
  XYZ UnitElse(const XYZ& fallback) const {
    Real m = Mag();
    if (m == 0) return fallback;
    Real s = 1.0 / m;
    return {mX * s, mY * s, mZ * s};
  }
  void Normalize() { *this = Unit(); }
  XYZ Negative() const { return {-mX, -mY, -mZ}; }
  XYZ ScaledBy(Real s) const { return {s * mX, s * mY, s * mZ}; }
  Real Dot(const XYZ& o) const { return o.mX * mX + o.mY * mY + o.mZ * mZ; }
  XYZ Cross(const XYZ& o) const {
    return {mY * o.mZ - mZ * o.mY, mZ * o.mX - mX * o.mZ,
            mX * o.mY - mY * o.mX};
  }
  XYZ VectorTo(const XYZ& o) const { return {o.mX - mX, o.mY - mY, o.mZ - mZ}; }
  Real DistFrom(const XYZ& o) const { return VectorTo(o).Mag(); }
  XYZ operator+(const XYZ& o) const { return {mX + o.mX, mY + o.mY, mZ + o.mZ}; }
  XYZ operator-(const XYZ& o) const { return {mX - o.mX, mY - o.mY, mZ - o.mZ}; }
  Matrix Outer(const XYZ& o) const;
};

class Matrix {
 protected:
  Real m[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};

 public:
  Matrix() = default;
  Matrix(Real xx, Real xy, Real xz, Real yx, Real yy, Real yz, Real zx,
         Real zy, Real zz)
      : m{{xx, xy, xz}, {yx, yy, yz}, {zx, zy, zz}} {}
  Real xx() const { return m[0][0]; }
  Real xy() const { return m[0][1]; }
  Real xz() const { return m[0][2]; }
  Real yx() const { return m[1][0]; }
  Real yy() const { return m[1][1]; }
  Real yz() const { return m[1][2]; }
  Real zx() const { return m[2][0]; }
  Real zy() const { return m[2][1]; }
  Real zz() const { return m[2][2]; }
  Real at(int r, int c) const { return m[r][c]; }

  Matrix T() const {
    return {m[0][0], m[1][0], m[2][0], m[0][1], m[1][1],
            m[2][1], m[0][2], m[1][2], m[2][2]};
  }
  Real Trace() const { return m[0][0] + m[1][1] + m[2][2]; }
  Real Mag2() const {
    Real s = 0;
    for (auto& r : m)
      for (Real v : r) s += v * v;
    return s;
  }
  Real Mag() const { return std::sqrt(Mag2()); }
  Matrix& operator*=(Real s) {
    for (auto& r : m)
      for (Real& v : r) v *= s;
    return *this;
  }
  Matrix& operator+=(const Matrix& o) {
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) m[r][c] += o.m[r][c];
    return *this;
  }
  Matrix& operator*=(const Matrix& o) {
    Matrix t;
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++)
        t.m[r][c] = m[r][0] * o.m[0][c] + m[r][1] * o.m[1][c] +
                    m[r][2] * o.m[2][c];
    return *this = t;
  }
  XYZ operator*(const XYZ& v) const {
    return {m[0][0] * v.x() + m[0][1] * v.y() + m[0][2] * v.z(),
            m[1][0] * v.x() + m[1][1] * v.y() + m[1][2] * v.z(),
            m[2][0] * v.x() + m[2][1] * v.y() + m[2][2] * v.z()};
  }
  void ScaleBy(Real s) { *this *= s; }
  void OutputContents(std::ostream& out) const;   // labelled dump (reference geom_r3.cpp:174-190)
  // Rescale so that the Frobenius norm squared becomes n2
  // (reference geom_r3.hpp:382-385).
  void SetSquaredMag(Real n2) {
    if (n2 < 0) throw std::domain_error("Matrix::SetSquaredMag: negative arg");
    ScaleBy(std::sqrt(n2 / Mag2()));
  }
  // this <- M this M^T   (reference geom_r3.hpp:387-391)
  void Transform(Matrix M) {
    (*this) *= M.T();
    M *= (*this);
    *this = M;
  }
};

inline void Matrix::OutputContents(std::ostream& out) const {
  Real mag = Mag(), tr = Trace();
  Real iso = tr * tr / (3.0 * mag * mag);
  if (tr < 0) iso *= -1;
  out << "The contents of this matrix are:" << std::endl
      << "\tx\ty\tz\t" << std::endl
      << "    +---------------------------" << std::endl
      << "  x |\t" << m[0][0] << "\t" << m[0][1] << "\t" << m[0][2] << "\t\tMagnitude: " << mag << "\n"
      << "  y |\t" << m[1][0] << "\t" << m[1][1] << "\t" << m[1][2] << "\t\tTrace:     " << tr << "\n"
      << "  z |\t" << m[2][0] << "\t" << m[2][1] << "\t" << m[2][2] << "\t\tIso Frac:  " << iso << "\n"
      << std::endl;
}
inline Matrix operator*(Matrix a, const Matrix& b) { return a *= b; }
inline Matrix operator*(Matrix a, Real s) { return a *= s; }
inline Matrix operator*(Real s, Matrix a) { return a *= s; }
inline Matrix operator+(Matrix a, const Matrix& b) { return a += b; }

inline Matrix XYZ::Outer(const XYZ& o) const {
  return {mX * o.mX, mX * o.mY, mX * o.mZ, mY * o.mX, mY * o.mY,
          mY * o.mZ, mZ * o.mX, mZ * o.mY, mZ * o.mZ};
}

}  // namespace R3

#endif
