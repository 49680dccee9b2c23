// ecs.hpp -- Earth coordinate system: maps the user's grid coordinates
// (ENU or range/azimuth/elevation, optionally curved or Earth-flattened) to
// the Cartesian model space the transport engine works in, and provides the
// local up/north/east frames used for seismometer axes and moment-tensor
// rotation.
//
// Public surface mirrors the reference's EarthCoords + global `ECS`
// (reference ecs.hpp:34-473; mappings ecs.cpp:319-372; frames :147-306;
// Earth-flattening :540-677) for the subset the model builder needs.
#ifndef R3DH_ECS_HPP_
#define R3DH_ECS_HPP_

#include "elastic.hpp"
#include "geom.hpp"

class EarthCoords {
 public:
  // Untyped coordinate triple in the user's chosen scheme.
  class Generic {
    Real a = 0, b = 0, c = 0;

   public:
    Generic() = default;
    Generic(Real x1, Real x2, Real x3) : a(x1), b(x2), c(x3) {}
    void SetTriple(Real x1, Real x2, Real x3) { a = x1, b = x2, c = x3; }
    Real x1() const { return a; }
    Real x2() const { return b; }
    Real x3() const { return c; }
    bool IsNull() const { return a == 0.0 && b == 0.0 && c == 0.0; }
    Real Radius(const EarthCoords& ecs) const { return ecs.ExtractRadius(*this); }
  };

  enum earthcoords_e { ENU_ORTHO, RAE_ORTHO, RAE_CURVED, RAE_SPHERICAL, MAP_UNSUPPORTED };
  enum outcoords_e { OUT_NOTRANSFORM, OUT_ECS, OUT_ENU_ORTHO };

  void SetEarthFlattening(bool on) { mFlatten = on; }
  void SetEarthRadius(Real r) { mRadE = r; }
  void SetMapping(earthcoords_e m) { mMap = m; }
  void SetOCSMapping(outcoords_e m) { mOut = m; }
  earthcoords_e Mapping() const { return mMap; }
  Real GetEarthRadius() const { return mRadE; }
  bool IsEarthFlattening() const { return mFlatten; }
  bool CurvedCoords() const { return mMap == RAE_CURVED || mMap == RAE_SPHERICAL; }

  // Reference points (only meaningful for the curved mappings).
  R3::XYZ GetEarthCenter() const;
  R3::XYZ GetNorthPole() const;

  Real ExtractElevation(Generic g) const { return g.x3(); }
  Real ExtractRadius(Generic g) const;

  R3::XYZ GetUp(R3::XYZ from) const;
  R3::XYZ GetNorth(R3::XYZ from) const;
  R3::XYZ GetEast(R3::XYZ from) const;
  R3::XYZ GetDown(R3::XYZ from) const { return GetUp(from).Negative(); }
  R3::XYZ GetSouth(R3::XYZ from) const { return GetNorth(from).Negative(); }
  R3::XYZ GetRadial(R3::XYZ ref, R3::XYZ from) const;
  R3::XYZ GetTransverse(R3::XYZ ref, R3::XYZ from) const;

  R3::XYZ Convert(Generic g) const;                               // grid -> model space
  Elastic::HElastic Convert(Generic g, Elastic::HElastic p) const; // ... for properties
  Generic OutConvert(R3::XYZ loc) const;                          // model -> output coords
  Generic OutConvertDirectional(R3::XYZ loc, R3::XYZ dir) const;  // direction at loc -> output axes
  Elastic::HElastic OutConvert(R3::XYZ loc, Elastic::HElastic p) const;

  Real FlattenDepth(Real z) const { return mRadE * std::log((mRadE + z) / mRadE); }
  Real UnflattenDepth(Real zf) const { return mRadE * std::expm1(zf / mRadE); }

  // Rotation taking moment tensors given in local North-East-Down axes to
  // model XYZ at `from` (reference ecs.cpp:704-722).
  R3::Matrix GetXYZToLocalNEDRotation(R3::XYZ from) const;

  // Restore defaults (the reference's ECS is a process-wide singleton that is
  // only ever configured once; the library can build several models).
  void Reset() { *this = EarthCoords(); }

 private:
  void need_curved(const char* what) const;
  earthcoords_e mMap = ENU_ORTHO;
  outcoords_e mOut = OUT_ENU_ORTHO;
  bool mFlatten = false;
  Real mRadE = 6371.0;
};

extern EarthCoords ECS;

#endif
