// ecs.cpp -- see ecs.hpp.  Behaviour follows reference ecs.cpp (cited per
// function); written table-free and without the reference's position cache.
#include "ecs.hpp"

EarthCoords ECS;

void EarthCoords::need_curved(const char* what) const {
  if (!CurvedCoords())
    throw Invalid(Text("Invalid Mapping Operation in current map mode: ") + what);
}

// reference ecs.cpp:107-136 (RefreshCache): curved maps put the model origin
// on the surface above the Earth's centre; spherical maps put it AT the centre.
R3::XYZ EarthCoords::GetEarthCenter() const {
  need_curved("EarthCenter");
  return mMap == RAE_CURVED ? R3::XYZ(0, 0, -mRadE) : R3::XYZ(0, 0, 0);
}
R3::XYZ EarthCoords::GetNorthPole() const {
  need_curved("NorthPole");
  return mMap == RAE_CURVED ? R3::XYZ(0, mRadE, -mRadE) : R3::XYZ(0, mRadE, 0);
}

// reference ecs.cpp:60-83
Real EarthCoords::ExtractRadius(Generic g) const {
  need_curved("ExtractRadius");
  Real r = mRadE + g.x3();
  if (r < 0)
    throw Runtime("ECS: Elevation component implies a negative Radius component. "
                  "Check model grid, or check that correct Earth radius specified.");
  return r;
}

// reference ecs.cpp:147-167
R3::XYZ EarthCoords::GetUp(R3::XYZ loc) const {
  if (!CurvedCoords()) return {0, 0, 1};
  return GetEarthCenter().VectorTo(loc).UnitElse(R3::XYZ(0, 1, 0));
}
// reference ecs.cpp:176-196
R3::XYZ EarthCoords::GetNorth(R3::XYZ loc) const {
  if (!CurvedCoords()) return {0, 1, 0};
  return GetUp(loc).Cross(GetEast(loc));
}
// reference ecs.cpp:205-231
R3::XYZ EarthCoords::GetEast(R3::XYZ loc) const {
  if (!CurvedCoords()) return {1, 0, 0};
  R3::XYZ chord_north = loc.VectorTo(GetNorthPole());
  R3::XYZ upward = GetEarthCenter().VectorTo(loc);
  return chord_north.Cross(upward).UnitElse(R3::XYZ(1, 0, 0));
}
// reference ecs.cpp:242-262
R3::XYZ EarthCoords::GetRadial(R3::XYZ ref, R3::XYZ loc) const {
  return GetUp(loc).Cross(GetTransverse(ref, loc));
}
// reference ecs.cpp:275-306: transverse = chord x up (clockwise seen from
// above), falling back on South when loc sits on top of ref.
R3::XYZ EarthCoords::GetTransverse(R3::XYZ ref, R3::XYZ loc) const {
  R3::XYZ t = ref.VectorTo(loc).Cross(GetUp(loc));
  if (t.IsSquaredZero()) t = GetSouth(loc);
  return t.Unit();
}

// reference ecs.cpp:319-372
R3::XYZ EarthCoords::Convert(Generic g) const {
  switch (mMap) {
    case ENU_ORTHO:
      return {g.x1(), g.x2(), mFlatten ? FlattenDepth(g.x3()) : g.x3()};
    case RAE_ORTHO: {
      Real phi = Geometry::DtoR * (90.0 - g.x2());
      return {g.x1() * std::cos(phi), g.x1() * std::sin(phi),
              mFlatten ? FlattenDepth(g.x3()) : g.x3()};
    }
    case RAE_CURVED:
    case RAE_SPHERICAL: {
      Real theta = g.x1() / mRadE;
      Real phi = Geometry::DtoR * (90.0 - g.x2());
      Real r = mRadE + g.x3();
      if (r < 0)
        throw Runtime("ECS: Elevation component implies a negative Radius component. "
                      "Check model grid, or check that correct Earth radius specified.");
      if (theta > Geometry::Pi180)
        throw Runtime("ECS: Range component exceeds half-circumference. "
                      "Check model grid, or check that correct Earth radius specified.");
      return {r * std::sin(theta) * std::cos(phi), r * std::sin(theta) * std::sin(phi),
              r * std::cos(theta) + GetEarthCenter().z()};
    }
    default:
      throw std::invalid_argument("ECS Convert: Unknown Map Code");
  }
}

// reference ecs.cpp:465-476 + :564-677: Earth-flattening scales velocities by
// R/(R+z) and leaves density, Q and heterogeneity untouched.
Elastic::HElastic EarthCoords::Convert(Generic g, Elastic::HElastic p) const {
  if (!mFlatten) return p;
  Real f = mRadE / (mRadE + ExtractElevation(g));
  return Elastic::HElastic(Elastic::VpVs(p.Vp() * f, p.Vs() * f), p.getDens(), p.getQ(),
                           p.getHS());
}

// reference ecs.cpp:394-434
EarthCoords::Generic EarthCoords::OutConvert(R3::XYZ p) const {
  if (mOut == OUT_NOTRANSFORM) return {p.x(), p.y(), p.z()};
  if (mOut == OUT_ECS) throw Runtime("ECS BackConvert: Unimplemented for this mapping");
  if (CurvedCoords()) {
    Real zeta = p.z() - GetEarthCenter().z();
    Real r = std::sqrt(p.x() * p.x() + p.y() * p.y() + zeta * zeta);
    Real rxy = std::sqrt(p.x() * p.x() + p.y() * p.y());
    Real ranges = mRadE * std::acos(zeta / r);
    return {rxy > 0 ? ranges * (p.x() / rxy) : 0, rxy > 0 ? ranges * (p.y() / rxy) : 0,
            r - mRadE};
  }
  if (mFlatten) return {p.x(), p.y(), UnflattenDepth(p.z())};
  return {p.x(), p.y(), p.z()};
}

// reference ecs.cpp:436-460: local (east, north, up) components unless --ocsraw
EarthCoords::Generic EarthCoords::OutConvertDirectional(R3::XYZ loc, R3::XYZ dir) const {
  if (mOut == OUT_NOTRANSFORM) return {dir.x(), dir.y(), dir.z()};
  if (mOut == OUT_ECS) throw Runtime("ECS OutConvertDirectional: Unimplemented for this mapping");
  return {dir.Dot(GetEast(loc)), dir.Dot(GetNorth(loc)), dir.Dot(GetUp(loc))};
}

// reference ecs.cpp:478-505 + :579-590
Elastic::HElastic EarthCoords::OutConvert(R3::XYZ loc, Elastic::HElastic p) const {
  if (mOut == OUT_NOTRANSFORM || !mFlatten) return p;
  Real f = (mRadE + UnflattenDepth(loc.z())) / mRadE;
  return Elastic::HElastic(Elastic::VpVs(p.Vp() * f, p.Vs() * f), p.getDens(), p.getQ(),
                           p.getHS());
}

// reference ecs.cpp:704-722: M[i][j] = X_i . N_j with N = (north, east, down).
R3::Matrix EarthCoords::GetXYZToLocalNEDRotation(R3::XYZ from) const {
  R3::XYZ n = GetNorth(from), e = GetEast(from), d = GetDown(from);
  return {n.x(), e.x(), d.x(), n.y(), e.y(), d.y(), n.z(), e.z(), d.z()};
}
