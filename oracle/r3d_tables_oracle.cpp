// r3d_tables_oracle.cpp -- CPU restatement of the reference's TABLE BUILDERS: everything the hot
// path reads that is computed once, before the first phonon -- the take-off-angle set, the
// scatterers' radiation-pattern tables / mean free paths / dipole moments, the event source's
// moment tensor and P / SH / SV patterns, the seismometers' axes and gather areas, and the cell arrays
// of the three model kinds (faces, links, flags, velocity / density fits, scatterer sharing).
//
// ***  TEST INFRASTRUCTURE ONLY.  ***  Nothing in the product may include, link or call this
// file; only tests/ use it, as the checker for the two table builders the product has (the
// host builder radiative3d_amd/host/ and the HIP builder csrc/r3d_tables_build.hip).
//
// PARITY STATUS: "parity unpinned" in the same sense as r3d_oracle.cpp (the reference cannot be
// built under this project's rules and ships no fixtures).  What pins this file: the scatterer
// row the survey recorded on the unmodified reference (Halfspace, TOA degree 9: MFP 1564.43 /
// 594.537, dipoles 0.6881 / 0.8776 -- tests/golden/reference_recorded.json, asserted on THIS
// code in tests/test_tables_oracle.py), the reference's own closed forms (sum of the P / SH / SV
// whole-space energies of a double couple = 2 : 3 split, tensors of unit Frobenius norm), and
// counts (20 * 4^degree take-off angles).
//
// Style: the reference's formulation, function by function, with its own operation order
// (pow(), tgamma(), running sums in index order, recursion for the tessellation); every
// function cites the lines it follows.  No attempt at speed.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../include/r3d.h"   // (plain-C table layouts only: r3d_cell / r3d_face are what the builders fill)

namespace {

const double PI = 3.14159265358979323846;   // Geometry::Pi, geom_base.hpp:32-44
const double DtoR = PI / 180.0;

// ------------------------------------------------------------------ vectors --
struct V {
  double x, y, z;
};
inline V mk(double x, double y, double z) { return V{x, y, z}; }
inline V operator-(V a, V b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V operator*(double s, V a) { return mk(s * a.x, s * a.y, s * a.z); }
inline double dot(V a, V b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V cross(V a, V b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline double mag(V a) { return std::sqrt(dot(a, a)); }
inline bool is_squared_zero(V a) { return dot(a, a) == 0; }   // XYZ::IsSquaredZero, geom_r3.hpp
inline V unit(V a) {
  double s = 1.0 / mag(a);
  return mk(a.x * s, a.y * s, a.z * s);
}
inline V unit_else(V a, V fb) {   // XYZ::UnitElse, geom_r3.hpp:127-132
  double m = mag(a);
  if (m == 0) return fb;
  double s = 1.0 / m;
  return mk(a.x * s, a.y * s, a.z * s);
}
inline V neg(V a) { return mk(-a.x, -a.y, -a.z); }

// ============================================================ take-off set ==
// S2::Node (geom_s2.hpp:108-164, geom_s2.cpp:296-351): a point that is normalised on
// construction from coordinates and on ASSIGNMENT, but not by operator+ and not by the copy
// constructor.
struct Node {
  double x = 0, y = 0, z = 0;
};
inline Node normalized(Node n) {   // Node::normalize, geom_s2.cpp:340-351
  if (n.x == 0 && n.y == 0 && n.z == 0) return n;
  const double norm = std::sqrt(n.x * n.x + n.y * n.y + n.z * n.z);
  n.x /= norm, n.y /= norm, n.z /= norm;
  return n;
}
inline Node node(double x, double y, double z) {   // Node(Real, Real, Real): normalises
  Node n;
  n.x = x, n.y = y, n.z = z;
  return normalized(n);
}
inline Node add(Node a, Node b) {   // Node::operator+ : plain sum
  Node n;
  n.x = a.x + b.x, n.y = a.y + b.y, n.z = a.z + b.z;
  return n;
}
// Triangle::Triangle / subdivide / populate_lists, geom_s2.cpp:221-292: the centres of the 4^degree
// leaf triangles in the order the recursion appends them (children (n1,q1,q3), (q1,n2,q2),
// (q3,q2,n3), (q2,q3,q1)); a centre is ThetaPhi(Node): theta = acos(z), phi = atan2(y, x)
// (geom_s2.hpp:202-205).
void triangle(Node n1, Node n2, Node n3, int degree, std::vector<double>& out) {
  if (degree > 0) {
    const Node q1 = normalized(add(n1, n2));   // "q1 = n1 + n2;" -- assignment normalises
    const Node q2 = normalized(add(n2, n3));
    const Node q3 = normalized(add(n3, n1));
    triangle(n1, q1, q3, degree - 1, out);
    triangle(q1, n2, q2, degree - 1, out);
    triangle(q3, q2, n3, degree - 1, out);
    triangle(q2, q3, q1, degree - 1, out);
    return;
  }
  const Node c = normalized(add(add(n1, n2), n3));   // "center = n1 + n2 + n3;"
  out.push_back(std::acos(c.z));
  out.push_back(std::atan2(c.y, c.x));
}
// TesselSphere::init_icosahedron, geom_s2.cpp:60-130: twelve corners, twenty faces in this order.
void tessellate(int degree, std::vector<double>& out) {
  const double phi = (1. + std::sqrt(5.0)) / 2.0;
  const Node NF = node(1, 0, phi), NB = node(-1, 0, phi), SF = node(1, 0, -phi), SB = node(-1, 0, -phi);
  const Node FL = node(phi, -1, 0), FR = node(phi, 1, 0), BL = node(-phi, -1, 0), BR = node(-phi, 1, 0);
  const Node RN = node(0, phi, 1), RS = node(0, phi, -1), LN = node(0, -phi, 1), LS = node(0, -phi, -1);
  const Node* fa[20][3] = {
      {&NF, &NB, &LN}, {&NF, &NB, &RN}, {&SF, &SB, &LS}, {&SF, &SB, &RS}, {&FL, &FR, &NF},
      {&FL, &FR, &SF}, {&BR, &BL, &NB}, {&BR, &BL, &SB}, {&LN, &LS, &FL}, {&LN, &LS, &BL},
      {&RN, &RS, &FR}, {&RN, &RS, &BR}, {&NF, &LN, &FL}, {&NF, &RN, &FR}, {&NB, &LN, &BL},
      {&NB, &RN, &BR}, {&SF, &LS, &FL}, {&SF, &RS, &FR}, {&SB, &LS, &BL}, {&SB, &RS, &BR}};
  for (int f = 0; f < 20; f++) triangle(*fa[f][0], *fa[f][1], *fa[f][2], degree, out);
}

// ============================================================== scatterers ==
struct Het {   // ScatterParams members, scatparams.hpp:60-77
  double nu, eps, a, kappa, el, gam0;
};
// ScatterParams::PSATO, scatparams.cpp:184-194 (the von Karman PSDF, Sato & Fehler 2.10)
double PSATO(const Het& h, double m) {
  const double pi32 = std::pow(PI, 1.5);
  const double numer = (8. * pi32 * h.eps * h.eps * h.a * h.a * h.a) * std::tgamma(h.kappa + 1.5) / std::tgamma(h.kappa);
  const double denom = std::pow((1. + h.a * h.a * m * m), (h.kappa + 1.5));
  return numer / denom;
}
// ScatterParams::XSATO, scatparams.cpp:138-165 (Sato & Fehler 4.50); psi = toa.Theta(), zeta = toa.Phi()
void XSATO(const Het& h, double theta, double phi, double& xpp, double& xps, double& xsp, double& xss_psi,
           double& xss_zeta) {
  const double gam0 = h.gam0, nu = h.nu;
  const double gam2 = gam0 * gam0;
  const double cpsi = std::cos(theta);
  const double c2psi = std::cos(2. * theta);
  const double spsi = std::sin(theta);
  const double czeta = std::cos(phi);
  const double szeta = std::sin(phi);
  const double spsi2 = spsi * spsi;
  xpp = (1. / gam2) * (nu * (-1. + cpsi + (2. / gam2) * spsi2) - 2. + (4. / gam2) * spsi2);
  xps = -spsi * (nu * (1. - (2. / gam0) * cpsi) - (4. / gam0) * cpsi);
  xsp = (1. / gam2) * spsi * czeta * (nu * (1. - (2. / gam0) * cpsi) - (4. / gam0) * cpsi);
  xss_psi = czeta * (nu * (cpsi - c2psi) - 2. * c2psi);
  xss_zeta = szeta * (nu * (cpsi - 1.) + 2. * cpsi);
}
// ScatterParams::GSATO, scatparams.cpp:75-118 (Sato & Fehler 4.52), with the "< 1e-30 -> 0" clamps
void GSATO(const Het& h, double theta, double phi, double& gpp, double& gps, double& gsp, double& gss,
           double& spol) {
  const double el = h.el, gam0 = h.gam0;
  const double pi4 = 4. * PI;
  const double el4 = std::pow(el, 4);
  const double gam2 = std::pow(gam0, 2);
  const double psi = theta;
  double xpp, xps, xsp, xss_psi, xss_zeta, arg;
  XSATO(h, theta, phi, xpp, xps, xsp, xss_psi, xss_zeta);
  const double xpp2 = xpp * xpp, xps2 = xps * xps, xsp2 = xsp * xsp;
  const double xss_psi2 = xss_psi * xss_psi, xss_zeta2 = xss_zeta * xss_zeta;
  arg = (2. * el / gam0) * std::sin(psi / 2.);
  gpp = (el4 / pi4) * xpp2 * PSATO(h, arg);
  if (gpp < 1.e-30) gpp = 0.;
  arg = (el / gam0) * std::sqrt(1. + gam2 - 2. * gam0 * std::cos(psi));
  gps = (1. / gam0) * (el4 / pi4) * xps2 * PSATO(h, arg);
  if (gps < 1.e-30) gps = 0.;
  gsp = gam0 * (el4 / pi4) * xsp2 * PSATO(h, arg);
  if (gsp < 1.e-30) gsp = 0.;
  arg = 2. * el * std::sin(psi / 2.);
  gss = (el4 / pi4) * (xss_psi2 + xss_zeta2) * PSATO(h, arg);
  if (gss < 1.e-30) gss = 0.;
  spol = std::atan2(xss_zeta, xss_psi);
}

// ProbDist::GetDiffProb on an integrated distribution, probability.cpp:60-78
double diff_prob(const double* cum, uint64_t n, uint64_t idx) {
  const double magn = cum[n - 1];
  const double prev = idx > 0 ? cum[idx - 1] : 0;
  const double diff = cum[idx] - prev;
  return magn == 0 ? 0 : diff / magn;
}

// ============================================================ event source ==
struct M3 {   // R3::Matrix, geom_r3.hpp
  double xx, xy, xz, yx, yy, yz, zx, zy, zz;
};
M3 mul(const M3& a, const M3& b) {   // Matrix::operator*=, geom_r3.hpp:400-413
  M3 t;
  t.xx = (a.xx * b.xx) + (a.xy * b.yx) + (a.xz * b.zx);
  t.xy = (a.xx * b.xy) + (a.xy * b.yy) + (a.xz * b.zy);
  t.xz = (a.xx * b.xz) + (a.xy * b.yz) + (a.xz * b.zz);
  t.yx = (a.yx * b.xx) + (a.yy * b.yx) + (a.yz * b.zx);
  t.yy = (a.yx * b.xy) + (a.yy * b.yy) + (a.yz * b.zy);
  t.yz = (a.yx * b.xz) + (a.yy * b.yz) + (a.yz * b.zz);
  t.zx = (a.zx * b.xx) + (a.zy * b.yx) + (a.zz * b.zx);
  t.zy = (a.zx * b.xy) + (a.zy * b.yy) + (a.zz * b.zy);
  t.zz = (a.zx * b.xz) + (a.zy * b.yz) + (a.zz * b.zz);
  return t;
}
M3 transpose(const M3& a) { return M3{a.xx, a.yx, a.zx, a.xy, a.yy, a.zy, a.xz, a.yz, a.zz}; }
M3 transform(const M3& self, const M3& m) {   // Matrix::Transform, geom_r3.hpp:390-397: M self M^T
  return mul(m, mul(self, transpose(m)));
}
double mag2(const M3& a) {   // Frobenius self-product, geom_r3.hpp:341-357
  return a.xx * a.xx + a.xy * a.xy + a.xz * a.xz + a.yx * a.yx + a.yy * a.yy + a.yz * a.yz + a.zx * a.zx +
         a.zy * a.zy + a.zz * a.zz;
}
M3 scaled(const M3& a, double s) {
  return M3{a.xx * s, a.xy * s, a.xz * s, a.yx * s, a.yy * s, a.yz * s, a.zx * s, a.zy * s, a.zz * s};
}
M3 plus(const M3& a, const M3& b) {
  return M3{a.xx + b.xx, a.xy + b.xy, a.xz + b.xz, a.yx + b.yx, a.yy + b.yy, a.yz + b.yz, a.zx + b.zx, a.zy + b.zy,
            a.zz + b.zz};
}
M3 set_squared_mag(const M3& a, double n2) { return scaled(a, std::sqrt(n2 / mag2(a))); }   // geom_r3.hpp:385-388
// Tensor::USGS, tensors.hpp:147-155 (Harvard / USGS r, theta, phi convention -> x = N, y = E, z = D)
M3 usgs(double rr, double tt, double pp, double rt, double rp, double tp) {
  return M3{tt, -tp, rt, -tp, pp, -rp, rt, -rp, rr};
}
// Tensor::EulerSDR, tensors.hpp:182-204
M3 euler_sdr(double alpha, double beta, double gamma) {
  const double ca = std::cos(alpha), cb = std::cos(beta), cg = std::cos(gamma);
  const double sa = std::sin(alpha), sb = std::sin(beta), sg = std::sin(gamma);
  M3 m;
  m.xx = cb * cg - ca * sb * sg;
  m.xy = -cb * sg - ca * sb * cg;
  m.xz = sa * sb;
  m.yx = sb * cg + ca * cb * sg;
  m.yy = -sb * sg + ca * cb * cg;
  m.yz = -sa * cb;
  m.zx = sa * sg;
  m.zy = sa * cg;
  m.zz = ca;
  return m;
}
// Tensor::SDR, tensors.hpp:242-280 (degrees; iso = signed isotropic energy fraction)
M3 sdr(double strike, double dip, double rake, double iso, double moment) {
  M3 t{0, 0, -1, 0, 0, 0, -1, 0, 0};
  strike *= DtoR, dip *= DtoR, rake *= DtoR;
  t = transform(t, euler_sdr(dip, strike, -rake));
  const double I2 = std::fabs(iso), D2 = 1.0 - I2;
  t = set_squared_mag(t, D2);
  iso = (iso >= 0) ? 1 : -1;
  M3 ISO{iso, 0, 0, 0, iso, 0, 0, 0, iso};
  ISO = set_squared_mag(ISO, I2);
  return scaled(plus(t, ISO), moment);
}

// =================================================== Earth coordinate frames ==
// EarthCoords, ecs.cpp:100-306.  map: 0 ENU_ORTHO, 1 RAE_ORTHO, 2 RAE_CURVED, 3 RAE_SPHERICAL
// (ecs.hpp:242-257); model space is Cartesian in all of them.
struct Ecs {
  int map;
  double radE;
  bool curved() const { return map >= 2; }
  V center() const { return map == 2 ? mk(0, 0, -radE) : mk(0, 0, 0); }        // :100-128
  V north_pole() const { return map == 2 ? mk(0, radE, -radE) : mk(0, radE, 0); }
  V up(V loc) const {                                                         // GetUp :147-167
    if (!curved()) return mk(0, 0, +1);
    return unit_else(loc - center(), mk(0, 1, 0));
  }
  V east(V loc) const {                                                       // GetEast :190-212
    if (!curved()) return mk(+1, 0, 0);
    const V chord_north = north_pole() - loc;
    const V upward = loc - center();
    return unit_else(cross(chord_north, upward), mk(1, 0, 0));
  }
  V north(V loc) const {                                                      // GetNorth :169-188
    if (!curved()) return mk(0, +1, 0);
    return cross(up(loc), east(loc));
  }
  V south(V loc) const { return neg(north(loc)); }
  V transverse(V ref, V loc) const {                                          // GetTransverse :262-306
    const V chord_radial = loc - ref;
    V t = cross(chord_radial, up(loc));
    if (is_squared_zero(t)) t = south(loc);
    return unit(t);
  }
  V radial(V ref, V loc) const { return cross(up(loc), transverse(ref, loc)); }   // GetRadial :242-260
  M3 xyz_to_ned(V from) const {                                               // :704-722
    const V n1 = north(from), n2 = east(from), n3 = neg(up(from));
    // sum over |Xi> <Xi|Na> <Xa| : element (i, a) = Xi . Na
    return M3{n1.x, n2.x, n3.x, n1.y, n2.y, n3.y, n1.z, n2.z, n3.z};
  }
};

}  // namespace

extern "C" {

// S2::TesselSphere(TESS_ICO, degree) (geom_s2.cpp:40-130): fills toa[2 * 20 * 4^degree] with
// (theta, phi) pairs in the reference's order; returns the number of take-off angles.
uint64_t r3d_oracle_toa(int degree, double* toa) {
  std::vector<double> out;
  out.reserve((size_t)40 << (2 * degree));
  tessellate(degree, out);
  if (toa) std::memcpy(toa, out.data(), out.size() * sizeof(double));
  return out.size() / 2;
}

// ScatterParams::GSATO at one take-off angle (scatparams.cpp:75-118): out = gpp, gps, gsp, gss, spol;
// het = nu, eps, a, kappa, el, gam0.
void r3d_oracle_gsato(const double het[6], double theta, double phi, double out[5]) {
  const Het h{het[0], het[1], het[2], het[3], het[4], het[5]};
  GSATO(h, theta, phi, out[0], out[1], out[2], out[3], out[4]);
}

// Scatterer::Scatterer (scatterers.cpp:97-122) over a take-off set of n (theta, phi) pairs:
//   cdf[k]   the integrated GPP, GPS, GSP, GSS distributions (PopulateProbDists :134-161 +
//            ProbDist::Integrate, probability.cpp:21-35), n doubles each; spol likewise (may be NULL)
//   whole    mWholeProbs[IN_P], [IN_S] integrated (PopulateWholeProbs :172-184)
//   mfp      ComputeMFPs :195-220 (1 / (sum of G / nTOA)), or the override (--overridemfp)
//   dipole   ComputeDipoles :244-259, or 1, 1 under --nodeflect
void r3d_oracle_scatterer(const double het[6], const double* toa, uint64_t n, int mfp_override,
                          const double mfp_given[2], int no_deflect, double* cdf[4], double* spol,
                          double whole[2][4], double mfp[2], double dipole[2]) {
  const Het h{het[0], het[1], het[2], het[3], het[4], het[5]};
  for (uint64_t k = 0; k < n; k++) {
    double g[4], sp;
    GSATO(h, toa[2 * k], toa[2 * k + 1], g[0], g[1], g[2], g[3], sp);
    for (int c = 0; c < 4; c++) cdf[c][k] = g[c];
    if (spol) spol[k] = sp;
  }
  for (int c = 0; c < 4; c++)
    for (uint64_t i = 1; i < n; i++) cdf[c][i] += cdf[c][i - 1];
  const double magn[4] = {cdf[0][n - 1], cdf[1][n - 1], cdf[2][n - 1], cdf[3][n - 1]};
  const double wp[2][4] = {{magn[0], magn[1], 0, 0}, {0, 0, magn[2], magn[3]}};
  for (int in = 0; in < 2; in++) {
    whole[in][0] = wp[in][0];
    for (int c = 1; c < 4; c++) whole[in][c] = whole[in][c - 1] + wp[in][c];
  }
  if (!mfp_override) {
    double imfp_p = whole[0][3], imfp_s = whole[1][3];
    imfp_p /= (double)(int)n;   // "imfp_p /= nTOA" with int nTOA
    imfp_s /= (double)(int)n;
    mfp[0] = 1.0 / imfp_p, mfp[1] = 1.0 / imfp_s;
  } else {
    mfp[0] = mfp_given[0], mfp[1] = mfp_given[1];
  }
  if (!no_deflect) {
    double moments[4] = {0, 0, 0, 0};
    for (uint64_t idx = 0; idx < n; idx++) {
      // toa[idx].z() of S2Point = cos(theta) through XYZ(ThetaPhi) (geom_s2.cpp:132-136)
      const double costh = std::cos(toa[2 * idx]);
      for (int c = 0; c < 4; c++) moments[c] += costh * diff_prob(cdf[c], n, idx);
    }
    dipole[0] = moments[0] * diff_prob(whole[0], 4, 0) + moments[1] * diff_prob(whole[0], 4, 1);
    dipole[1] = moments[2] * diff_prob(whole[1], 4, 2) + moments[3] * diff_prob(whole[1], 4, 3);
  } else {
    dipole[0] = dipole[1] = 1.0;
  }
}

// The event's moment tensor as the command line gives it (main.cpp:485-560) and as the source sees
// it (model.cpp:431-433: rotated to the local north-east-down frame at the event).
//   kind 0: SDR strike, dip, rake [degrees], iso fraction, moment (Tensor::SDR, tensors.hpp:242-280)
//   kind 1: USGS rr, tt, pp, rt, rp, tp (tensors.hpp:147-155); "EQ" = 0,-1,1,0,0,0, "EXPL" = 1,1,1,0,0,0
// map / radE / event_loc: the coordinate system (ecs.hpp:242-257) and the event's model-space location.
// out: xx, yy, zz, xy, xz, yz of the rotated tensor; out_sym: max |Mij - Mji| (0 for a symmetric input).
void r3d_oracle_moment_tensor(int kind, const double p[6], int map, double radE, const double event_loc[3],
                              double out[6], double* out_asym) {
  M3 t = kind == 0 ? sdr(p[0], p[1], p[2], p[3], p[4] == 0 ? 1.0 : p[4]) : usgs(p[0], p[1], p[2], p[3], p[4], p[5]);
  const Ecs ecs{map, radE};
  t = transform(t, ecs.xyz_to_ned(mk(event_loc[0], event_loc[1], event_loc[2])));
  out[0] = t.xx, out[1] = t.yy, out[2] = t.zz, out[3] = t.xy, out[4] = t.xz, out[5] = t.yz;
  if (out_asym)
    *out_asym = std::fmax(std::fabs(t.xy - t.yx), std::fmax(std::fabs(t.xz - t.zx), std::fabs(t.yz - t.zy)));
}

// ShearDislocation::ShearDislocation (events.cpp:42-107): the integrated P, SH, SV radiation patterns
// of moment tensor mt = xx, yy, zz, xy, xz, yz over the take-off set (Aki & Richards box 9.10 forms as
// the reference writes them), and the integrated whole-space weights mWholeProbs[0].
void r3d_oracle_source(const double mt[6], const double* toa, uint64_t n, double* cdf[3], double whole[3]) {
  const double mxx = mt[0], myy = mt[1], mzz = mt[2], mxy = mt[3], mxz = mt[4], myz = mt[5];
  for (uint64_t ctr = 0; ctr < n; ctr++) {
    const double theta = toa[2 * ctr], az = toa[2 * ctr + 1];
    double prob;
    prob = std::pow(std::sin(theta), 2) *
               (mxx * std::pow(std::cos(az), 2) + mxy * std::sin(2 * az) + myy * std::pow(std::sin(az), 2) - mzz) +
           2 * std::sin(theta) * std::cos(theta) * (mxz * std::cos(az) + myz * std::sin(az)) + mzz;
    cdf[0][ctr] = prob * prob;
    prob = std::sin(theta) * (0.5 * std::sin(2 * az) * (myy - mxx) + std::cos(2 * az) * mxy) +
           std::cos(theta) * (std::cos(az) * myz - std::sin(az) * mxz);
    cdf[1][ctr] = prob * prob;
    prob = std::sin(theta) * std::cos(theta) *
               (mxx * std::pow(std::cos(az), 2) + mxy * std::sin(2 * az) + myy * std::pow(std::sin(az), 2) - mzz) +
           (1.0 - 2 * std::pow(std::sin(theta), 2)) * (mxz * std::cos(az) + myz * std::sin(az));
    cdf[2][ctr] = prob * prob;
  }
  for (int c = 0; c < 3; c++)
    for (uint64_t i = 1; i < n; i++) cdf[c][i] += cdf[c][i - 1];
  whole[0] = cdf[0][n - 1];
  whole[1] = whole[0] + cdf[1][n - 1];
  whole[2] = whole[1] + cdf[2][n - 1];
}

// Seismometer::Seismometer (dataout.cpp:42-71) with the X1 axis Model::Model hands it
// (model.cpp:486-491: east for ENZ, radial from the event for RTZ): axes[0..2] = X1, X2, X3 (X3 up,
// X2 = X3 x X1 normalised or north, X1 = X2 x X3), area[t] = pi (r_out^2 - r_in^2).
void r3d_oracle_seismometer(int map, double radE, const double event_loc[3], const double loc[3], int rtz,
                            const double r_in[2], const double r_out[2], double axes[3][3], double area[2]) {
  const Ecs ecs{map, radE};
  const V where = mk(loc[0], loc[1], loc[2]);
  const V x1_given = rtz ? ecs.radial(mk(event_loc[0], event_loc[1], event_loc[2]), where) : ecs.east(where);
  const V x3 = ecs.up(where);
  V x2 = cross(x3, x1_given);
  x2 = is_squared_zero(x2) ? ecs.north(where) : unit(x2);
  const V x1 = cross(x2, x3);
  const V ax[3] = {x1, x2, x3};
  for (int k = 0; k < 3; k++) axes[k][0] = ax[k].x, axes[k][1] = ax[k].y, axes[k][2] = ax[k].z;
  for (int t = 0; t < 2; t++) area[t] = (r_out[t] * r_out[t] - r_in[t] * r_in[t]) * PI;
}


}  // extern "C"

// ============================================================ cell builders ==
// Model::BuildCellArray_Cylinder / _SphericalShells / _WCGTetra (model.cpp:647-934, :1017-1228) with
// the cell and face constructors they call (media.cpp:130-156, :353-395, :578-626;
// media_cellface.cpp:46-70, :176-214; geom_r4.cpp:15-66) and the scatterer sharing rule
// (scatterers.cpp:45-91), from the grid's nodes as Model sees them: model-space location, both
// attribute sets after the coordinate system's conversion, the discontinuity flag.
namespace {

struct NodeSide {   // GridData as the builders read it
  double vp, vs, rho, qp, qs, nu, eps, a, kappa;
};
struct GridNodeIn {   // (the C layout of r3d_oracle_node, below)
  double loc[3];
  double radius;      // GetRawLoc().Radius(ECS), spherical grids
  double side[2][9];  // [GN_ABOVE], [GN_BELOW]
  int32_t n_sets;     // 2: the node is a first-order discontinuity
  int32_t pad_;
};
inline V loc_of(const GridNodeIn& n) { return mk(n.loc[0], n.loc[1], n.loc[2]); }
inline NodeSide side_of(const GridNodeIn& n, int s) {
  const double* d = n.side[s];
  return NodeSide{d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], d[8]};
}
enum { ABOVE = 0, BELOW = 1 };

struct Builder {
  std::vector<r3d_cell> cells;
  std::vector<Het> scat;     // in creation order
  double omega;
  bool one_dummy;            // --overridemfp and --nodeflect together (scatterers.cpp:48-52)

  // Scatterer::GetScattererMatchingParams, scatterers.cpp:45-91; ScatterParams(V, HS), scatparams.hpp:124-134
  int scatterer_for(const NodeSide& v, const NodeSide& hs) {
    Het par{hs.nu, hs.eps, hs.a, hs.kappa, omega / v.vs, v.vp / v.vs};
    if (one_dummy) par = Het{1.0, 0.0, 1.0, 1.0, 1.0, 1.0};   // HSneak(1.0, 0.0, 1.0, 1.0), el 1, gam0 1
    for (size_t i = 0; i < scat.size(); i++) {
      const Het& o = scat[i];   // CompareRoughly, scatparams.cpp:38-49: sum of squared differences <= 0
      const double dnu = o.nu - par.nu, deps = o.eps - par.eps, da = o.a - par.a, dk = o.kappa - par.kappa;
      const double del = o.el - par.el, dg = o.gam0 - par.gam0;
      if (dnu * dnu + deps * deps + da * da + dk * dk + del * del + dg * dg <= 0) return (int)i;
    }
    scat.push_back(par);
    return (int)scat.size() - 1;
  }
};

r3d_face no_face() {
  r3d_face f;
  std::memset(&f, 0, sizeof f);
  f.neighbor = -1;
  return f;
}
r3d_cell no_cell(int n_faces) {
  r3d_cell c;
  std::memset(&c, 0, sizeof c);
  c.scatterer = -1, c.n_faces = n_faces;
  for (auto& f : c.faces) f = no_face();
  return c;
}
// PlaneFace(N1, N2, N3), media_cellface.cpp:176-190: normal (N2 - N1) x (N3 - N1), normalised; point N1
r3d_face plane3(V n1, V n2, V n3) {
  r3d_face f = no_face();
  V nrm = cross(n2 - n1, n3 - n1);
  const double m = mag(nrm);
  nrm = mk(nrm.x / m, nrm.y / m, nrm.z / m);   // XYZ::Normalize
  f.normal[0] = nrm.x, f.normal[1] = nrm.y, f.normal[2] = nrm.z;
  f.point[0] = n1.x, f.point[1] = n1.y, f.point[2] = n1.z;
  return f;
}
// PlaneFace(N1, N2, N3, N4), :196-214: the same, turned away from the excluded node N4
r3d_face plane4(V n1, V n2, V n3, V n4) {
  r3d_face f = plane3(n1, n2, n3);
  const V nrm = mk(f.normal[0], f.normal[1], f.normal[2]);
  if (dot(nrm, n4 - n1) > 0) f.normal[0] = -nrm.x, f.normal[1] = -nrm.y, f.normal[2] = -nrm.z;
  return f;
}
// CellFace::LinkTo(other, disc) / LinkTo(other), media_cellface.cpp:46-70
void link(std::vector<r3d_cell>& cells, int ca, int fa, int cb, int fb, bool disc) {
  r3d_face &A = cells[ca].faces[fa], &B = cells[cb].faces[fb];
  A.neighbor = cb, B.neighbor = ca;
  A.flags |= R3D_FACE_ADJOIN, B.flags |= R3D_FACE_ADJOIN;
  A.flags = disc ? (A.flags | R3D_FACE_DISCON) : (A.flags & ~(uint32_t)R3D_FACE_DISCON);
  B.flags = disc ? (B.flags | R3D_FACE_DISCON) : (B.flags & ~(uint32_t)R3D_FACE_DISCON);
}
void link_keep(std::vector<r3d_cell>& cells, int ca, int fa, int cb, int fb) {
  const bool disc = ((cells[ca].faces[fa].flags | cells[cb].faces[fb].flags) & R3D_FACE_DISCON) != 0;
  link(cells, ca, fa, cb, fb, disc);
}

// R4::Matrix::SolveAXB, geom_r4.cpp:15-66: Gauss-Jordan on the augmented rows, with the reference's own
// pivot search (it compares every candidate with ROW i, not with the best so far, and so ends on the
// LAST row that beats row i), rows scaled by 1 / pivot, elimination below, then back-substitution.
void solve_axb(const double A[4][4], const double b[4], double x[4]) {
  double row[4][5];
  for (int r = 0; r < 4; r++) {
    for (int c = 0; c < 4; c++) row[r][c] = A[r][c];
    row[r][4] = b[r];
  }
  for (int i = 0; i < 4; i++) {
    int largest = i;
    for (int k = i; k < 4; k++)
      if (std::abs(row[k][i]) > std::abs(row[i][i])) largest = k;
    if (largest != i)
      for (int c = 0; c < 5; c++) std::swap(row[largest][c], row[i][c]);
    if (row[i][i] != 0) {
      const double s = 1 / row[i][i];
      for (int c = 0; c < 5; c++) row[i][c] *= s;
      for (int j = 1; j < 4 - i; j++) {
        const double f = row[i + j][i];
        for (int c = 0; c < 5; c++) row[i + j][c] -= row[i][c] * f;
      }
    }
  }
  for (int j = 3; j >= 1; j--)
    for (int i = 0; i <= j - 1; i++) {
      const double f = row[i][j];
      for (int c = 0; c < 5; c++) row[i][c] -= row[j][c] * f;
    }
  for (int r = 0; r < 4; r++) x[r] = row[r][4];
}

// Tetra::Tetra, media.cpp:353-395
r3d_cell tetra(const V n[4], const NodeSide d[4]) {
  r3d_cell c = no_cell(4);
  c.faces[0] = plane4(n[1], n[2], n[3], n[0]);
  c.faces[1] = plane4(n[2], n[3], n[0], n[1]);
  c.faces[2] = plane4(n[3], n[0], n[1], n[2]);
  c.faces[3] = plane4(n[0], n[1], n[2], n[3]);
  double A[4][4], col[3][4], x[4];
  for (int r = 0; r < 4; r++) {
    A[r][0] = n[r].x, A[r][1] = n[r].y, A[r][2] = n[r].z, A[r][3] = 1;
    col[0][r] = d[r].vp, col[1][r] = d[r].vs, col[2][r] = d[r].rho;
  }
  for (int t = 0; t < 2; t++) {
    solve_axb(A, col[t], x);
    for (int k = 0; k < 3; k++) c.vel_grad[t][k] = x[k];
    c.vel_c[t] = x[3];
  }
  solve_axb(A, col[2], x);
  for (int k = 0; k < 3; k++) c.rho_grad[k] = x[k];
  c.rho_c = x[3];
  c.q[0] = (d[0].qp + d[1].qp + d[2].qp + d[3].qp) / 4;
  c.q[1] = (d[0].qs + d[1].qs + d[2].qs + d[3].qs) / 4;
  return c;
}

}  // namespace

extern "C" {

typedef struct r3d_oracle_node {
  double  loc[3];       /* GridNode::Loc(): model space                                        */
  double  radius;       /* GetRawLoc().Radius(ECS) (spherical grids; else unused)              */
  double  side[2][9];   /* Data(GN_ABOVE), Data(GN_BELOW): vp vs rho qp qs nu eps a kappa      */
  int32_t n_sets;       /* attribute sets given: 2 = discontinuous node                        */
  int32_t pad_;
} r3d_oracle_node;

// kind: R3D_CELL_*; nodes in the grid's own order (k slowest, then j, then i: grid.hpp flat index);
// cells_out has room for cells_cap cells, het_out for het_cap scatterers of six doubles (nu, eps, a,
// kappa, el, gam0).  Returns the number of cells (negative on a size error); *n_scat the scatterers.
int r3d_oracle_build_cells(int kind, int ni, int nj, int nk, const r3d_oracle_node* nodes_c, double frequency,
                           double cylinder_range, int one_dummy_scatterer, r3d_cell* cells_out, int cells_cap,
                           double* het_out, int het_cap, int* n_scat) {
  const GridNodeIn* nodes = reinterpret_cast<const GridNodeIn*>(nodes_c);
  static_assert(sizeof(GridNodeIn) == sizeof(r3d_oracle_node), "one layout");
  auto node = [&](int i, int j, int k) -> const GridNodeIn& { return nodes[(size_t)k * nj * ni + (size_t)j * ni + i]; };
  Builder B;
  B.omega = 2.0 * frequency * PI;   // ScatterParams::SetFrequencyHertz, scatparams.hpp
  B.one_dummy = one_dummy_scatterer != 0;
  if (kind == R3D_CELL_CYLINDER) {   // model.cpp:647-724 + RCUCylinder, media.cpp:130-156
    const int nc = nk - 1;
    for (int k = 0; k < nc; k++) {
      const NodeSide top = side_of(node(0, 0, k), BELOW);
      r3d_cell c = no_cell(3);
      c.faces[0] = plane3(loc_of(node(0, 0, k)), loc_of(node(1, 0, k)), loc_of(node(2, 0, k)));               // up
      c.faces[1] = plane3(loc_of(node(0, 0, k + 1)), loc_of(node(2, 0, k + 1)), loc_of(node(1, 0, k + 1)));   // down
      c.faces[2].radius = cylinder_range;   // cmLossFace: shared, never linked (media.cpp:124-125)
      c.vel_c[0] = top.vp, c.vel_c[1] = top.vs;
      c.rho_c = top.rho;
      c.q[0] = top.qp, c.q[1] = top.qs;
      c.scatterer = B.scatterer_for(top, top);
      B.cells.push_back(c);
    }
    for (int k = 1; k < nc; k++) link(B.cells, k - 1, 1, k, 0, node(0, 0, k).n_sets == 2);
    B.cells[0].faces[0].flags |= R3D_FACE_COLLECT | R3D_FACE_REFLECT;
  } else if (kind == R3D_CELL_SPHERESHELL) {   // model.cpp:732-795 + SphereShell, media.cpp:578-626
    const int nc = nk - 1;
    for (int k = 0; k < nc; k++) {
      const NodeSide top = side_of(node(0, 0, k), BELOW), bot = side_of(node(0, 0, k + 1), ABOVE);
      const double RadTop = node(0, 0, k).radius, RadBot = node(0, 0, k + 1).radius;
      r3d_cell c = no_cell(2);
      c.faces[0].radius = RadTop;    // SphereFace(RadTop, F_TOP): outward
      c.faces[1].radius = -RadBot;   // SphereFace(RadBot, F_BOTTOM): inward (signed radius, include/r3d.h)
      const double denom = RadTop * RadTop - RadBot * RadBot;
      c.vel_a[0] = (top.vp - bot.vp) / denom;
      c.vel_a[1] = (top.vs - bot.vs) / denom;
      c.rho_a = (top.rho - bot.rho) / denom;
      c.vel_c[0] = top.vp - (c.vel_a[0] * RadTop * RadTop);
      c.vel_c[1] = top.vs - (c.vel_a[1] * RadTop * RadTop);
      c.rho_c = top.rho - (c.rho_a * RadTop * RadTop);
      c.zero_rad2[0] = -c.vel_c[0] / c.vel_a[0];
      c.zero_rad2[1] = -c.vel_c[1] / c.vel_a[1];
      c.q[0] = top.qp, c.q[1] = top.qs;
      c.scatterer = B.scatterer_for(top, top);
      B.cells.push_back(c);
    }
    for (int k = 1; k < nc; k++) link(B.cells, k - 1, 1, k, 0, node(0, 0, k).n_sets == 2);
    B.cells[0].faces[0].flags |= R3D_FACE_COLLECT | R3D_FACE_REFLECT;
  } else {   // model.cpp:880-934, WCGBuildBasicPattern :1017-1148, WCGLinkBlocksForward :1181-1228
    const int nI = ni - 1, nJ = nj - 1, nK = nk - 1;
    auto base = [&](int i, int j, int k) { return (i * (nK * nJ) + j * nK + k) * 5; };
    enum { FA = 0, FB = 1, FC = 2, FD = 3 };
    auto link_blocks = [&](int block, int adjacent, int faceid, bool mirror) {
      int c0, c1, c2, c3;
      if (faceid == FB) c0 = block + 2, c1 = adjacent + 1, c2 = block + 3, c3 = adjacent + 4;
      else if (faceid == FC) c0 = block + 4, c1 = adjacent + 1, c2 = block + 3, c3 = adjacent + 2;
      else {
        if (mirror) block += 1, adjacent += 1;
        c0 = block + 1, c1 = adjacent + 1, c2 = block + 3, c3 = adjacent + 3;
      }
      link_keep(B.cells, c0, faceid, c1, faceid);
      link_keep(B.cells, c2, faceid, c3, faceid);
    };
    for (int i = 0; i < nI; i++)
      for (int j = 0; j < nJ; j++)
        for (int k = 0; k < nK; k++) {
          const GridNodeIn* N[8] = {&node(i, j, k),     &node(i, j, k + 1),     &node(i, j + 1, k),     &node(i, j + 1, k + 1),
                                    &node(i + 1, j, k), &node(i + 1, j, k + 1), &node(i + 1, j + 1, k), &node(i + 1, j + 1, k + 1)};
          const bool mirror = (((i + j + k) % 2) == 1), surface = (k == 0);
          NodeSide D[8];
          V L[8];
          bool dis[8];
          for (int q = 0; q < 8; q++) {
            D[q] = side_of(*N[q], (q % 2 == 0) ? BELOW : ABOVE);   // even corners are block tops, odd ones bottoms
            L[q] = loc_of(*N[q]);
            dis[q] = N[q]->n_sets == 2;
          }
          auto T = [&](int a, int b, int c, int d) {
            const V n[4] = {L[a], L[b], L[c], L[d]};
            const NodeSide dd[4] = {D[a], D[b], D[c], D[d]};
            return tetra(n, dd);
          };
          r3d_cell t[5];
          int vel_from[5];
          if (!mirror) {
            t[0] = T(6, 5, 0, 3), t[1] = T(1, 3, 5, 0), t[2] = T(2, 0, 6, 3), t[3] = T(7, 5, 3, 6), t[4] = T(4, 6, 0, 5);
            const int from[5] = {0, 0, 2, 6, 4};
            std::memcpy(vel_from, from, sizeof from);
          } else {
            t[0] = T(7, 4, 1, 2), t[1] = T(0, 2, 4, 1), t[2] = T(3, 1, 7, 2), t[3] = T(6, 4, 2, 7), t[4] = T(5, 7, 1, 4);
            const int from[5] = {4, 0, 2, 6, 4};
            std::memcpy(vel_from, from, sizeof from);
          }
          for (int q = 0; q < 5; q++) t[q].scatterer = B.scatterer_for(D[vel_from[q]], D[0]);   // HetSpec always from node [0]
          auto set_dis = [&](int cell, bool v) {
            if (v) t[cell].faces[FD].flags |= R3D_FACE_DISCON;
          };
          if (!mirror) {
            set_dis(4, dis[0] || dis[4] || dis[6]), set_dis(2, dis[0] || dis[2] || dis[6]);
            set_dis(1, dis[1] || dis[5] || dis[3]), set_dis(3, dis[5] || dis[7] || dis[3]);
          } else {
            set_dis(1, dis[0] || dis[4] || dis[2]), set_dis(3, dis[4] || dis[2] || dis[6]);
            set_dis(4, dis[1] || dis[5] || dis[7]), set_dis(2, dis[1] || dis[7] || dis[3]);
          }
          const int b0 = (int)B.cells.size();
          for (int q = 0; q < 5; q++) B.cells.push_back(t[q]);
          link(B.cells, b0 + 0, FA, b0 + 1, FA, false);
          link(B.cells, b0 + 0, FB, b0 + 2, FA, false);
          link(B.cells, b0 + 0, FC, b0 + 3, FA, false);
          link(B.cells, b0 + 0, FD, b0 + 4, FA, false);
          if (surface) {
            const int s1 = mirror ? 1 : 2, s2 = mirror ? 3 : 4;
            B.cells[b0 + s1].faces[FD].flags |= R3D_FACE_REFLECT | R3D_FACE_COLLECT;
            B.cells[b0 + s2].faces[FD].flags |= R3D_FACE_REFLECT | R3D_FACE_COLLECT;
          }
          const int block = base(i, j, k);
          const bool adjmirror = !mirror;
          if (i > 0) link_blocks(base(i - 1, j, k), block, FC, adjmirror);
          if (j > 0) link_blocks(base(i, j - 1, k), block, FB, adjmirror);
          if (k > 0) link_blocks(base(i, j, k - 1), block, FD, adjmirror);
        }
  }
  if ((int)B.cells.size() > cells_cap || (int)B.scat.size() > het_cap) return -1;
  for (size_t i = 0; i < B.cells.size(); i++) cells_out[i] = B.cells[i];
  for (size_t s = 0; s < B.scat.size(); s++) {
    const Het& h = B.scat[s];
    const double v[6] = {h.nu, h.eps, h.a, h.kappa, h.el, h.gam0};
    for (int k = 0; k < 6; k++) het_out[6 * s + k] = v[k];
  }
  if (n_scat) *n_scat = (int)B.scat.size();
  return (int)B.cells.size();
}


// ------------------------------------------------------ grid nodes -> model space --
// What Model sees of a grid node: GridNode::Loc() = ECS.Convert(raw location) (grid.cpp:95-97,
// ecs.cpp:319-372), GetRawLoc().Radius(ECS) (ecs.cpp:61-92), and GridNode::Data(side)
// (grid.cpp:106-124: the one attribute set for both sides, or the first above / the second below)
// through ECS.Convert(location, data) = Earth-flattening of the velocities (ecs.cpp:540-577; density,
// Q and heterogeneity spectrum untouched, :595-677), with Qp / Qs solved from the two Qs given
// (elastic.cpp:10-52).  map: 0 ENU_ORTHO, 1 RAE_ORTHO, 2 RAE_CURVED, 3 RAE_SPHERICAL.
typedef struct r3d_oracle_node_raw {
  double  x[3];
  double  set[2][11];   /* vp vs rho | unknown-Q code (0 Qp, 1 Qs, 2 Qk), Qp, Qs, Qk | nu eps a kappa */
  int32_t n_sets;
  int32_t pad_;
} r3d_oracle_node_raw;

int r3d_oracle_convert_nodes(int map, double radE, int flatten, size_t n, const r3d_oracle_node_raw* raw,
                             r3d_oracle_node* out) {
  const double DtoR_ = PI / 180.0;
  for (size_t i = 0; i < n; i++) {
    const r3d_oracle_node_raw& r = raw[i];
    r3d_oracle_node& o = out[i];
    std::memset(&o, 0, sizeof o);
    // EarthCoords::FlattenDepth, ecs.cpp:540-545
    auto flatten_depth = [&](double z) { return radE * std::log((radE + z) / radE); };
    double X, Y, Z;
    if (map == 0) {
      X = r.x[0], Y = r.x[1], Z = flatten ? flatten_depth(r.x[2]) : r.x[2];
    } else if (map == 1) {
      const double range = r.x[0], phi = DtoR_ * (90.0 - r.x[1]);
      X = range * std::cos(phi), Y = range * std::sin(phi), Z = flatten ? flatten_depth(r.x[2]) : r.x[2];
    } else {
      const double range = r.x[0], theta = range / radE, phi = DtoR_ * (90.0 - r.x[1]);
      const double rr = radE + r.x[2];
      if (rr < 0 || theta > PI) return 1;
      X = rr * std::sin(theta) * std::cos(phi), Y = rr * std::sin(theta) * std::sin(phi);
      Z = (rr * std::cos(theta)) + (map == 2 ? -radE : 0.0);   // + GetEarthCenter().z()
      o.radius = rr;                                            // ExtractRadius
    }
    o.loc[0] = X, o.loc[1] = Y, o.loc[2] = Z;
    o.n_sets = r.n_sets;
    if (r.n_sets == 0) continue;
    for (int side = 0; side < 2; side++) {
      const double* d = r.set[r.n_sets == 1 ? 0 : side];   // grid.cpp:113-120
      double vp = d[0], vs = d[1];
      if (flatten) {                                        // FlattenVelocity, ecs.cpp:564-577
        const double f = radE / (radE + r.x[2]);
        vp = vp * f, vs = vs * f;
      }
      // Elastic::Q::Qp / Qs, elastic.cpp:10-52 (L = (4/3)(beta/alpha)^2)
      double L = vs / vp;
      L *= L;
      L *= (4. / 3.);
      const int unknown = (int)d[3];
      const double mQp = d[4], mQs = d[5], mQk = d[6];
      double qp = mQp, qs = mQs;
      if (unknown == 0) {
        double q = (L == 0.) ? 0 : L / mQs;
        q += (1. - L) / mQk;
        qp = 1. / q;
      } else if (unknown == 1) {
        double q = 1. / mQp;
        q -= (1. - L) / mQk;
        qs = L / q;
      }
      const double v[9] = {vp, vs, d[2], qp, qs, d[7], d[8], d[9], d[10]};
      for (int q = 0; q < 9; q++) o.side[side][q] = v[q];
    }
  }
  return 0;
}

}  // extern "C"
