// r3d_tables_oracle.cpp -- CPU restatement of the reference's TABLE BUILDERS: everything the hot
// path reads that is computed once, before the first phonon -- the take-off-angle set, the
// scatterers' radiation-pattern tables / mean free paths / dipole moments, the event source's
// moment tensor and P / SH / SV patterns, the seismometers' axes and gather areas.
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
