// tensors.hpp -- moment-tensor constructors in North-East-Down axes.
//
// Same names and argument conventions as the reference (tensors.hpp:126-155
// USGS, :182-204 EulerSDR, :242-295 SDR) so source specifications given on
// the command line (--source=EQ | EXPL | USGS,... | SDR,strike,dip,rake[,iso
// [,moment]]) produce the same tensors.
#ifndef R3DH_TENSORS_HPP_
#define R3DH_TENSORS_HPP_

#include "geom.hpp"

namespace Tensor {

using Tensor = R3::Matrix;

struct Symmetric : Tensor {
  Symmetric(Real xx, Real yy, Real zz, Real xy = 0, Real xz = 0, Real yz = 0)
      : Tensor(xx, xy, xz, xy, yy, yz, xz, yz, zz) {}
};

// Harvard/USGS (r, theta, phi) = (up, south, east) components.
struct USGS : Tensor {
  USGS(Real rr, Real tt, Real pp, Real rt = 0, Real rp = 0, Real tp = 0)
      : Tensor(tt, -tp, rt, -tp, pp, -rp, rt, -rp, rr) {}
};

// Euler rotation used to orient the canonical double couple.
struct EulerSDR : Tensor {
  EulerSDR(Real alpha, Real beta, Real gamma) {
    Real ca = std::cos(alpha), cb = std::cos(beta), cg = std::cos(gamma);
    Real sa = std::sin(alpha), sb = std::sin(beta), sg = std::sin(gamma);
    static_cast<Tensor&>(*this) =
        Tensor(cb * cg - ca * sb * sg, -cb * sg - ca * sb * cg, sa * sb,
               sb * cg + ca * cb * sg, -sb * sg + ca * cb * cg, -sa * cb,
               sa * sg, sa * cg, ca);
  }
};

// Double couple from strike/dip/rake (degrees) with an optional isotropic
// fraction in [-1, 1] (squared-magnitude split) and total moment.
struct SDR : Tensor {
  SDR(Real strike, Real dip, Real rake, Real iso = 0.0, Real moment = 1.0)
      : Tensor(0, 0, -1, 0, 0, 0, -1, 0, 0) {
    if (iso > 1.0 || iso < -1.0) throw std::domain_error("Tensor::SDR: iso not in [-1.0, 1.0]");
    if (moment == 0) throw std::domain_error("Tensor::SDR: moment must be non-zero");
    Transform(EulerSDR(dip * Geometry::DtoR, strike * Geometry::DtoR, -rake * Geometry::DtoR));
    Real iso2 = std::fabs(iso);
    SetSquaredMag(1.0 - iso2);
    Real sgn = (iso >= 0) ? 1 : -1;
    Tensor isotropic = Symmetric(sgn, sgn, sgn);
    isotropic.SetSquaredMag(iso2);
    (*this) += isotropic;
    (*this) *= moment;
  }
  // An "iso" argument outside [-1,1] is taken as an angle in degrees
  // (reference tensors.hpp:282-295).
  static Real IsoFracFromIsoAngle(Real deg) {
    if (deg > 90.0 || deg < -90.0)
      throw std::domain_error("Tensor::SDR: isoangle not in [-90.0, 90.0]");
    if (deg <= -89.999) return -1.0;
    if (deg >= 89.999) return 1.0;
    Real t = std::tan(Geometry::DtoR * deg);
    Real f = t * t / (1 + t * t);
    return deg < 0 ? -f : f;
  }
};

}  // namespace Tensor

#endif
