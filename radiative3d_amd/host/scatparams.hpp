// scatparams.hpp -- Sato & Fehler von-Karman scattering coefficients.
//
// ScatterParams keeps the reference's public surface (scatparams.hpp:70-250:
// constructors, Get*, CompareRoughly, GSATO, XSATO, PSATO,
// SetFrequencyHertz).  GSATO is Sato & Fehler eq. 4.52, XSATO eq. 4.50 and
// PSATO the von-Karman PSDF (reference scatparams.cpp:75-194), including the
// "< 1e-30 -> 0" clamps inherited from PSPhonon.
#ifndef R3DH_SCATPARAMS_HPP_
#define R3DH_SCATPARAMS_HPP_

#include "elastic.hpp"
#include "geom.hpp"

class ScatterParams {
  Real nu, eps, a, kappa;
  Real el;    // S wavenumber omega / Vs
  Real gam0;  // Vp / Vs
  Real psdf_numer_;  // 8 pi^1.5 eps^2 a^3 Gamma(k+1.5)/Gamma(k), hoisted out of PSATO

  static Real& omega() {
    static Real w = 1.0;
    return w;
  }
  static bool& omega_known() {
    static bool k = false;
    return k;
  }
  void hoist() {
    psdf_numer_ = (8. * std::pow(Geometry::Pi, 1.5) * eps * eps * a * a * a) *
                  std::tgamma(kappa + 1.5) / std::tgamma(kappa);
  }

 public:
  static void SetFrequencyHertz(Real f) {
    omega() = 2.0 * f * Geometry::Pi;
    omega_known() = true;
  }

  ScatterParams(Elastic::Velocity v, Elastic::HetSpec hs)
      : nu(hs.nu()), eps(hs.eps()), a(hs.a()), kappa(hs.kappa()),
        el(omega() / v.Vs()), gam0(v.Vp() / v.Vs()) {
    if (!omega_known())
      throw Invalid("Can't construct ScatterParams before frequency is known.");
    hoist();
  }
  ScatterParams(Elastic::HetSpec hs, Real el_, Real gam0_)
      : nu(hs.nu()), eps(hs.eps()), a(hs.a()), kappa(hs.kappa()), el(el_), gam0(gam0_) {
    hoist();
  }

  Real GetNu() const { return nu; }
  Real GetEps() const { return eps; }
  Real GetA() const { return a; }
  Real GetKappa() const { return kappa; }
  Real GetL() const { return el; }
  Real GetPsdfNumer() const { return psdf_numer_; }
  Real GetGam0() const { return gam0; }

  // Sum of squared parameter differences; scatterers are shared between
  // cells only when this is <= 0, i.e. on exact equality
  // (reference scatparams.cpp:38-49, scatterers.cpp:60-62).
  Real CompareRoughly(const ScatterParams& o) const {
    Real d[6] = {o.nu - nu, o.eps - eps, o.a - a, o.kappa - kappa, o.el - el, o.gam0 - gam0};
    Real s = 0;
    for (Real v : d) s += v * v;
    return s;
  }

  Real PSATO(Real m) const { return psdf_numer_ / std::pow(1. + a * a * m * m, kappa + 1.5); }

  // psi = deflection colatitude (toa.Theta), zeta = azimuth (toa.Phi).
  void XSATO(S2::ThetaPhi toa, Real& xpp, Real& xps, Real& xsp, Real& xss_psi,
             Real& xss_zeta) const {
    const Real g2 = gam0 * gam0;
    const Real cpsi = std::cos(toa.Theta()), c2psi = std::cos(2. * toa.Theta());
    const Real spsi = std::sin(toa.Theta());
    const Real czeta = std::cos(toa.Phi()), szeta = std::sin(toa.Phi());
    const Real spsi2 = spsi * spsi;
    xpp = (1. / g2) * (nu * (-1. + cpsi + (2. / g2) * spsi2) - 2. + (4. / g2) * spsi2);
    xps = -spsi * (nu * (1. - (2. / gam0) * cpsi) - (4. / gam0) * cpsi);
    xsp = (1. / g2) * spsi * czeta * (nu * (1. - (2. / gam0) * cpsi) - (4. / gam0) * cpsi);
    xss_psi = czeta * (nu * (cpsi - c2psi) - 2. * c2psi);
    xss_zeta = szeta * (nu * (cpsi - 1.) + 2. * cpsi);
  }

  void GSATO(S2::ThetaPhi toa, Real& gpp, Real& gps, Real& gsp, Real& gss, Real& spol) const {
    const Real pi4 = 4. * Geometry::Pi;
    const Real el4 = std::pow(el, 4);
    const Real g2 = std::pow(gam0, 2);
    const Real psi = toa.Theta();
    Real xpp, xps, xsp, xss_psi, xss_zeta;
    XSATO(toa, xpp, xps, xsp, xss_psi, xss_zeta);
    Real m = (2. * el / gam0) * std::sin(psi / 2.);
    gpp = (el4 / pi4) * (xpp * xpp) * PSATO(m);
    if (gpp < 1.e-30) gpp = 0.;
    m = (el / gam0) * std::sqrt(1. + g2 - 2. * gam0 * std::cos(psi));
    Real pm = PSATO(m);
    gps = (1. / gam0) * (el4 / pi4) * (xps * xps) * pm;
    if (gps < 1.e-30) gps = 0.;
    gsp = gam0 * (el4 / pi4) * (xsp * xsp) * pm;
    if (gsp < 1.e-30) gsp = 0.;
    m = 2. * el * std::sin(psi / 2.);
    gss = (el4 / pi4) * (xss_psi * xss_psi + xss_zeta * xss_zeta) * PSATO(m);
    if (gss < 1.e-30) gss = 0.;
    spol = std::atan2(xss_zeta, xss_psi);
  }
};

#endif
