// elastic.hpp -- elastic-property value types used by model definitions.
//
// Same public names and argument orders as the reference's Elastic namespace
// (reference elastic.hpp:43-322) so model-definition code compiles unchanged:
// Velocity/VpVs, Density, Q (+ Qinf, QpQs, QmQk, QkQm, QpQk), HetSpec/HSneak,
// SElastic, HElastic.  Q solves for the missing member of {Qp, Qs, Qkappa}
// from  1/Qp = L/Qs + (1-L)/Qk,  L = (4/3)(Vs/Vp)^2  (reference
// elastic.cpp:17-65).
#ifndef R3DH_ELASTIC_HPP_
#define R3DH_ELASTIC_HPP_

#include <limits>

#include "typedefs.hpp"

namespace Elastic {

class Velocity {
  Real mVp = 0, mVs = 0;

 protected:
  Velocity(Real vp, Real vs) : mVp(vp), mVs(vs) {}

 public:
  Velocity() = default;
  Real Vp() const { return mVp; }
  Real Vs() const { return mVs; }
};

struct VpVs : Velocity {
  VpVs(Real vp, Real vs) : Velocity(vp, vs) {}
};

class Density {
  Real mRho;

 public:
  Density(Real rho) : mRho(rho) {}
  Real Value() const { return mRho; }
};

class Q {
 protected:
  enum Missing { NEED_QP, NEED_QS, NEED_QK };
  Q(Missing w, Real qp, Real qs, Real qk) : mMissing(w), mQp(qp), mQs(qs), mQk(qk) {}

 private:
  static constexpr Real kInf = std::numeric_limits<Real>::infinity();
  Missing mMissing = NEED_QK;
  Real mQp = kInf, mQs = kInf, mQk = kInf;
  static Real L(Velocity v) {
    Real r = v.Vs() / v.Vp();
    return (4. / 3.) * r * r;
  }

 public:
  Q() = default;
  Real Qp(Velocity v) const {
    if (mMissing != NEED_QP) return mQp;
    Real l = L(v);
    Real inv = (l == 0.) ? 0. : l / mQs;  // Vs->0 wins over Qs->0
    inv += (1. - l) / mQk;
    return 1. / inv;
  }
  Real Qs(Velocity v) const {
    if (mMissing != NEED_QS) return mQs;
    Real l = L(v);
    return l / (1. / mQp - (1. - l) / mQk);
  }
  Real Qk(Velocity v) const {
    if (mMissing != NEED_QK) return mQk;
    Real l = L(v);
    return (1. - l) / (1. / mQp - l / mQs);
  }
  // As given (for tests that restate the conversions): which of the three is the unknown
  // (0 Qp, 1 Qs, 2 Qk) and the three stored values.
  void Stored(int& missing, Real& qp, Real& qs, Real& qk) const {
    missing = (int)mMissing, qp = mQp, qs = mQs, qk = mQk;
  }
};

struct Qinf : Q {
  Qinf()
      : Q(NEED_QK, std::numeric_limits<Real>::infinity(),
          std::numeric_limits<Real>::infinity(), 0) {}
};
struct QpQs : Q {
  QpQs(Real qp, Real qs) : Q(NEED_QK, qp, qs, 0) {}
};
struct QmQk : Q {  // Q_mu (= Q_s) and Q_kappa; AK135 style
  QmQk(Real qs, Real qk = std::numeric_limits<Real>::infinity())
      : Q(NEED_QP, 0, qs, qk) {}
};
struct QkQm : Q {
  QkQm(Real qk, Real qm) : Q(NEED_QP, 0, qm, qk) {}
};
struct QpQk : Q {
  QpQk(Real qp, Real qk = std::numeric_limits<Real>::infinity())
      : Q(NEED_QS, qp, 0, qk) {}
};

class HetSpec {  // von Karman heterogeneity spectrum: nu, eps, a, kappa
  Real mNu = 0.8, mEps = 0.0, mA = 1.0, mKappa = 0.5;

 protected:
  HetSpec(Real nu, Real eps, Real a, Real k) : mNu(nu), mEps(eps), mA(a), mKappa(k) {}

 public:
  HetSpec() = default;
  Real nu() const { return mNu; }
  Real eps() const { return mEps; }
  Real a() const { return mA; }
  Real kappa() const { return mKappa; }
};
struct HSneak : HetSpec {
  HSneak(Real nu, Real eps, Real a, Real k) : HetSpec(nu, eps, a, k) {}
};

class SElastic {
 protected:
  Velocity mV;
  Density mRho;
  Q mQ;

 public:
  SElastic(Velocity v, Density rho, Q q) : mV(v), mRho(rho), mQ(q) {}
  SElastic(Density rho, Velocity v, Q q) : mV(v), mRho(rho), mQ(q) {}
  Velocity getV() const { return mV; }
  Density getDens() const { return mRho; }
  Q getQ() const { return mQ; }
  Real Vp() const { return mV.Vp(); }
  Real Vs() const { return mV.Vs(); }
  Real Rho() const { return mRho.Value(); }
  Real Qp() const { return mQ.Qp(mV); }
  Real Qs() const { return mQ.Qs(mV); }
  Real Qk() const { return mQ.Qk(mV); }
};

class HElastic : public SElastic {
 protected:
  HetSpec mH;

 public:
  HElastic(Velocity v, Density rho, Q q, HetSpec h) : SElastic(v, rho, q), mH(h) {}
  HElastic(Density rho, Velocity v, Q q, HetSpec h) : SElastic(v, rho, q), mH(h) {}
  HetSpec getHS() const { return mH; }
};

}  // namespace Elastic

#endif
