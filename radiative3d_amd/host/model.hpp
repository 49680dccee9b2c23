// model.hpp -- host-side model builder: turns run parameters + a Grid into
// the flat, immutable tables the transport engine consumes (r3d_model_desc,
// include/r3d.h).
//
// Mirrors the reference's ModelParams / Model pair (model.hpp:59-243,
// :262-418): same parameter names and defaults, same build order (TOA set ->
// statics -> grid -> cells + scatterers -> event source -> seismometers,
// model.cpp:220-501).  What differs is the product: instead of a graph of
// heap objects with virtual dispatch, build() emits POD arrays.  The outer
// N-phonon loop (Model::RunSimulation, model.cpp:602-633) is NOT here: that
// is the path the HIP engine replaces.
#ifndef R3DH_MODEL_HPP_
#define R3DH_MODEL_HPP_

#include <iosfwd>
#include <memory>
#include <vector>

#include "../../include/r3d.h"
#include "grid.hpp"
#include "scatparams.hpp"
#include "tensors.hpp"

class ModelParams {
 public:
  enum grid_source_e { GRID_UNSPEC, GRID_FROMFILE, GRID_COMPILED };
  enum axes_scheme_e { AX_ENU, AX_RTZ };

  struct SeisRequest {
    EarthCoords::Generic Location;
    axes_scheme_e Orientation;
    Real GatherRadiusInner[2];
    Real GatherRadiusOuter[2];
    bool RadiiUnitsAreWavelengths;
  };

  // Defaults: reference model.hpp:176-188.
  Tensor::Tensor EventSourceMT = Tensor::USGS(0, -1, 1, 0, 0, 0);
  EarthCoords::Generic EventSourceLoc{0, 0, -1};
  int TOA_Degree = 7;
  long NumPhonons = 10;
  Real PhononTTL = 60.0;
  Real Frequency = 4.0;
  Real TimeBinsPerCycle = 0.0;
  Real TimeBinSize = 2.0;
  grid_source_e GridSource = GRID_UNSPEC;
  Real CylinderRange = 600.0;
  int CompiledSelector = 0;
  std::vector<Real> CompiledArgs;

  // Switches the reference pokes straight into class statics / the ECS
  // singleton while parsing (main.cpp:462,465,569,576); carried here so a
  // process can build more than one model.
  bool Flatten = false;
  Real EarthRadius = 6371.0;
  bool OverrideMFP = false;
  Real MFPOverride[2] = {0, 0};
  bool NoDeflect = false;
  bool OcsRaw = false;
  // --device-tables (not a reference option): leave the scattering tables to the engine,
  // which evaluates them in HBM (include/r3d.h r3d_scatterer, build-on-device form).
  bool DeviceTables = false;
  // --host-tables: keep them on the host even where the device build is the default (./main runs)
  bool HostTables = false;

  void AddSeismometerByWavelength(EarthCoords::Generic loc, axes_scheme_e ax, Real radius_wl);
  void AddSeismometerFixedRadius(EarthCoords::Generic loc, axes_scheme_e ax, Real radius);
  void AddSeismometerRing(EarthCoords::Generic loc, axes_scheme_e ax, Real r_in, Real r_out);

  Real GetBinSize() const {
    return TimeBinsPerCycle == 0 ? TimeBinSize : 1.0 / (Frequency * TimeBinsPerCycle);
  }
  const std::vector<SeisRequest>& SeisRequests() const { return mSReqList; }

 private:
  std::vector<SeisRequest> mSReqList;
};

// Everything the post-build summary prints about one scatterer
// (reference scatterers.cpp:420-478).
struct ScattererInfo {
  Real nu, eps, a, kappa, el, gam0;
  Real mfp[2];
  Real dipole[2];
};

class Model {
 public:
  explicit Model(const ModelParams& par, std::ostream* log = nullptr);
  ~Model();
  Model(const Model&) = delete;
  Model& operator=(const Model&) = delete;

  const r3d_model_desc& Desc() const { return mDesc; }
  const Grid& GetGridRef() const { return mGrid; }
  long NumPhonons() const { return mNumPhonons; }
  const std::vector<ScattererInfo>& Scatterers() const { return mScatInfo; }
  // With --device-tables the mean free paths and dipole moments are the engine's output
  // (r3d_engine_scatterer_stats): record them for the scatterer dump.
  void SetScattererStats(int s, const double mfp[2], const double dipole[2]) {
    for (int t = 0; t < 2; t++) mScatInfo.at(s).mfp[t] = mfp[t], mScatInfo.at(s).dipole[t] = dipole[t];
    for (int t = 0; t < 2; t++) mScatDesc.at(s).mfp[t] = mfp[t];
  }
  bool DeviceTables() const { return mDeviceTables; }
  const std::vector<std::string>& SeisAxesDesc() const { return mSeisAxes; }   // "RTZ" / "ENZ" per seismometer
  const Tensor::Tensor& EventMT() const { return mEventMTUser; }               // as given by the user (NED)
  R3::XYZ EventLoc() const { return mEventLoc; }
  bool OverridesMFP() const { return mOverrideMFP; }
  bool NoDeflect() const { return mNoDeflect; }
  const std::vector<S2::ThetaPhi>& TOA() const { return mTOA; }
  // The coordinate system the model was built in (the global ECS moves on with the next model):
  // EarthCoords::earthcoords_e code and Earth radius.
  int MapCode() const { return mMapCode; }
  Real EarthRadius() const { return mRadE; }

  // Locators (reference model.cpp:521-551, :562-594) over the flat tables.
  int FindCellContainingPoint(const R3::XYZ& loc) const;
  R3::XYZ FindSurface(R3::XYZ loc) const;

 private:
  struct ScatStore;  // owns one scatterer's CDF arrays

  void BuildCellArray_Cylinder(Real range);
  void BuildCellArray_SphericalShells();
  void BuildCellArray_WCGTetra();
  int ScattererFor(const ScatterParams& par);
  void BuildSource(const ModelParams& par);
  void BuildSeismometers(const ModelParams& par);

  Grid mGrid;
  long mNumPhonons = 0;
  size_t mNumTOA = 0;   // 20 * 4^degree (the set itself is only built when the tables are made on the host)
  std::vector<S2::ThetaPhi> mTOA;
  std::vector<double> mTOAFlat;
  std::vector<r3d_cell> mCells;
  std::vector<int> mSurfaceFaces;  // cell*4+face of registered surface faces
  std::vector<std::unique_ptr<ScatStore>> mScatStore;
  std::vector<ScatterParams> mScatParams;
  std::vector<ScattererInfo> mScatInfo;
  std::vector<r3d_scatterer> mScatDesc;
  std::vector<double> mSrcCdf[3];
  std::vector<r3d_seismometer> mSeis;
  std::vector<std::string> mSeisAxes;
  Tensor::Tensor mEventMTUser;
  R3::XYZ mEventLoc;
  bool mOverrideMFP = false, mNoDeflect = false, mDeviceTables = false;
  Real mMFPOverride[2] = {0, 0};
  int mMapCode = 0;
  Real mRadE = 6371.0;
  r3d_model_desc mDesc{};
};

#endif
