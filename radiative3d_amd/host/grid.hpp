// grid.hpp -- the node lattice a model definition fills in.
//
// Source-compatible with what the reference's user model files use
// (reference grid.hpp:44-403; user.cpp includes only "grid.hpp"):
//   Grid::{SetSize, SetIndexBase, SetMapping, WNode, Node, RelNode,
//          ConstructGridManual, GetModelType, DumpGridToAscii}, the
//   Grid::GC_* enums, GridNode::{SetLocation, AdjustLocation, SetAttributes
//   (both argument orders), ClearAttributes, Loc, GetRawLoc, Data,
//   IsDiscontinuous}, GridData.
// A node given ONE attribute set is continuous; a node given TWO (above,
// below) marks a first-order discontinuity (reference grid.cpp:73-83,106-124).
#ifndef R3DH_GRID_HPP_
#define R3DH_GRID_HPP_

#include <ostream>
#include <string>
#include <vector>

#include "ecs.hpp"

class GridData : public Elastic::HElastic {
 public:
  GridData(Elastic::Velocity v, Real rho, Elastic::Q q, Elastic::HetSpec hs)
      : Elastic::HElastic(v, rho, q, hs) {}
  GridData(const Elastic::HElastic& he) : Elastic::HElastic(he) {}
  GridData()
      : Elastic::HElastic(Elastic::VpVs(0, 0), 0, Elastic::Qinf(), Elastic::HetSpec()) {}
};

class GridNode {
 public:
  enum layers_e { GN_ABOVE, GN_BELOW, GN_NLAY };

  void SetLocation(Real x, Real y, Real z);  // first call wins
  void SetLocation(EarthCoords::Generic g) { SetLocation(g.x1(), g.x2(), g.x3()); }
  void AdjustLocation(Real dx, Real dy, Real dz);
  void SetAttributes(GridData d);
  void SetAttributes(Elastic::Velocity v, Real rho, Elastic::Q q, Elastic::HetSpec hs) {
    SetAttributes(GridData(v, rho, q, hs));
  }
  void SetAttributes(Real rho, Elastic::Velocity v, Elastic::Q q, Elastic::HetSpec hs) {
    SetAttributes(GridData(v, rho, q, hs));
  }
  void ClearAttributes() { mFilled = 0; }

  R3::XYZ Loc() const { return ECS.Convert(mLoc); }  // model-space location
  EarthCoords::Generic GetRawLoc() const { return mLoc; }
  GridData Data(layers_e side) const;  // properties (Earth-flattened if enabled)
  GridData RawData(int slot) const { return mData[slot]; }
  int NumAttributeSets() const { return mFilled; }
  bool IsDiscontinuous() const { return mFilled == 2; }
  void OutputAsAscii(std::ostream& out, const std::string& prefix) const;

 private:
  EarthCoords::Generic mLoc;
  int mFilled = 0;  // attribute sets given so far (0..2)
  GridData mData[2];
};

class Grid {
 public:
  enum model_target_e { MOD_AUTO, MOD_CYLINDER, MOD_TETRAWCG, MOD_SPHERESHELL };
  enum gs_coords_e { GC_ENU, GC_RAE, GC_LLE };
  enum curvature_e { GC_ORTHO, GC_FLATTENED, GC_CURVED, GC_SPHERICAL };

  void SetSize(Count ni, Count nj, Count nk);
  void SetIndexBase(Index base) { mBase = base; }
  void SetMapping(gs_coords_e, curvature_e);
  GridNode& WNode(Index i, Index j, Index k);
  const GridNode& Node(Index i, Index j, Index k) const { return mNodes[flat(i, j, k)]; }
  const GridNode& RelNode(Index i, Index j, Index k, RelIndex ri, RelIndex rj,
                          RelIndex rk) const {
    return mNodes[flat(i + ri, j + rj, k + rk)];
  }
  model_target_e GetModelType() const;
  Count N() const { return mNi * mNj * mNk; }
  Count Ni() const { return mNi; }
  Count Nj() const { return mNj; }
  Count Nk() const { return mNk; }

  // Dispatcher to the compiled-in model definitions (reference user.cpp:59).
  void ConstructGridManual(int Selection, const std::vector<Real>& args);
  void DumpGridToAscii(std::ostream& out) const;
  void DumpGridToAscii() const;

 private:
  size_t flat(Index i, Index j, Index k) const {
    return (size_t)k * (mNj * mNi) + (size_t)j * mNi + i;
  }
  Count mNi = 0, mNj = 0, mNk = 0;
  Index mBase = 0;
  std::vector<GridNode> mNodes;
};

#endif
