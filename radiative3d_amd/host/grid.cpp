// grid.cpp -- see grid.hpp.  Where the reference calls exit(1) on misuse
// (grid.cpp:218-219, 308-309) this library throws Runtime instead: nothing
// below the C ABI may terminate the host process.
#include "grid.hpp"

#include <iomanip>
#include <iostream>

// reference grid.cpp:44-48: a location can be given once; later calls are
// ignored unless the stored triple is still (0,0,0).
void GridNode::SetLocation(Real x, Real y, Real z) {
  if (mLoc.IsNull()) mLoc.SetTriple(x, y, z);
}

void GridNode::AdjustLocation(Real dx, Real dy, Real dz) {
  mLoc.SetTriple(mLoc.x1() + dx, mLoc.x2() + dy, mLoc.x3() + dz);
}

// reference grid.cpp:73-83
void GridNode::SetAttributes(GridData d) {
  if (mFilled == 2)
    throw Runtime("GridNode: SetAttributes: Too many definitions for GridNode.\n");
  mData[mFilled++] = d;
}

// reference grid.cpp:106-124: with a single attribute set both sides see it.
GridData GridNode::Data(layers_e side) const {
  if (mFilled == 0) throw Runtime("GridNode: Data: No attributes set for GridNode.\n");
  const GridData& pick = (mFilled == 1) ? mData[0] : mData[side == GN_ABOVE ? 0 : 1];
  return ECS.Convert(mLoc, pick);
}

// reference grid.cpp:135-190 (column layout of the grid dump)
void GridNode::OutputAsAscii(std::ostream& out, const std::string& prefix) const {
  std::ios_base::fmtflags saved = out.flags();
  out.precision(5);
  out << std::fixed << std::right;
  R3::XYZ iloc = ECS.Convert(mLoc);
  EarthCoords::Generic oloc = ECS.OutConvert(iloc);
  auto put_loc = [&](const char* gap) {
    out << prefix << std::setw(11) << oloc.x1() << " " << std::setw(11) << oloc.x2() << " "
        << std::setw(11) << oloc.x3() << gap;
  };
  if (mFilled == 0) {
    put_loc("  ");
    out << "        ***       ***       ***         ***       *** "
        << "        ***       ***       ***       ***\n";
  }
  for (int s = 0; s < mFilled; s++) {
    GridData od = ECS.OutConvert(iloc, ECS.Convert(mLoc, mData[s]));
    put_loc("    ");
    out << std::setw(9) << od.Vp() << " " << std::setw(9) << od.Vs() << " " << std::setw(9)
        << od.Rho() << "   ";
    out.precision(1);
    out << std::setw(9) << od.Qp() << " " << std::setw(9) << od.Qs() << "   ";
    out.precision(5);
    out << std::setw(9) << od.getHS().nu() << " " << std::setw(9) << od.getHS().eps() << " "
        << std::setw(9) << od.getHS().a() << " " << std::setw(9) << od.getHS().kappa()
        << "\n";
  }
  out.flags(saved);
}

void Grid::SetSize(Count ni, Count nj, Count nk) {
  if (!mNodes.empty()) throw Runtime("Error: Attempt to resize an already-sized grid.");
  mNi = ni, mNj = nj, mNk = nk;
  mNodes.assign((size_t)ni * nj * nk, GridNode());
}

// reference grid.cpp:262-292
void Grid::SetMapping(gs_coords_e coords, curvature_e curve) {
  struct Row {
    gs_coords_e c;
    curvature_e k;
    EarthCoords::earthcoords_e map;
    bool flatten;
  };
  static const Row table[] = {
      {GC_ENU, GC_ORTHO, EarthCoords::ENU_ORTHO, false},
      {GC_ENU, GC_FLATTENED, EarthCoords::ENU_ORTHO, true},
      {GC_RAE, GC_ORTHO, EarthCoords::RAE_ORTHO, false},
      {GC_RAE, GC_FLATTENED, EarthCoords::RAE_ORTHO, true},
      {GC_RAE, GC_CURVED, EarthCoords::RAE_CURVED, false},
      {GC_RAE, GC_SPHERICAL, EarthCoords::RAE_SPHERICAL, false},
  };
  for (const Row& r : table)
    if (r.c == coords && r.k == curve) {
      ECS.SetMapping(r.map);
      ECS.SetEarthFlattening(r.flatten);
      return;
    }
  ECS.SetMapping(EarthCoords::MAP_UNSUPPORTED);
}

GridNode& Grid::WNode(Index i, Index j, Index k) {
  i -= mBase, j -= mBase, k -= mBase;
  if (i >= mNi || j >= mNj || k >= mNk) throw Runtime("ERROR: Grid index out of bounds.");
  return mNodes[flat(i, j, k)];
}

// reference grid.cpp:358-370: 3x1xN -> layered cylinder, 1x1xN -> spherical
// shells, anything else -> warped-cartesian tetrahedra.
Grid::model_target_e Grid::GetModelType() const {
  if (mNi == 3 && mNj == 1 && mNk > 1) return MOD_CYLINDER;
  if (mNi == 1 && mNj == 1 && mNk > 1) return MOD_SPHERESHELL;
  return MOD_TETRAWCG;
}

// reference grid.cpp:376-405
void Grid::DumpGridToAscii(std::ostream& out) const {
  out << "#  R3D_GRID:\n"
      << "#  Line 1:  ni nj nk\n"
      << "#  Line 2:  Index_Base\n"
      << "#  Lines 3 and up describe grid nodes:\n"
      << "#  i j k    x y x    vp vs rho qp qs    nu eps a k\n"
      << "#\n"
      << mNi << " " << mNj << " " << mNk << "\n"
      << 0 << "\n";
  for (Index k = 0; k < mNk; k++)
    for (Index j = 0; j < mNj; j++)
      for (Index i = 0; i < mNi; i++) {
        std::ostringstream pre;
        pre << std::right << std::setw(3) << i << " " << std::setw(3) << j << " " << std::setw(3)
            << k << "  ";
        Node(i, j, k).OutputAsAscii(out, pre.str());
      }
  out << "#  END R3D_GRID\n";
}

void Grid::DumpGridToAscii() const { DumpGridToAscii(std::cout); }
