// r3d_tables_build.h -- device-side construction of the scattering tables
// (implementation and references: r3d_tables_build.hip).
#ifndef R3D_TABLES_BUILD_H_
#define R3D_TABLES_BUILD_H_

#include <hip/hip_runtime.h>

#include <cstdint>

#include "r3d_tables.h"

namespace r3d {

// Fill d_cdf[0..3] (GPP, GPS, GSP, GSS; cumulative, n entries each) and d_spol (n) for the
// heterogeneity parameters het = {nu, eps, a, kappa, el, gam0} over the take-off set d_toa
// ((theta, phi) pairs).  totals[c] = d_cdf[c][n-1]; cos_sums[c] = sum cos(theta_k) w_c[k]
// over the raw weights.  Blocks until done.
hipError_t build_scatterer_tables(const double het[6], double psdf_numer, const double* d_toa, uint64_t n,
                                  double* d_cdf[4], double* d_spol, double totals[4], double cos_sums[4],
                                  hipStream_t stream);

// The take-off set of degree `degree` ((theta, phi) pairs, n = 20 * 4^degree) in the host
// builder's order (asynchronous).
hipError_t build_toa_on_device(int degree, uint64_t n, double* d_toa, hipStream_t stream);
// Unit vectors of the take-off directions, theta clamped to [min_theta, max_theta] (asynchronous).
hipError_t build_toa_xyz_on_device(const double* d_toa, uint64_t n, double min_theta, double max_theta,
                                   double* d_xyz, hipStream_t stream);
// Cumulative P / SH / SV radiation patterns of the moment tensor (xx, yy, zz, xy, xz, yz; local
// north-east-down frame) over the take-off set; totals[c] = d_cdf[c][n-1].  Blocks until done.
hipError_t build_source_tables(const double moment[6], const double* d_toa, uint64_t n, double* d_cdf[3],
                               double totals[3], hipStream_t stream);

// (cos, sin) pairs of the n polarisation angles d_spol (asynchronous).
hipError_t build_spol_cs_on_device(const double* d_spol, uint64_t n, double* d_cs, hipStream_t stream);

// The 2^bits guide cells of a cumulative table (r3d_tables.h GuideCell; asynchronous).
hipError_t build_guide_on_device(const double* d_cdf, uint64_t n, uint32_t bits, GuideCell* d_guide,
                                 hipStream_t stream);

}  // namespace r3d
#endif
