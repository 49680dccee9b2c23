// r3d_tables_build.hip -- scattering tables built on the device.
//
// What the reference does per scatterer on the host (Scatterer::Scatterer,
// scatterers.cpp:97-220; ScatterParams::GSATO / XSATO / PSATO,
// scatparams.cpp:75-194): evaluate the Sato & Fehler scattering coefficients
// g_PP, g_PS, g_SP, g_SS (eq. 4.52 with the basic patterns of eq. 4.50 and the
// von-Karman power spectrum) and the S->S polarisation angle at every one of
// the 20 * 4^degree take-off directions, then integrate each into a cumulative
// table.  At degree 9 that is 5.2 M directions x 5 arrays per scatterer -- about
// 2 s of host time for the 7 NSCP scatterers on 16 cores, plus 0.5 s to copy the
// 1.5 GB result into HBM -- against 0.3 s of kernel time for 1e8 histories.  Here
// the tables are produced where they are used:
//   gsato_kernel      one work-item per direction: the four weights + spol, and the
//                     cos(theta)-weighted block sums the dipole diagnostics need
//   scan_blocks       inclusive sum inside 4096-element blocks (16 per work-item)
//   scan_block_sums   one work-item adds up the block totals in order
//   add_offsets       block offsets added back
//   guide_kernel      the search guides of r3d_pack.h build_guide, by bisection
// fp64 throughout; HBM-bound streaming kernels (~0.3 GB of traffic per scatterer).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <vector>

#include "r3d_tables_build.h"

namespace r3d {

namespace {

constexpr int kThreads = 256;
constexpr int kItems = 16;                       // per work-item in the scan
constexpr int kScanBlock = kThreads * kItems;    // elements per block

struct Het {
  double nu, eps, a, kappa, el, gam0, psdf_numer;
};

// ScatterParams::PSATO, scatparams.cpp:160-194 (numerator evaluated once on the host)
__device__ __forceinline__ double psato(const Het& h, double m) {
  return h.psdf_numer / pow(1. + h.a * h.a * m * m, h.kappa + 1.5);
}

// ScatterParams::GSATO + XSATO, scatparams.cpp:75-158.  Weights below 1e-30 are zeroed as
// the reference does.
__global__ __launch_bounds__(kThreads) void gsato_kernel(Het h, const double* __restrict__ toa, uint64_t n,
                                                         double* __restrict__ w0, double* __restrict__ w1,
                                                         double* __restrict__ w2, double* __restrict__ w3,
                                                         double* __restrict__ spol,
                                                         double* __restrict__ block_cos /* [blocks][4] */) {
  const uint64_t k = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
  double g[4] = {0, 0, 0, 0}, cz = 0;
  if (k < n) {
    const double psi = toa[2 * k], zeta = toa[2 * k + 1];
    const double g2 = h.gam0 * h.gam0;
    const double cpsi = cos(psi), c2psi = cos(2. * psi), spsi = sin(psi);
    const double czeta = cos(zeta), szeta = sin(zeta);
    const double spsi2 = spsi * spsi;
    const double xpp = (1. / g2) * (h.nu * (-1. + cpsi + (2. / g2) * spsi2) - 2. + (4. / g2) * spsi2);
    const double xps = -spsi * (h.nu * (1. - (2. / h.gam0) * cpsi) - (4. / h.gam0) * cpsi);
    const double xsp = (1. / g2) * spsi * czeta * (h.nu * (1. - (2. / h.gam0) * cpsi) - (4. / h.gam0) * cpsi);
    const double xss_psi = czeta * (h.nu * (cpsi - c2psi) - 2. * c2psi);
    const double xss_zeta = szeta * (h.nu * (cpsi - 1.) + 2. * cpsi);
    const double pi4 = 4. * 3.14159265358979323846;
    const double el4 = (h.el * h.el) * (h.el * h.el);
    double m = (2. * h.el / h.gam0) * sin(psi / 2.);
    g[0] = (el4 / pi4) * (xpp * xpp) * psato(h, m);
    m = (h.el / h.gam0) * sqrt(1. + g2 - 2. * h.gam0 * cpsi);
    const double pm = psato(h, m);
    g[1] = (1. / h.gam0) * (el4 / pi4) * (xps * xps) * pm;
    g[2] = h.gam0 * (el4 / pi4) * (xsp * xsp) * pm;
    m = 2. * h.el * sin(psi / 2.);
    g[3] = (el4 / pi4) * (xss_psi * xss_psi + xss_zeta * xss_zeta) * psato(h, m);
#pragma unroll
    for (int c = 0; c < 4; c++)
      if (g[c] < 1.e-30) g[c] = 0.;
    w0[k] = g[0], w1[k] = g[1], w2[k] = g[2], w3[k] = g[3];
    spol[k] = atan2(xss_zeta, xss_psi);
    cz = cpsi;
  }
  // block sums of cos(theta) * weight (dipole moments, scatterers.cpp:244-259)
  __shared__ double s_part[kThreads / 64][4];
#pragma unroll
  for (int c = 0; c < 4; c++) {
    double v = cz * g[c];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6][c] = v;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    double v = 0;
    for (int wv = 0; wv < kThreads / 64; wv++) v += s_part[wv][threadIdx.x];
    block_cos[(size_t)blockIdx.x * 4 + threadIdx.x] = v;
  }
}

// In-place inclusive sum inside each block of kScanBlock elements; block totals out.
// grid.y selects the array.
__global__ __launch_bounds__(kThreads) void scan_blocks(double* const* __restrict__ arrays, uint64_t n,
                                                        double* __restrict__ block_sums, uint32_t n_blocks) {
  double* v = arrays[blockIdx.y];
  const uint64_t base = (uint64_t)blockIdx.x * kScanBlock + (uint64_t)threadIdx.x * kItems;
  double x[kItems];
  double run = 0;
#pragma unroll
  for (int i = 0; i < kItems; i++) {
    const uint64_t k = base + i;
    run += (k < n) ? v[k] : 0.0;
    x[i] = run;
  }
  // Exclusive prefix of the work-items' totals, added up IN ORDER by one work-item: the
  // table must be non-decreasing to the last bit (engine and oracle bisect it from different
  // brackets), and with a sequential chain "last element of work-item t" and "base of
  // work-item t+1" are the same rounded number.  256 additions per 4096 elements.
  __shared__ double s_run[kThreads];
  s_run[threadIdx.x] = run;
  __syncthreads();
  if (threadIdx.x == 0) {
    double acc = 0;
    for (int t = 0; t < kThreads; t++) {
      const double r = s_run[t];
      s_run[t] = acc;
      acc += r;
    }
  }
  __syncthreads();
  const double before = s_run[threadIdx.x];
#pragma unroll
  for (int i = 0; i < kItems; i++) {
    const uint64_t k = base + i;
    if (k < n) v[k] = before + x[i];
  }
  if (threadIdx.x == kThreads - 1) block_sums[(size_t)blockIdx.y * n_blocks + blockIdx.x] = before + run;
}

// Exclusive sum of the block totals, in order (a few hundred to a few thousand entries).
__global__ void scan_block_sums(double* __restrict__ block_sums, uint32_t n_blocks) {
  double* s = block_sums + (size_t)blockIdx.x * n_blocks;
  if (threadIdx.x == 0) {
    double run = 0;
    for (uint32_t b = 0; b < n_blocks; b++) {
      const double t = s[b];
      s[b] = run;
      run += t;
    }
  }
}

__global__ __launch_bounds__(kThreads) void add_offsets(double* const* __restrict__ arrays, uint64_t n,
                                                        const double* __restrict__ block_sums, uint32_t n_blocks) {
  double* v = arrays[blockIdx.y];
  const double off = block_sums[(size_t)blockIdx.y * n_blocks + blockIdx.x];
  if (blockIdx.x == 0) return;
  const uint64_t base = (uint64_t)blockIdx.x * kScanBlock;
  for (int i = threadIdx.x; i < kScanBlock; i += kThreads) {
    const uint64_t k = base + i;
    if (k < n) v[k] += off;
  }
}

// Guide cell j (r3d_tables.h GuideCell; r3d_pack.h build_guide_cells): k1 / k2 = smallest k with
// total * (j / G) resp. total * ((j + 1) / G) <= cdf[k], and the table's entries from k1 on.
__device__ uint64_t guide_bound(const double* __restrict__ cdf, uint64_t n, uint64_t j, uint64_t G) {
  const double r = cdf[n - 1] * ((double)j / (double)G);
  uint64_t lo = 0, hi = n - 1;   // first k in [0, n-1] with r <= cdf[k], else n-1
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    if (r <= cdf[mid]) hi = mid;
    else lo = mid + 1;
  }
  return lo;
}
__global__ __launch_bounds__(kThreads) void guide_kernel(const double* __restrict__ cdf, uint64_t n, uint32_t bits,
                                                         GuideCell* __restrict__ guide) {
  const uint64_t G = 1ull << bits;
  const uint64_t j = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
  if (j >= G) return;
  GuideCell c;
  const uint64_t k1 = guide_bound(cdf, n, j, G), k2 = guide_bound(cdf, n, j + 1, G);
  c.k1 = (uint32_t)k1, c.k2 = (uint32_t)k2;
#pragma unroll
  for (int i = 0; i < kGuideVals; i++)
    c.c[i] = (k2 - k1 <= (uint64_t)kGuideVals) ? cdf[k1 + i < k2 ? k1 + i : k2] : cdf[guide_pivot(k1, k2, i)];
  guide[j] = c;
}

// ---- take-off set -------------------------------------------------------------------------
// The icosahedron-based tessellation of the unit sphere the host builder makes recursively
// (host/model.cpp TesselSphereIco / split; reference S2::TesselSphere, geom_s2.cpp:60-292): 20
// faces, each split `degree` times into four by the normalised edge mid-points, a leaf's
// direction = the normalised sum of its corners, leaves in depth-first order.  Leaf k is
// therefore reached by reading k in base 4 from the top: no recursion, one work-item per leaf.
// (No fused multiply-adds here, so that the corner arithmetic matches the host's bit for bit;
// theta and phi still differ from the host's in the last place where acos / atan2 do.)
struct P3 {
  double x, y, z;
};
__device__ __forceinline__ P3 unit_sum2(P3 a, P3 b) {
#pragma clang fp contract(off)
  const double sx = a.x + b.x, sy = a.y + b.y, sz = a.z + b.z;
  const double m = sqrt(sx * sx + sy * sy + sz * sz);
  return P3{sx / m, sy / m, sz / m};
}
__global__ __launch_bounds__(kThreads) void toa_kernel(int degree, uint64_t n, double* __restrict__ toa) {
#pragma clang fp contract(off)
  const uint64_t k = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
  if (k >= n) return;
  // twelve icosahedron vertices: cyclic permutations of (+-1, 0, +-g), normalised
  const double g = (1. + sqrt(5.0)) / 2.0;
  const double m1 = sqrt(1.0 + g * g);
  const double u = 1.0 / m1, w = g / m1;
  const P3 v[12] = {{u, 0, w},  {-u, 0, w}, {u, 0, -w}, {-u, 0, -w}, {w, -u, 0}, {w, u, 0},
                    {-w, -u, 0}, {-w, u, 0}, {0, w, u},  {0, w, -u},  {0, -w, u}, {0, -w, -u}};
  const int face[20][3] = {{0, 1, 10}, {0, 1, 8},  {2, 3, 11}, {2, 3, 9},  {4, 5, 0},  {4, 5, 2},  {7, 6, 1},
                           {7, 6, 3},  {10, 11, 4}, {10, 11, 6}, {8, 9, 5},  {8, 9, 7},  {0, 10, 4}, {0, 8, 5},
                           {1, 10, 6}, {1, 8, 7},  {2, 11, 4}, {2, 9, 5},  {3, 11, 6}, {3, 9, 7}};
  const uint32_t f = (uint32_t)(k >> (2 * degree));
  P3 a = v[face[f][0]], b = v[face[f][1]], c = v[face[f][2]];
  for (int level = degree - 1; level >= 0; level--) {
    const uint32_t d = (uint32_t)(k >> (2 * level)) & 3u;
    const P3 ab = unit_sum2(a, b), bc = unit_sum2(b, c), ca = unit_sum2(c, a);
    if (d == 0) b = ab, c = ca;
    else if (d == 1) a = ab, c = bc;
    else if (d == 2) a = ca, b = bc;
    else a = bc, b = ca, c = ab;
  }
  const double sx = a.x + b.x + c.x, sy = a.y + b.y + c.y, sz = a.z + b.z + c.z;
  const double m = sqrt(sx * sx + sy * sy + sz * sz);
  toa[2 * k] = acos(sz / m);
  toa[2 * k + 1] = atan2(sy / m, sx / m);
}
// Unit vectors of the take-off directions, theta nudged away from the poles as
// Phonon::nudge_if_singular does (phonons.hpp:335-344; r3d_pack.h pack_model).
__global__ __launch_bounds__(kThreads) void toa_xyz_kernel(const double* __restrict__ toa, uint64_t n, double min_theta,
                                                           double max_theta, double* __restrict__ xyz) {
#pragma clang fp contract(off)
  const uint64_t k = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
  if (k >= n) return;
  double th = toa[2 * k];
  const double ph = toa[2 * k + 1];
  if (th < min_theta) th = min_theta;
  if (th > max_theta) th = max_theta;
  // {cos theta, cos phi, sin phi, sin theta} (r3d_tables.h KArgs::toa_dir)
  xyz[4 * k] = cos(th), xyz[4 * k + 1] = cos(ph), xyz[4 * k + 2] = sin(ph), xyz[4 * k + 3] = sin(th);
}
// ---- source radiation patterns --------------------------------------------------------------
// P, SH and SV energy radiated into each take-off direction by a moment tensor given in the local
// north-east-down frame (host/model.cpp BuildSource; reference ShearDislocation, events.cpp:66-105).
struct Moment {
  double xx, yy, zz, xy, xz, yz;
};
__global__ __launch_bounds__(kThreads) void source_kernel(Moment mt, const double* __restrict__ toa, uint64_t n,
                                                          double* __restrict__ w0, double* __restrict__ w1,
                                                          double* __restrict__ w2) {
  const uint64_t k = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
  if (k >= n) return;
  const double th = toa[2 * k], az = toa[2 * k + 1];
  const double st = sin(th), ct = cos(th), sa = sin(az), ca = cos(az);
  const double horiz = mt.xx * ca * ca + mt.xy * sin(2 * az) + mt.yy * sa * sa - mt.zz;
  const double vert = mt.xz * ca + mt.yz * sa;
  const double p = st * st * horiz + 2 * st * ct * vert + mt.zz;
  const double sh = st * (0.5 * sin(2 * az) * (mt.yy - mt.xx) + cos(2 * az) * mt.xy) + ct * (ca * mt.yz - sa * mt.xz);
  const double sv = st * ct * horiz + (1.0 - 2 * st * st) * vert;
  w0[k] = p * p, w1[k] = sh * sh, w2[k] = sv * sv;
}

// In-place cumulative sums of `count` arrays of n doubles (device pointers in host array d_arr).
hipError_t scan_arrays(double* const* d_arr, int count, uint64_t n, hipStream_t stream) {
  const uint32_t s_blocks = (uint32_t)((n + kScanBlock - 1) / kScanBlock);
  double* d_sums = nullptr;
  double** d_arrays = nullptr;
  hipError_t err = hipMalloc(&d_sums, (size_t)s_blocks * count * sizeof(double));
  if (err == hipSuccess) err = hipMalloc(&d_arrays, count * sizeof(double*));
  if (err == hipSuccess) err = hipMemcpyAsync(d_arrays, d_arr, count * sizeof(double*), hipMemcpyHostToDevice, stream);
  if (err == hipSuccess) {
    scan_blocks<<<dim3(s_blocks, count), kThreads, 0, stream>>>(d_arrays, n, d_sums, s_blocks);
    scan_block_sums<<<count, 64, 0, stream>>>(d_sums, s_blocks);
    add_offsets<<<dim3(s_blocks, count), kThreads, 0, stream>>>(d_arrays, n, d_sums, s_blocks);
    err = hipGetLastError();
  }
  if (err == hipSuccess) err = hipStreamSynchronize(stream);
  (void)hipFree(d_sums), (void)hipFree(d_arrays);
  return err;
}

}  // namespace

hipError_t build_toa_on_device(int degree, uint64_t n, double* d_toa, hipStream_t stream) {
  if (degree < 0 || degree > 12 || n != ((uint64_t)20 << (2 * degree))) return hipErrorInvalidValue;
  toa_kernel<<<(uint32_t)((n + kThreads - 1) / kThreads), kThreads, 0, stream>>>(degree, n, d_toa);
  return hipGetLastError();
}

hipError_t build_toa_xyz_on_device(const double* d_toa, uint64_t n, double min_theta, double max_theta,
                                   double* d_xyz, hipStream_t stream) {
  toa_xyz_kernel<<<(uint32_t)((n + kThreads - 1) / kThreads), kThreads, 0, stream>>>(d_toa, n, min_theta, max_theta, d_xyz);
  return hipGetLastError();
}

hipError_t build_source_tables(const double moment[6], const double* d_toa, uint64_t n, double* d_cdf[3],
                               double totals[3], hipStream_t stream) {
  if (n == 0) return hipErrorInvalidValue;
  const Moment mt{moment[0], moment[1], moment[2], moment[3], moment[4], moment[5]};
  source_kernel<<<(uint32_t)((n + kThreads - 1) / kThreads), kThreads, 0, stream>>>(mt, d_toa, n, d_cdf[0], d_cdf[1], d_cdf[2]);
  hipError_t err = hipGetLastError();
  if (err == hipSuccess) err = scan_arrays(d_cdf, 3, n, stream);
  for (int c = 0; c < 3 && err == hipSuccess; c++)
    err = hipMemcpy(&totals[c], d_cdf[c] + (n - 1), sizeof(double), hipMemcpyDeviceToHost);
  return err;
}

hipError_t build_scatterer_tables(const double het[6], double psdf_numer, const double* d_toa, uint64_t n,
                                  double* d_cdf[4], double* d_spol, double totals[4], double cos_sums[4],
                                  hipStream_t stream) {
  if (n == 0) return hipErrorInvalidValue;
  Het h{het[0], het[1], het[2], het[3], het[4], het[5], psdf_numer};
  const uint32_t g_blocks = (uint32_t)((n + kThreads - 1) / kThreads);
  const uint32_t s_blocks = (uint32_t)((n + kScanBlock - 1) / kScanBlock);
  double *d_cos = nullptr, *d_sums = nullptr;
  double** d_arrays = nullptr;
  hipError_t err = hipMalloc(&d_cos, (size_t)g_blocks * 4 * sizeof(double));
  if (err == hipSuccess) err = hipMalloc(&d_sums, (size_t)s_blocks * 4 * sizeof(double));
  if (err == hipSuccess) err = hipMalloc(&d_arrays, 4 * sizeof(double*));
  if (err == hipSuccess) err = hipMemcpyAsync(d_arrays, d_cdf, 4 * sizeof(double*), hipMemcpyHostToDevice, stream);
  if (err == hipSuccess) {
    gsato_kernel<<<g_blocks, kThreads, 0, stream>>>(h, d_toa, n, d_cdf[0], d_cdf[1], d_cdf[2], d_cdf[3], d_spol, d_cos);
    scan_blocks<<<dim3(s_blocks, 4), kThreads, 0, stream>>>(d_arrays, n, d_sums, s_blocks);
    scan_block_sums<<<4, 64, 0, stream>>>(d_sums, s_blocks);
    add_offsets<<<dim3(s_blocks, 4), kThreads, 0, stream>>>(d_arrays, n, d_sums, s_blocks);
    err = hipGetLastError();
  }
  std::vector<double> h_cos((size_t)g_blocks * 4);
  if (err == hipSuccess)
    err = hipMemcpyAsync(h_cos.data(), d_cos, h_cos.size() * sizeof(double), hipMemcpyDeviceToHost, stream);
  for (int c = 0; c < 4 && err == hipSuccess; c++)
    err = hipMemcpyAsync(&totals[c], d_cdf[c] + (n - 1), sizeof(double), hipMemcpyDeviceToHost, stream);
  if (err == hipSuccess) err = hipStreamSynchronize(stream);
  if (err == hipSuccess) {
    for (int c = 0; c < 4; c++) cos_sums[c] = 0;
    for (uint32_t b = 0; b < g_blocks; b++)
      for (int c = 0; c < 4; c++) cos_sums[c] += h_cos[(size_t)b * 4 + c];
  }
  (void)hipFree(d_cos), (void)hipFree(d_sums), (void)hipFree(d_arrays);
  return err;
}

__global__ __launch_bounds__(kThreads) void spol_cs_kernel(const double* __restrict__ spol, uint64_t n,
                                                           double* __restrict__ cs) {
  const uint64_t k = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
  if (k >= n) return;
  double s, c;
  sincos(spol[k], &s, &c);
  cs[2 * k] = c, cs[2 * k + 1] = s;
}
hipError_t build_spol_cs_on_device(const double* d_spol, uint64_t n, double* d_cs, hipStream_t stream) {
  spol_cs_kernel<<<(uint32_t)((n + kThreads - 1) / kThreads), kThreads, 0, stream>>>(d_spol, n, d_cs);
  return hipGetLastError();
}

hipError_t build_guide_on_device(const double* d_cdf, uint64_t n, uint32_t bits, GuideCell* d_guide,
                                 hipStream_t stream) {
  const uint64_t G = 1ull << bits;
  guide_kernel<<<(uint32_t)((G + kThreads - 1) / kThreads), kThreads, 0, stream>>>(d_cdf, n, bits, d_guide);
  return hipGetLastError();
}

}  // namespace r3d
