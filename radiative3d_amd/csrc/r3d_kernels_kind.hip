// r3d_kernels_kind.hip -- the traversal kernels of ONE cell kind (-DR3D_KIND=0 cylinder, 1 tetra,
// 2 sphere shell; the Makefile compiles this file three times): pool_kernel<KIND, ...> diagnostic
// and chain-step variants, pool_job_kernel<KIND, ...> and pool_drain_kernel<KIND, ...> for every table
// residency the kind can run with, and the launch entry points r3d_engine.hip calls (r3d_kernels.h).
//
// One unit per kind because the kinds want different instruction scheduling: the spherical-shell
// and layered kernels run 3.4 % / 0.4 % faster under the compiler's max-ILP strategy, the tetra
// kernel 1.4 % slower (it then spills ten registers) -- and three units compile side by side.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "../../include/r3d.h"
#include "r3d_kernels.h"
#include "r3d_pool.h"

#ifndef R3D_KIND
#error "compile with -DR3D_KIND=0 (cylinder), 1 (tetra) or 2 (sphere shell)"
#endif

namespace r3d {
namespace {

constexpr int K = R3D_KIND;

// Call f(cells_in_lds, scatterer_heads_in_lds) for the residency `res` as compile-time constants.
// (Tetra grids run to thousands of cells: their records are never staged in LDS, so that variant is
//  not compiled.)
template <class F>
hipError_t with_res(int res, F&& f) {
  if (res == RES_NONE) return f(std::false_type{}, std::false_type{});
  if (res == RES_TABLES || K == CELL_TET) return f(std::false_type{}, std::true_type{});
  if constexpr (K != CELL_TET) return f(std::true_type{}, std::true_type{});
  return hipErrorInvalidValue;
}

hipError_t launch_kind(int res, bool trace, bool drain_only, unsigned grid, size_t lds_bytes, hipStream_t s,
                       const KArgs& a) {
  return with_res(res, [&](auto cells, auto scat) {
    constexpr bool C = decltype(cells)::value, H = decltype(scat)::value;
    if (drain_only && !trace)
      hipLaunchKernelGGL((pool_drain_kernel<K, C, H>), dim3(grid), dim3(kPoolBlock), lds_bytes, s, a);
    else if (trace)
      hipLaunchKernelGGL((pool_kernel<K, C, H, true>), dim3(grid), dim3(kPoolBlock), lds_bytes, s, a);
    else if (a.carry_out)   // a step launch of a chain: what is unfinished is parked for the next one
      hipLaunchKernelGGL((pool_kernel<K, C, H, false>), dim3(grid), dim3(kPoolBlock), lds_bytes, s, a);
    else                    // a launch that drains its own stragglers
      hipLaunchKernelGGL((pool_job_kernel<K, C, H>), dim3(grid), dim3(kPoolBlock), lds_bytes, s, a);
    return hipGetLastError();
  });
}

hipError_t lds_attr_kind(int res, int lds_bytes) {
  return with_res(res, [&](auto cells, auto scat) {
    constexpr bool C = decltype(cells)::value, H = decltype(scat)::value;
    hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(&pool_kernel<K, C, H, false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (r != hipSuccess) return r;
    r = hipFuncSetAttribute(reinterpret_cast<const void*>(&pool_drain_kernel<K, C, H>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (r != hipSuccess) return r;
    r = hipFuncSetAttribute(reinterpret_cast<const void*>(&pool_job_kernel<K, C, H>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (r != hipSuccess) return r;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&pool_kernel<K, C, H, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  });
}

#ifdef R3D_PHASE_TIMING
int stats_kind(unsigned long long out[40]) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pool_stats), 40 * sizeof(unsigned long long)) != hipSuccess) return 1;
  unsigned long long zero[40] = {};
  return hipMemcpyToSymbol(HIP_SYMBOL(g_pool_stats), zero, sizeof zero) != hipSuccess;
}
#endif

}  // namespace

#if R3D_KIND == 0
#define R3D_KIND_NAME(stem) stem##_cyl
#elif R3D_KIND == 1
#define R3D_KIND_NAME(stem) stem##_tet
#else
#define R3D_KIND_NAME(stem) stem##_sph
#endif

hipError_t R3D_KIND_NAME(launch_pool)(int res, bool trace, bool drain_only, unsigned grid, size_t lds_bytes,
                                      hipStream_t s, const KArgs& a) {
  return launch_kind(res, trace, drain_only, grid, lds_bytes, s, a);
}
hipError_t R3D_KIND_NAME(pool_lds_attr)(int res, int lds_bytes) { return lds_attr_kind(res, lds_bytes); }
#ifdef R3D_PHASE_TIMING
int R3D_KIND_NAME(pool_stats)(unsigned long long out[40]) { return stats_kind(out); }
#endif

}  // namespace r3d
