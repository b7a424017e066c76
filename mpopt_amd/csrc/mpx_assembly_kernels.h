// mpx_assembly_kernels.h -- point kernels of assembled contexts (include/mpx.h, mpx_create_assembled).
//
// Compiled per problem together with the generated `mpxgen::Pt<FID>` structs (mpopt_amd/assembly.py):
// straight-line code of a point function phi(loc; cst) -> out, its structural Jacobian entries and the
// structural entries of the Hessian of sum_r mu_r * out_r.
//
// lane <-> point.  A lane gathers its local variables as short fixed-order sums over z (node values: one
// term; interpolated mid-point values: degree+1 terms), evaluates the generated code and writes every
// result to raw[slot][point], i.e. one contiguous run per slot and wavefront.  No reductions happen here:
// all sums over points are rows of the gather pass (mpx_assembly.cpp), whose order is fixed.
#pragma once
#include <hip/hip_runtime.h>

#include "mpx_device.h"

namespace mpxk {

template <int FID, int MODE>
__device__ __forceinline__ void point_body(const MpxPtArgs& A) {
  using F = mpxgen::Pt<FID>;
  constexpr int NLOC = F::NLOC, NCST = F::NCST, NOUT = F::NOUT, NJ = F::NJ, NH = F::NH;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= A.n) return;
  const int64_t n = A.n;
  double cst[NCST > 0 ? NCST : 1];
#pragma unroll
  for (int k = 0; k < NCST; ++k) cst[k] = A.cst[(int64_t)k * n + p];
  const int b0 = blockIdx.y * A.b_per_block;
  const int b1 = (b0 + A.b_per_block < A.B) ? b0 + A.b_per_block : A.B;
  for (int b = b0; b < b1; ++b) {
    const double* __restrict__ zb = A.z + (int64_t)b * A.z_stride;
    double loc[NLOC > 0 ? NLOC : 1];
#pragma unroll
    for (int v = 0; v < NLOC; ++v) {
      double acc = 0;
      for (int t = A.loc_toff[v]; t < A.loc_toff[v + 1]; ++t) acc = fma(A.loc_coef[(int64_t)t * n + p], zb[A.loc_idx[(int64_t)t * n + p]], acc);
      loc[v] = acc;
    }
    double* __restrict__ rb = A.raw + (int64_t)b * A.raw_stride;
    if constexpr (MODE == MPX_MODE_HESS) {
      double mu[NOUT > 0 ? NOUT : 1];
      const double* __restrict__ lb = A.lam + (int64_t)b * A.lam_stride;
      const double sg = A.sigma[b];
#pragma unroll
      for (int r = 0; r < NOUT; ++r) {
        double acc = 0;
        for (int t = A.mu_toff[r]; t < A.mu_toff[r + 1]; ++t) {
          const int ix = A.mu_idx[(int64_t)t * n + p];
          acc = fma(A.mu_coef[(int64_t)t * n + p], ix == A.n_g ? sg : lb[ix], acc);
        }
        mu[r] = acc;
      }
      double H[NH > 0 ? NH : 1];
      F::hes(loc, cst, mu, H);
#pragma unroll
      for (int q = 0; q < NH; ++q) rb[(int64_t)q * n + p] = H[q];
    } else {
      double out[NOUT > 0 ? NOUT : 1];
      if constexpr (MODE == MPX_MODE_FGJ) {
        double J[NJ > 0 ? NJ : 1];
        F::jac(loc, cst, out, J);
#pragma unroll
        for (int q = 0; q < NJ; ++q) rb[(int64_t)(NOUT + q) * n + p] = J[q];
      } else {
        F::val(loc, cst, out);
      }
#pragma unroll
      for (int r = 0; r < NOUT; ++r) rb[(int64_t)r * n + p] = out[r];
    }
  }
}

}  // namespace mpxk

#define MPX_INSTANTIATE_POINTS(FID)                                                                      \
  extern "C" __global__ __launch_bounds__(64) void mpx_pt_val_##FID(const MpxPtArgs A) {                 \
    mpxk::point_body<FID, MPX_MODE_FG>(A);                                                               \
  }                                                                                                      \
  extern "C" __global__ __launch_bounds__(64) void mpx_pt_jac_##FID(const MpxPtArgs A) {                 \
    mpxk::point_body<FID, MPX_MODE_FGJ>(A);                                                              \
  }                                                                                                      \
  extern "C" __global__ __launch_bounds__(64) void mpx_pt_hes_##FID(const MpxPtArgs A) {                 \
    mpxk::point_body<FID, MPX_MODE_HESS>(A);                                                             \
  }
