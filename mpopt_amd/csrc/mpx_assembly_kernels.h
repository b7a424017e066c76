// mpx_assembly_kernels.h -- point kernels of assembled contexts (include/mpx.h, mpx_create_assembled).
//
// Compiled per problem together with the generated `mpxgen::Pt<FID>` structs (mpopt_amd/assembly.py):
// straight-line code of a point function phi(loc; cst) -> out, its structural Jacobian entries and the
// structural entries of the Hessian of sum_r mu_r * out_r.
//
// One launch covers every point set of the context (64-lane blocks, a block belongs to one set and
// dispatches on its function id); lane <-> point.  A lane gathers its local variables as short
// fixed-order sums over z (node values: one term; interpolated mid-point values: degree+1 terms; the
// running width sum: one term per earlier segment), evaluates the generated code and writes every result
// to raw[slot][point], i.e. one contiguous run per slot and wavefront.  No reductions happen here: all
// sums over points are rows of the gather pass (mpx_assembly.cpp), whose order is fixed.
#pragma once
#include <hip/hip_runtime.h>

#include "mpx_device.h"

namespace mpxk {

#ifndef MPX_PTS_UNROLL
#define MPX_PTS_UNROLL 1
#endif

template <int FID, int MODE, int U>
__device__ __forceinline__ void point_eval(const MpxPtSet& S, const MpxPtCall& A, int p, int64_t n, const double* cst, double* __restrict__ raw0, int b) {
  using F = mpxgen::Pt<FID>;
  constexpr int NLOC = F::NLOC, NOUT = F::NOUT, NJ = F::NJ, NH = F::NH;
  const double* __restrict__ zb = A.z + (int64_t)b * A.z_stride;
  double loc[U][NLOC > 0 ? NLOC : 1];
#pragma unroll
  for (int v = 0; v < NLOC; ++v) {
    double acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] = 0;
    for (int t = S.loc_toff[v]; t < S.loc_toff[v + 1]; ++t) {
      const double cf = S.loc_coef[(int64_t)t * n + p];
      const int ix = S.loc_idx[(int64_t)t * n + p];
#pragma unroll
      for (int u = 0; u < U; ++u) acc[u] = fma(cf, zb[(int64_t)u * A.z_stride + ix], acc[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) loc[u][v] = acc[u];
  }
  if constexpr (MODE == MPX_MODE_HESS) {
    double mu[U][NOUT > 0 ? NOUT : 1];
    const double* __restrict__ lb = A.lam + (int64_t)b * A.lam_stride;
#pragma unroll
    for (int r = 0; r < NOUT; ++r) {
      double acc[U];
#pragma unroll
      for (int u = 0; u < U; ++u) acc[u] = 0;
      for (int t = S.mu_toff[r]; t < S.mu_toff[r + 1]; ++t) {
        const int ix = S.mu_idx[(int64_t)t * n + p];
        const double cf = S.mu_coef[(int64_t)t * n + p];
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = fma(cf, ix == A.n_g ? A.sigma[b + u] : lb[(int64_t)u * A.lam_stride + ix], acc[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) mu[u][r] = acc[u];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      double* __restrict__ rb = raw0 + (int64_t)(b + u) * A.raw_stride;
      double H[NH > 0 ? NH : 1];
      F::hes(loc[u], cst, mu[u], H);
#pragma unroll
      for (int q = 0; q < NH; ++q) rb[(int64_t)q * n + p] = H[q];
    }
  } else {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      double* __restrict__ rb = raw0 + (int64_t)(b + u) * A.raw_stride;
      double out[NOUT > 0 ? NOUT : 1];
      if constexpr (MODE == MPX_MODE_FGJ) {
        double J[NJ > 0 ? NJ : 1];
        F::jac(loc[u], cst, out, J);
#pragma unroll
        for (int q = 0; q < NJ; ++q) rb[(int64_t)(NOUT + q) * n + p] = J[q];
      } else {
        F::val(loc[u], cst, out);
      }
#pragma unroll
      for (int r = 0; r < NOUT; ++r) rb[(int64_t)r * n + p] = out[r];
    }
  }
}

template <int FID, int MODE>
__device__ __forceinline__ void point_body(const MpxPtSet& S, const MpxPtCall& A, int blk) {
  using F = mpxgen::Pt<FID>;
  constexpr int NLOC = F::NLOC, NCST = F::NCST, NOUT = F::NOUT, NJ = F::NJ, NH = F::NH;
  if constexpr (MODE == MPX_MODE_HESS && NH == 0) return;
  const int p = blk * 64 + threadIdx.x;
  if (p >= S.n) return;
  const int64_t n = S.n;
  double cst[NCST > 0 ? NCST : 1];
#pragma unroll
  for (int k = 0; k < NCST; ++k) cst[k] = S.cst[(int64_t)k * n + p];
  const int b0 = blockIdx.y * A.b_per_block;
  const int b1 = (b0 + A.b_per_block < A.B) ? b0 + A.b_per_block : A.B;
  double* __restrict__ raw0 = A.raw + (MODE == MPX_MODE_HESS ? S.rawh_off : S.raw_off);
  int b = b0;
  // Several evaluation points per lane go through the gathers TOGETHER (one read of a term's index and coefficient, U loads
  // of z in flight); per point the terms are added in the same order as in the single-point pass, so results are identical.
  if constexpr (MPX_PTS_UNROLL > 1)
    for (; b + MPX_PTS_UNROLL <= b1; b += MPX_PTS_UNROLL) point_eval<FID, MODE, MPX_PTS_UNROLL>(S, A, p, n, cst, raw0, b);
  for (; b < b1; ++b) point_eval<FID, MODE, 1>(S, A, p, n, cst, raw0, b);
}

template <int MODE, int FID>
struct PtDispatch {
  __device__ static __forceinline__ void run(const MpxPtSet& S, const MpxPtCall& A, int blk) {
    if (S.fid == FID)
      point_body<FID, MODE>(S, A, blk);
    else
      PtDispatch<MODE, FID - 1>::run(S, A, blk);
  }
};
template <int MODE>
struct PtDispatch<MODE, -1> {
  __device__ static __forceinline__ void run(const MpxPtSet&, const MpxPtCall&, int) {}
};

template <int MODE, int NF>
__device__ __forceinline__ void points_body(const MpxPtCall& A) {
  // rotate the block index with the evaluation point so that a multiple-of-8 block count does not pin every
  // block to one XCD (see mpx_gather_kernel)
  const int bx = (int)((blockIdx.x + blockIdx.y) % gridDim.x);
  int k = 0;
  while (k + 1 < A.n_sets && bx >= A.sets[k + 1].block_first) ++k;
  const MpxPtSet S = A.sets[k];
  PtDispatch<MODE, NF - 1>::run(S, A, bx - S.block_first);
}

}  // namespace mpxk

// mpx_pts_points_per_lane tells the host how many evaluation points a lane of these kernels takes together.
#define MPX_INSTANTIATE_POINTS(NF)                                                                       \
  extern "C" __device__ __attribute__((used)) const int mpx_pts_points_per_lane = MPX_PTS_UNROLL;        \
  extern "C" __global__ __launch_bounds__(64) void mpx_pts_val(const MpxPtCall A) {                      \
    mpxk::points_body<MPX_MODE_FG, NF>(A);                                                               \
  }                                                                                                      \
  extern "C" __global__ __launch_bounds__(64) void mpx_pts_jac(const MpxPtCall A) {                      \
    mpxk::points_body<MPX_MODE_FGJ, NF>(A);                                                              \
  }                                                                                                      \
  extern "C" __global__ __launch_bounds__(64) void mpx_pts_hes(const MpxPtCall A) {                      \
    mpxk::points_body<MPX_MODE_HESS, NF>(A);                                                             \
  }
