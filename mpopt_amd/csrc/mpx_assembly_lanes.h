// mpx_assembly_lanes.h -- hess_l of assembled contexts for batches, lane <-> evaluation point (round 5).
//
// Included at the end of the generated translation unit of an assembled context when mpopt_amd/assembly_lanes.py found groups of
// point tasks (mpxgen::LaneGrp<G>: for mpopt_adaptive one collocation segment each, reference mpopt.py:3034-3124).  The fused
// kernel of round 3 (mpx_assembly_fused.h, mpx_asm_hes) walks the tables of the pass with one lane per ROW and sits at a quarter of
// the HBM roofline on its own dependent chains -- table entry -> decode -> two LDS reads -> fma, per term
// (profiles/r5_adaptive_hess).  Here the tables ARE the code: a wavefront takes one group and 64 evaluation points, and every term of
// every local variable, multiplier and Hessian row is one v_fma_f64 with a literal coefficient over 64 points.
//
//   workgroup = one wavefront = (group g, block of 64 evaluation points)
//   1. the columns of z and lam_g the group reads (a few contiguous runs per evaluation point) -> registers -> LDS tile
//      T[column][point]: consecutive lanes on consecutive addresses of one evaluation point's run, ALL loads in flight together;
//   2. LaneGrp<G>::compute: lane = point; local variables, multipliers, the generated point Hessians (mpxgen::Pt<FID>::hes) of the
//      group's tasks and of the halo tasks of its neighbour, then the group's rows of hess_l -- the fma chains and term orders of
//      mpx_assembly_kernels.h / mpx_gather_kernel, so every value equals theirs bit for bit;
//   3. rows -> the same tile -> hess_val, again run by run.
// All groups of a block of evaluation points run on ONE XCD (blockIdx -> (group, block) below), next to each other in time: what a
// group shares with its neighbour (the halo columns; the 64-byte sectors its output runs share with the neighbour's) meets in that L2.
#pragma once
#include <hip/hip_runtime.h>

#include "mpx_device.h"

#ifndef MPX_LANE_LDW
#define MPX_LANE_LDW 65  // doubles per tile row (64 points + 1: filled with lanes ACROSS rows, read with lanes ALONG a row)
#endif

namespace mpxk {

template <int NE>
struct LaneIO {
  const double* __restrict__ zb;  // the block's first evaluation point in z / lam_g / hess_val (uniform)
  const double* __restrict__ lb;
  double* __restrict__ ob;
  double* __restrict__ T;  // the tile
  int lane;
  double v[NE > 0 ? NE : 1];
  // A piece of 2^K columns starting at START of array SRC (0: z, 1: lam_g), tile rows E0 .. E0 + 2^K - 1: instruction i of its 2^K
  // moves the points i * PPI .. (i + 1) * PPI - 1, PPI = 64 >> K; lane = (point within the instruction) << K | column.  Every
  // address is  uniform base + a per-lane term that depends on K only + a compile-time constant  (the row strides are constants).
  template <int SRC, int START, int K, int E0>
  __device__ __forceinline__ void ld() {
    constexpr int PPI = 64 >> K, STRIDE = SRC == 0 ? MPX_LANE_ZS : MPX_LANE_LS;
    // (byte offsets in 32 bits: uniform base in scalar registers + one VGPR offset per access, no 64-bit address arithmetic per lane)
    const unsigned lo = 8u * (unsigned)((lane >> K) * STRIDE + (lane & ((1 << K) - 1)));
    const char* __restrict__ src = (const char*)(SRC == 0 ? zb : lb);
#pragma unroll
    for (int i = 0; i < (1 << K); ++i) {
      // (one 32-bit add per access, pinned: left alone the compiler widens  base + lane offset  to 64 bits once and then adds every
      // constant to that with a carry chain, two vector instructions per access)
      unsigned off = lo + 8u * (unsigned)(i * PPI * STRIDE + START);
      asm volatile("" : "+v"(off));
      v[E0 + i] = *(const double*)(src + off);
    }
  }
  template <int SRC, int START, int K, int E0>
  __device__ __forceinline__ void put() {
    constexpr int PPI = 64 >> K;
    const int lo = (lane & ((1 << K) - 1)) * MPX_LANE_LDW + (lane >> K);
#pragma unroll
    for (int i = 0; i < (1 << K); ++i) T[lo + (E0 * MPX_LANE_LDW + i * PPI)] = v[E0 + i];
  }
  template <int START, int K, int E0>
  __device__ __forceinline__ void st() {
    constexpr int PPI = 64 >> K;
    const unsigned lo = 8u * (unsigned)((lane >> K) * MPX_LANE_OS + (lane & ((1 << K) - 1)));
    const int lt = (lane & ((1 << K) - 1)) * MPX_LANE_LDW + (lane >> K);
    char* __restrict__ dst = (char*)ob;
    double w[1 << K];  // (all of the piece's tile reads first: the pinned offsets below keep the order they are written in)
#pragma unroll
    for (int i = 0; i < (1 << K); ++i) w[i] = T[lt + (E0 * MPX_LANE_LDW + i * PPI)];
#pragma unroll
    for (int i = 0; i < (1 << K); ++i) {
      unsigned off = lo + 8u * (unsigned)(i * PPI * MPX_LANE_OS + START);
      asm volatile("" : "+v"(off));
      *(double*)(dst + off) = w[i];
    }
  }
};

// (Blocks are always whole: the last block of a batch that is no multiple of 64 starts at B - 64 and repeats a few evaluation points
// of its neighbour -- the same values into the same places; B >= 64.)
#ifndef MPX_LANE_ABL
#define MPX_LANE_ABL 0  // ablations for timing (results wrong): 1 no loads, 2 no compute, 4 no stores
#endif
template <int G>
__device__ __forceinline__ void lane_group(const ::MpxLaneArgs& A, int blk, double* __restrict__ T) {
  using GR = mpxgen::LaneGrp<G>;
  const int lane = threadIdx.x;
  const int64_t b0 = (int64_t)blk * 64 + 64 <= A.B ? (int64_t)blk * 64 : (int64_t)A.B - 64;
  LaneIO<GR::NE> io{A.z + b0 * MPX_LANE_ZS, A.lam + b0 * MPX_LANE_LS, A.out + b0 * MPX_LANE_OS, T, lane, {}};
  if (!(MPX_LANE_ABL & 1)) GR::load(io);
  const double sg = A.sigma[b0 + lane];
  GR::fill(io);
  __syncthreads();
  double R[GR::NR];
  if (!(MPX_LANE_ABL & 2)) {
    GR::compute(T + lane, sg, R);
  } else {
#pragma unroll
    for (int r = 0; r < GR::NR; ++r) R[r] = T[(r % GR::NE) * MPX_LANE_LDW + lane] + sg;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < GR::NR; ++r) T[r * MPX_LANE_LDW + lane] = R[r];
  __syncthreads();
  if (!(MPX_LANE_ABL & 4)) GR::store(io);
}

template <int G>
struct LaneDispatch {
  __device__ static __forceinline__ void run(const ::MpxLaneArgs& A, int g, int blk, double* T) {
    if (g == G)
      lane_group<G>(A, blk, T);
    else
      LaneDispatch<G - 1>::run(A, g, blk, T);
  }
};
template <>
struct LaneDispatch<-1> {
  __device__ static __forceinline__ void run(const ::MpxLaneArgs&, int, int, double*) {}
};

}  // namespace mpxk

// Two wavefronts per SIMD (<= 256 registers instead of the 260 the allocator takes when left alone): a batch of 4096 evaluation points
// is 1280 wavefronts, and with five per compute unit (LDS) they are all resident at once -- 21.6 -> 18.3 us at moon lander 20x5
// (profiles/r5_lanes); 0: no bound.
#ifndef MPX_LANE_WAVES_PER_EU
#define MPX_LANE_WAVES_PER_EU 2
#endif
#if MPX_LANE_WAVES_PER_EU > 0
#define MPX_LANE_OCC __attribute__((amdgpu_waves_per_eu(MPX_LANE_WAVES_PER_EU)))
#else
#define MPX_LANE_OCC
#endif
// mpx_asml_info: {groups, tile doubles, nnz(hess_l) the code was generated for}
#define MPX_INSTANTIATE_LANES_HESS                                                                                             \
  extern "C" __device__ __attribute__((used)) const int mpx_asml_info[3] = {MPX_LANE_GROUPS, MPX_LANE_NE_MAX * MPX_LANE_LDW, MPX_LANE_NNZH}; \
  extern "C" __global__ __launch_bounds__(64) MPX_LANE_OCC void mpx_asml_hes(const MpxLaneArgs A) {                                         \
    __shared__ double T[MPX_LANE_NE_MAX * MPX_LANE_LDW];                                                                       \
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;                                                                     \
    const int g = idx % MPX_LANE_GROUPS, blk = (idx / MPX_LANE_GROUPS) * 8 + xcd;                                              \
    if (blk >= A.n_blocks) return;                                                                                             \
    mpxk::LaneDispatch<MPX_LANE_GROUPS - 1>::run(A, g, blk, T);                                                                \
  }
