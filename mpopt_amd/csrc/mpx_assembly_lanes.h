// mpx_assembly_lanes.h -- passes of assembled contexts for batches, lane <-> evaluation point (round 5).
//
// Included in the generated translation unit of an assembled context when mpopt_amd/assembly_lanes.py found groups of point tasks
// (mpxgen::LaneGrpHES<G> / LaneGrpFGJ<G>: for mpopt_adaptive one collocation segment each, reference mpopt.py:3034-3124).  The fused
// kernels of round 3 (mpx_assembly_fused.h) walk the tables of a pass with one lane per ROW; mpx_asm_hes sat at a quarter of the HBM
// roofline on its own dependent chains -- table entry -> decode -> two LDS reads -> fma, per term (profiles/r5_adaptive_hess).  Here
// the tables ARE the code: a wavefront takes one group and 64 evaluation points, and every term of every local variable,
// multiplier and row is one v_fma_f64 with a literal coefficient over 64 points.
//
//   workgroup = one wavefront = (group g, block of 64 evaluation points)
//   1. the columns of z and lam_g the group reads (a few contiguous runs per evaluation point) -> registers -> LDS tile
//      T[column][point]: consecutive lanes on consecutive addresses of one evaluation point's run, ALL loads in flight together;
//   2. LaneGrp*<G>::run: lane = point; local variables, multipliers, the generated point functions (mpxgen::Pt<FID>::hes / jac) of
//      the group's tasks and of the halo tasks of its neighbour, then the group's rows -- the fma chains and term orders of
//      mpx_assembly_kernels.h / mpx_gather_kernel, so every value equals theirs bit for bit;
//   3. rows -> tile -> output arrays, run by run, in chunks when the group has more rows than the tile;
//   4. raw values that GLOBAL rows read (sums over nearly all tasks: f, d f / d tf ...) -> scratch[block][slot][lane]; the second
//      kernel mpx_asml_*_global (one wavefront per global row and block, lanes <-> points) sums them in the canonical order.
// All groups of a block of evaluation points run on ONE XCD (blockIdx -> (group, block) below), next to each other in time: what a
// group shares with its neighbour (the halo columns; the 64-byte sectors its output runs share with the neighbour's) meets in that L2.
#pragma once
#include <hip/hip_runtime.h>

#include "mpx_device.h"

#ifndef MPX_LANE_LDW
#define MPX_LANE_LDW 65  // doubles per tile row (64 points + 1: filled with lanes ACROSS rows, read with lanes ALONG a row)
#endif
#ifndef MPX_LANE_ABL
#define MPX_LANE_ABL 0  // ablations for timing (results wrong): 1 no loads, 2 no compute / stores
#endif

// A workgroup is ONE wavefront and its LDS operations complete in order: all the tile needs between a phase that writes it and one that
// reads it is that the compiler keeps the order.  __syncthreads() would also wait for every outstanding global store (a workgroup-scope
// release): with ten chunks of rows per group the first-order pass then ran one store round trip per chunk, 113 us instead of the
// fused kernel's 56 (profiles/r5_lanes).
#define MPX_LANE_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

namespace mpxk {

template <int NE, class ST>
struct LaneIO {
  const double* __restrict__ zb;  // the block's first evaluation point in z / lam_g (uniform)
  const double* __restrict__ lb;
  double* ob[4];                  // ... and in the output arrays
  double* __restrict__ S;         // scratch of this block + lane
  double* __restrict__ T;         // the tile
  int lane;
  double v[NE > 0 ? NE : 1];
  // A piece of 2^K columns starting at START of array SRC (0: z, 1: lam_g), tile rows E0 .. E0 + 2^K - 1: instruction i of its 2^K
  // moves the points i * PPI .. (i + 1) * PPI - 1, PPI = 64 >> K; lane = (point within the instruction) << K | column.  Every
  // address is  uniform base + a per-lane term that depends on K only + a compile-time constant  (the row strides are constants).
  // (`start`: a parameter of the GROUP, read from the code object's constant table -- uniform, it joins the base pointer; everything
  // else of a piece is the group's SHAPE, compile time)
  template <int SRC, int K, int E0>
  __device__ __forceinline__ void ld(const int start) {
    constexpr int PPI = 64 >> K, STRIDE = SRC == 0 ? MPX_LANE_ZS : MPX_LANE_LS;
    // (byte offsets in 32 bits: uniform base in scalar registers + one VGPR offset per access, no 64-bit address arithmetic per lane)
    const unsigned lo = 8u * (unsigned)((lane >> K) * STRIDE + (lane & ((1 << K) - 1)));
    const char* __restrict__ src = (const char*)((SRC == 0 ? zb : lb) + start);
#pragma unroll
    for (int i = 0; i < (1 << K); ++i) {
      // (one 32-bit add per access, pinned: left alone the compiler widens  base + lane offset  to 64 bits once and then adds every
      // constant to that with a carry chain, two vector instructions per access)
      unsigned off = lo + 8u * (unsigned)(i * PPI * STRIDE);
      asm volatile("" : "+v"(off));
      v[E0 + i] = *(const double*)(src + off);
    }
  }
  template <int K, int E0>
  __device__ __forceinline__ void put() {
    constexpr int PPI = 64 >> K;
    const int lo = (lane & ((1 << K) - 1)) * MPX_LANE_LDW + (lane >> K);
#pragma unroll
    for (int i = 0; i < (1 << K); ++i) T[lo + (E0 * MPX_LANE_LDW + i * PPI)] = v[E0 + i];
  }
  // rows E0 .. of the tile -> entries START .. of output array ARR
  template <int ARR, int K, int E0>
  __device__ __forceinline__ void st(const int start) {
    constexpr int PPI = 64 >> K, OS = ST::stride(ARR);
    const unsigned lo = 8u * (unsigned)((lane >> K) * OS + (lane & ((1 << K) - 1)));
    const int lt = (lane & ((1 << K) - 1)) * MPX_LANE_LDW + (lane >> K);
    char* __restrict__ dst = (char*)(ob[ARR] + start);
    double w[1 << K];  // (all of the piece's tile reads first: the pinned offsets below keep the order they are written in)
#pragma unroll
    for (int i = 0; i < (1 << K); ++i) w[i] = T[lt + (E0 * MPX_LANE_LDW + i * PPI)];
#pragma unroll
    for (int i = 0; i < (1 << K); ++i) {
      unsigned off = lo + 8u * (unsigned)(i * PPI * OS);
      asm volatile("" : "+v"(off));
      *(double*)(dst + off) = w[i];
    }
  }
};

// (Blocks are always whole: the last block of a batch that is no multiple of 64 starts at B - 64 and repeats a few evaluation points
// of its neighbour -- the same values into the same places; B >= 64.)
template <class GR, class ST>
__device__ __forceinline__ void lane_group(const ::MpxLaneArgs& A, int blk, double* __restrict__ T, const int* __restrict__ par) {
  const int lane = threadIdx.x;
  const int64_t b0 = (int64_t)blk * 64 + 64 <= A.B ? (int64_t)blk * 64 : (int64_t)A.B - 64;
  LaneIO<GR::NE, ST> io{A.z + b0 * MPX_LANE_ZS, A.lam ? A.lam + b0 * MPX_LANE_LS : nullptr, {nullptr, nullptr, nullptr, nullptr},
                        A.scratch ? A.scratch + ((int64_t)blk * ST::NSID) * 64 + lane : nullptr, T, lane, {}};
#pragma unroll
  for (int a = 0; a < ST::NARR; ++a) io.ob[a] = A.out[a] + b0 * ST::stride(a);
  if (!(MPX_LANE_ABL & 1)) GR::load(io, par);
  const double sg = A.sigma ? A.sigma[b0 + lane] : 0.0;
  GR::fill(io);
  MPX_LANE_SYNC();
  if (!(MPX_LANE_ABL & 2)) GR::run(io, sg, par);
}

// (G: a group SHAPE -- the groups of a shape run the same struct with their own parameter rows `par`)
template <template <int> class GRT, class ST, int G>
struct LaneDispatch {
  __device__ static __forceinline__ void run(const ::MpxLaneArgs& A, int shape, int blk, double* T, const int* par) {
    if (shape == G)
      lane_group<GRT<G>, ST>(A, blk, T, par);
    else
      LaneDispatch<GRT, ST, G - 1>::run(A, shape, blk, T, par);
  }
};
template <template <int> class GRT, class ST>
struct LaneDispatch<GRT, ST, -1> {
  __device__ static __forceinline__ void run(const ::MpxLaneArgs&, int, int, double*, const int*) {}
};

// One global row of one block of evaluation points: lane <-> point, the row's terms from the constant table (uniform: scalar loads),
// its values from the scratch slots / z / 1.0.  Rows up to THR terms: one chain from 0 in order; longer rows: partial sum j takes the
// terms j, j + 64, ... in order, then the pairwise tree over the 64 partial sums -- what a wavefront of mpx_gather_kernel computes with
// mpx_wave_total (a missing partial sum is +0.0 there and here).  The generator pads every row with terms (0.0, the constant 1.0) to a
// multiple of 8 (64 for the long rows: glong): a padding term adds +0.0, and the loops below have no branch between the loads of a round
// -- all of them in flight together (with one branch per term every value was its own round trip: 35 us for three rows of 101 terms).
__device__ const double lane_one_ = 1.0;
template <class ST>
__device__ __forceinline__ void lane_global_row(const ::MpxLaneArgs& A, int blk, int row, const int* __restrict__ gptr, const int* __restrict__ gsrc,
                                                const double* __restrict__ gcoef, const int* __restrict__ garr, const int* __restrict__ gidx,
                                                const int* __restrict__ glong) {
  const int lane = threadIdx.x;
  const int64_t b0 = (int64_t)blk * 64 + 64 <= A.B ? (int64_t)blk * 64 : (int64_t)A.B - 64;
  const char* __restrict__ Sb = (const char*)(A.scratch + ((int64_t)blk * ST::NSID) * 64);
  const char* __restrict__ zb = (const char*)(A.z + b0 * MPX_LANE_ZS);
  const int e0 = gptr[row], e1 = gptr[row + 1];
  auto val = [&](int e) -> double {  // (branch-free: a uniform base and a uniform per-lane stride select the source)
    const int k = gsrc[e];
    const char* base = k >= 0 ? Sb + (int64_t)k * 512 : (k == -1 ? (const char*)&lane_one_ : zb + (int64_t)(-2 - k) * 8);
    const int mult = k >= 0 ? 8 : (k == -1 ? 0 : MPX_LANE_ZS * 8);
    return *(const double*)(base + (int64_t)lane * mult);
  };
  double s;
  if (!glong[row]) {
    s = 0.0;
    for (int e = e0; e < e1; e += 8) {
      double v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = val(e + j);
#pragma unroll
      for (int j = 0; j < 8; ++j) s = __builtin_fma(gcoef[e + j], v[j], s);
    }
  } else {
    double p[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) p[j] = 0.0;
    for (int e = e0; e < e1; e += 64) {
      double v[64];
#pragma unroll
      for (int j = 0; j < 64; ++j) v[j] = val(e + j);
#pragma unroll
      for (int j = 0; j < 64; ++j) p[j] = __builtin_fma(gcoef[e + j], v[j], p[j]);
    }
#pragma unroll
    for (int w = 1; w < 64; w *= 2) {
#pragma unroll
      for (int j = 0; j < 64; j += 2 * w) p[j] = p[j + w] + p[j];
    }
    s = p[0];
  }
  const int a = garr[row];
  int64_t os = ST::stride(0);
#pragma unroll
  for (int q = 1; q < ST::NARR; ++q) os = a == q ? (int64_t)ST::stride(q) : os;
  double* out = A.out[0];
#pragma unroll
  for (int q = 1; q < ST::NARR; ++q) out = a == q ? A.out[q] : out;
  out[(b0 + lane) * os + gidx[row]] = s;
}

}  // namespace mpxk

// Two wavefronts per SIMD (<= 256 registers instead of the 260 the allocator takes when left alone): a batch of 4096 evaluation points
// is 1280 wavefronts, and with five per compute unit (LDS) they are all resident at once -- 21.6 -> 18.3 us at moon lander 20x5
// (profiles/r5_lanes); 0: no bound.
// (Every wavefront of a small batch starts at the same time and the phases of a group take the same time everywhere: loads, then
// arithmetic, then stores, chip-wide.  MPX_LANE_STAGGER=n: every other wavefront starts n x 64 x 64 cycles late -- measured, see
// profiles/r5_lanes.)
#ifndef MPX_LANE_STAGGER
#define MPX_LANE_STAGGER 0
#endif
#ifndef MPX_LANE_WAVES_PER_EU
#define MPX_LANE_WAVES_PER_EU 2
#endif
#if MPX_LANE_WAVES_PER_EU > 0
#define MPX_LANE_OCC __attribute__((amdgpu_waves_per_eu(MPX_LANE_WAVES_PER_EU)))
#else
#define MPX_LANE_OCC
#endif
// mpx_asml_<pass>_info: {groups, tile doubles, check (nnz of the pass's reordered pattern), global rows, scratch slots per block,
// hash of that pattern in its order (assembly_lanes.py: pattern_hash)}
#define MPX_INSTANTIATE_LANES(kind, KIND, PASS)                                                                                 \
  namespace mpxgen {                                                                                                            \
  struct LaneST##KIND {                                                                                                         \
    static constexpr int NSID = MPX_LANE_##KIND##_NSID, THR = MPX_LANE_##KIND##_THR;                                           \
    static constexpr int strides_[] = MPX_LANE_##KIND##_STRIDES;                                                                \
    static constexpr int NARR = sizeof(strides_) / sizeof(int);                                                                 \
    __host__ __device__ static constexpr int stride(int a) { return strides_[a]; }                                             \
  };                                                                                                                            \
  }                                                                                                                             \
  extern "C" __device__ __attribute__((used)) const int mpx_asml_##kind##_info[6] = {                                          \
      MPX_LANE_##KIND##_GROUPS, MPX_LANE_##KIND##_TILE_ROWS * MPX_LANE_LDW, MPX_LANE_##KIND##_CHECK, MPX_LANE_##KIND##_NGLOBAL, \
      MPX_LANE_##KIND##_NSID, MPX_LANE_##KIND##_HASH};                                                                          \
  extern "C" __global__ __launch_bounds__(64) MPX_LANE_OCC void mpx_asml_##kind(const MpxLaneArgs A) {                         \
    __shared__ double T[MPX_LANE_##KIND##_TILE_ROWS * MPX_LANE_LDW];                                                            \
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, nb8 = (A.n_blocks + 7) >> 3;                                         \
    const int g = A.order ? idx / nb8 : idx % MPX_LANE_##KIND##_GROUPS;                                                         \
    const int blk = (A.order ? idx % nb8 : idx / MPX_LANE_##KIND##_GROUPS) * 8 + xcd;                                           \
    if (blk >= A.n_blocks) return;                                                                                              \
    if (MPX_LANE_STAGGER > 0 && (idx & 1))                                                                                      \
      for (int k = 0; k < MPX_LANE_STAGGER; ++k) __builtin_amdgcn_s_sleep(64);                                                  \
    mpxk::LaneDispatch<mpxgen::LaneGrp##KIND, mpxgen::LaneST##KIND, MPX_LANE_##KIND##_SHAPES - 1>::run(                        \
        A, mpxgen::lane_shape_##kind[g], blk, T, mpxgen::lane_par_##kind + mpxgen::lane_poff_##kind[g]);                       \
  }                                                                                                                             \
  extern "C" __global__ __launch_bounds__(64) void mpx_asml_##kind##_global(const MpxLaneArgs A) {                              \
    if constexpr (MPX_LANE_##KIND##_NGLOBAL > 0)                                                                                \
      mpxk::lane_global_row<mpxgen::LaneST##KIND>(A, blockIdx.x, blockIdx.y, mpxgen::lane_gptr_##kind, mpxgen::lane_gsrc_##kind, \
                                                  mpxgen::lane_gcoef_##kind, mpxgen::lane_garr_##kind, mpxgen::lane_gidx_##kind, mpxgen::lane_glong_##kind); \
  }
