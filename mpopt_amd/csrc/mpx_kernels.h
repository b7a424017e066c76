// mpx_kernels.h -- hand-written gfx950 (MI355X, CDNA4) kernels of the collocation hot path.
//
// Compiled once per problem together with the generated `mpxgen::Phase<PH>` structs
// (mpopt_amd/codegen.py), which hold the traced dynamics / path / cost functions and their
// symbolic first and second derivatives as straight-line code.
//
// What these kernels replace: CasADi's SX virtual machine evaluating nlp_f / nlp_g / nlp_grad_f /
// nlp_jac_g / nlp_hess_l for mpopt's transcription (created at mpopt.py:757, called from IPOPT at
// mpopt.py:804).  The mathematics per node follows mpopt.py:154-237, 330-377, 455 (see
// codegen.py and DESIGN.md section 3).
//
// Mapping to the hardware:
//   * one workgroup (4 wavefronts x 64 lanes) = one tile of whole collocation segments of one
//     (phase, degree) bucket; lane <-> collocation node; the tile loops over a chunk of the
//     batch so per-lane tables stay in registers;
//   * the lane's row of the differentiation matrix D and of the mid-point interpolation matrix
//     live in VGPRs for the whole batch loop; the segment's states/controls are staged through a
//     double-buffered LDS tile so the D.X contraction reads neighbours from LDS, one barrier per
//     evaluation point;
//   * every global access is lane-contiguous: z is state-major so X[.,a] is a run over nodes; the
//     Jacobian/Hessian value arrays are laid out tile-major / slot-major / lane-minor, i.e. each
//     store instruction of a wavefront writes one contiguous run;
//   * scalar sums (f, d/dt0, d/dtf, d/da, Hessian corner) are reduced in fixed order: DPP
//     shuffle tree per wavefront -> LDS -> per-tile partial in HBM -> boundary kernel, so results
//     do not depend on launch geometry or GPU count.
#pragma once
#include <hip/hip_runtime.h>

#include "mpx_device.h"

// A product and a sum are fused only where the source says fma(): the same expression (node time, node functions, residuals) is
// evaluated by several kernels of this file, whose results are promised to be bit-identical (light and heavy passes, fused and
// separate residual passes, sharded and unsharded evaluations); left to itself the compiler fuses differently per inlining context.
#pragma clang fp contract(off)

#ifndef MPX_TABLES_IN_LDS_ABOVE
#define MPX_TABLES_IN_LDS_ABOVE 12  // degrees above this keep the D / mid-point tables in LDS
#endif
// Degrees above MPX_TABLES_STREAM_ABOVE (mpx_device.h; the host checks the value a code object was built with) do not hold the
// tables at all: 2 (P + 1)^2 doubles are 37 KB at degree 48, 130 KB at 90 (ONE workgroup per compute unit) and 1 MB at 255 -- the
// reference documents and times 1 x 100 (docs/source/notebooks/getting_started.ipynb:721-743).  There every lane STREAMS its rows
// from the TRANSPOSED tables in global memory (L2-resident; DT[j][k] = D[k][j]: the lanes of a segment hold consecutive k, so a
// wavefront's load of one j is one or two contiguous runs), eight columns at a time, the next eight requested before the stores
// of the current ones (node_body: stream_tables).

#ifndef MPX_STREAM_CH
#define MPX_STREAM_CH 4  // columns per step of the streamed-table walk (even); two steps' values are live: 8 costs 64 VGPRs and spilled
#endif

namespace mpxk {

template <int N>
struct Vec {
  double v[N > 0 ? N : 1];
  __device__ __forceinline__ double& operator[](int i) { return v[i]; }
  __device__ __forceinline__ const double& operator[](int i) const { return v[i]; }
  __device__ __forceinline__ operator double*() { return v; }
};

// The wavefront's total in EVERY lane, on the DPP path (no LDS round trips): inclusive scan by row shifts and row broadcasts, then
// lane 63.  A fixed tree.
__device__ __forceinline__ double wave_total_dpp(double x) {
  auto dpp0 = [](double v, auto ctrl, auto row_mask) {
    constexpr int C = decltype(ctrl)::value, R = decltype(row_mask)::value;
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), C, R, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), C, R, 0xf, false);
    return __hiloint2double(hi, lo);
  };
  using std::integral_constant;
  x += dpp0(x, integral_constant<int, 0x111>{}, integral_constant<int, 0xf>{});  // row_shr:1
  x += dpp0(x, integral_constant<int, 0x112>{}, integral_constant<int, 0xf>{});  // row_shr:2
  x += dpp0(x, integral_constant<int, 0x114>{}, integral_constant<int, 0xf>{});  // row_shr:4
  x += dpp0(x, integral_constant<int, 0x118>{}, integral_constant<int, 0xf>{});  // row_shr:8
  x += dpp0(x, integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{});  // row_bcast:15 into rows 1 and 3
  x += dpp0(x, integral_constant<int, 0x143>{}, integral_constant<int, 0xc>{});  // row_bcast:31 into rows 2 and 3
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 63), __builtin_amdgcn_readlane(__double2loint(x), 63));
}
// M wavefront totals at once, each by the tree of wave_total_dpp (the same bits), step by step across the M values: one total is a
// chain of six dependent DPP moves and additions with their wait states, M of them written one after the other between other work
// ran as M chains in a row (twelve per item in light_low_body: 37 of 187 us of a config-5 nlp_f pass).
template <int M>
__device__ __forceinline__ void wave_total_dpp_n(double (&x)[M]) {
  auto dpp0 = [](double v, auto ctrl, auto row_mask) {
    constexpr int C = decltype(ctrl)::value, R = decltype(row_mask)::value;
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), C, R, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), C, R, 0xf, false);
    return __hiloint2double(hi, lo);
  };
  using std::integral_constant;
  auto step = [&](auto ctrl, auto row_mask) {
    double d[M];
#pragma unroll
    for (int m = 0; m < M; ++m) d[m] = dpp0(x[m], ctrl, row_mask);
#pragma unroll
    for (int m = 0; m < M; ++m) x[m] += d[m];
  };
  step(integral_constant<int, 0x111>{}, integral_constant<int, 0xf>{});  // row_shr:1
  step(integral_constant<int, 0x112>{}, integral_constant<int, 0xf>{});  // row_shr:2
  step(integral_constant<int, 0x114>{}, integral_constant<int, 0xf>{});  // row_shr:4
  step(integral_constant<int, 0x118>{}, integral_constant<int, 0xf>{});  // row_shr:8
  step(integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{});  // row_bcast:15 into rows 1 and 3
  step(integral_constant<int, 0x143>{}, integral_constant<int, 0xc>{});  // row_bcast:31 into rows 2 and 3
#pragma unroll
  for (int m = 0; m < M; ++m)
    x[m] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x[m]), 63), __builtin_amdgcn_readlane(__double2loint(x[m]), 63));
}
#ifndef MPX_WAVE_SUM_DPP
#define MPX_WAVE_SUM_DPP 1
#endif
// Sum over the wavefront, lane 0 (at least) holds the total; a fixed tree => deterministic.  On the DPP path since round 4
// (wave_total_dpp above): the __shfl_down tree is two ds_bpermute per step and double, twelve dependent LDS round trips per sum --
// config 2 f+g+grad_f+jac_g 1021 -> 1003 us per 4096 points, config 5 142 -> 138 us per 512 in process (tools/r4_light_ab.py HEAVY=1;
// -DMPX_WAVE_SUM_DPP=0 restores the old tree, whose sums differ in the last place).
__device__ __forceinline__ double wave_sum(double v) {
  if (MPX_WAVE_SUM_DPP) return wave_total_dpp(v);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

typedef double mpx_d2 __attribute__((ext_vector_type(2)));

// Scatter NS per-lane values into a tile block.  Layout of a block (n lanes, NS slots), chosen so
// that one store instruction of a wavefront writes ONE contiguous run with 16 bytes per lane:
//     slots are interleaved in pairs:  index(q, l) = (q/2)*2n + 2l + (q&1)      for q < NS - (NS&1)
//     an unpaired last slot is plain:  index(NS-1, l) = (NS-1)*n + l
// (the COO patterns reported by mpx_pattern_* follow the same formula, mpx_layout.cpp).  Measured on
// MI355X (tools/store_bw.hip): 16 B/lane contiguous stores stream at 5.2-5.5 TB/s, 8 B/lane at
// 4.4-5.0 TB/s, nontemporal stores are slower than both.
// Addressing: uniform base (SGPRs) + ONE running 32-bit byte offset per lane.  The offsets are loop
// invariant across the batch loop; the empty asm stops the compiler from hoisting one 64-bit address
// pair per slot out of it (that cost 90+ VGPRs and more than halved the occupancy).
struct AllSlots {
  __device__ __forceinline__ bool operator()(int) const { return true; }
};

// `keep(q)`: false for slots whose value is a constant of the grid and may be skipped (opt-in
// MPX_JAC_VARIABLE_ONLY); evaluated at compile time after unrolling, a pair is written if either slot is kept.
template <int NS, int FENCE = 0, class F, class K = AllSlots>
__device__ __forceinline__ void scatter_slots(double* __restrict__ blk, int64_t n, int l, bool own, bool vec, F sv, K keep = K()) {
  if (!own) return;
  char* __restrict__ base = reinterpret_cast<char*>(blk);
  const uint32_t nb = (uint32_t)n * 8u;  // bytes per slot
  uint32_t off = (uint32_t)l * 16u;
#pragma unroll
  for (int q = 0; q + 1 < NS; q += 2) {
    if (!(keep(q) || keep(q + 1))) {
      off += 2u * nb;
      continue;
    }
    asm volatile("" : "+v"(off));
    if (vec) {
      mpx_d2 w;
      w.x = sv(q);
      w.y = sv(q + 1);
      *reinterpret_cast<mpx_d2*>(base + off) = w;
    } else {  // block not 16-byte aligned (odd-sized blocks come last, mpx_layout.cpp)
      *reinterpret_cast<double*>(base + off) = sv(q);
      *reinterpret_cast<double*>(base + off + 8u) = sv(q + 1);
    }
    off += 2u * nb;
    // FENCE > 0 (tables in LDS): keep the compiler from hoisting every table read of the block above
    // the first store -- that alone held ~60 VGPRs live at degree 30
    if (FENCE > 0 && (q / 2) % FENCE == FENCE - 1) __builtin_amdgcn_sched_barrier(0);
  }
  if ((NS & 1) && keep(NS - 1)) {
    uint32_t o1 = (uint32_t)(NS - 1) * nb + (uint32_t)l * 8u;
    asm volatile("" : "+v"(o1));
    *reinterpret_cast<double*>(base + o1) = sv(NS - 1);
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// High degrees (tables in LDS): the Jacobian block of a lane is mostly COPIES of table rows -- per state row the lane's row
// of D (minus the dynamics derivative on the diagonal), per control row its row of D or of the mid-point interpolation
// matrix: the same (P+1)-long constants for every segment.  Unrolling ~100 slots with a select per slot cost 170 VGPRs at
// degree 30 (2 wavefronts per SIMD).  Here the rows are streamed by short runtime loops instead: values come from LDS by a
// running index, leave in slot pairs (16-byte stores, same layout as scatter_slots), and the odd value a row may leave
// over is carried into the next row (compile-time parity).  ~60 VGPRs.
// ---------------------------------------------------------------------------------------------------------------------
template <bool VEC>
struct SlotStream {
  char* __restrict__ base;
  uint32_t off;    // running byte offset of the next pair for this lane
  uint32_t step;   // bytes between consecutive pairs: 2 * n * 8
  double carry;
  __device__ __forceinline__ void pair(double a, double b) {
    if constexpr (VEC) {
      mpx_d2 w;
      w.x = a;
      w.y = b;
      *reinterpret_cast<mpx_d2*>(base + off) = w;
    } else {
      *reinterpret_cast<double*>(base + off) = a;
      *reinterpret_cast<double*>(base + off + 8u) = b;
    }
    off += step;
  }
};

// LEN values val(0..LEN-1) (val takes a RUNTIME index); HAVE: a value is waiting in S.carry.  Returns nothing; whether a value
// is left in S.carry afterwards is the compile-time constant ((LEN + HAVE) & 1).
template <bool HAVE, int LEN, bool VEC, class V>
__device__ __forceinline__ void stream_run(SlotStream<VEC>& S, V val) {
  int j = 0;
  if constexpr (HAVE) {
    S.pair(S.carry, val(0));
    j = 1;
  }
  constexpr int NPAIR = (LEN - (HAVE ? 1 : 0)) / 2;
#pragma unroll 4
  for (int t = 0; t < NPAIR; ++t, j += 2) {
    const double a = val(j), b = val(j + 1);
    S.pair(a, b);
  }
  if constexpr (((LEN - (HAVE ? 1 : 0)) & 1) != 0) S.carry = val(LEN - 1);
}

// compile-time recursion over the NX state rows / NU control rows of a block (parity threads through the template)
template <int R, int NR, bool HAVE, int LEN, bool VEC, class F>
__device__ __forceinline__ void stream_rows(SlotStream<VEC>& S, F row_val) {
  if constexpr (R < NR) {
    stream_run<HAVE, LEN, VEC>(S, [&](int j) { return row_val(R, j); });
    stream_rows<R + 1, NR, (((LEN + (HAVE ? 1 : 0)) & 1) != 0), LEN, VEC>(S, row_val);
  }
}
template <int I, int NV, bool HAVE, bool VEC, class F>
__device__ __forceinline__ void stream_regs(SlotStream<VEC>& S, F reg_val) {  // NV values from registers, compile-time indices
  if constexpr (I < NV) {
    if constexpr (HAVE) {
      S.pair(S.carry, reg_val(I));
      stream_regs<I + 1, NV, false, VEC>(S, reg_val);
    } else if constexpr (I + 1 < NV) {
      S.pair(reg_val(I), reg_val(I + 1));
      stream_regs<I + 2, NV, false, VEC>(S, reg_val);
    } else {
      S.carry = reg_val(I);
    }
  }
}

// Static LDS keyed by size and tag ONLY: node_body<PH, ...> of different phases (or modes) inlined into one kernel -- the all-phases
// kernels below, the resident kernel -- get the SAME block instead of one per instantiation (the phases of an OCP have the same
// nx / nu, mpopt.py:3407-3409, so their blocks have the same size; a workgroup runs one phase).
template <int N_DOUBLES, int TAG>
__device__ __forceinline__ double* mpx_lds() {
  __shared__ double blk[N_DOUBLES > 0 ? N_DOUBLES : 1];
  return blk;
}

template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for_n(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for_n<N, I + 1>(f);
  }
}

template <int PH, int P, int MODE>
__device__ __forceinline__ void node_body(const MpxNodeArgs& A, const int res_bx = -1, const int item_bx = -1, const unsigned item_by = 0) {
  // res_bx >= 0: called by the resident kernel for tile res_bx of the bucket;  item_bx >= 0: called by the all-phases kernel
  // (node_all below) with the workgroup's tile of THIS phase's range and its chunk of evaluation points already worked out
  using G = mpxgen::Phase<PH>;
  constexpr int NX = G::NX, NU = G::NU, NA = G::NA, NC = G::NC;
  constexpr int P1 = P + 1;
  constexpr int SEGS = MPX_TILE / P;
  constexpr int SLOTS = SEGS * P1;  // (the hess_l pass stages X / U too when it evaluates the mid-point residuals, MPX_MID_RESID)
  constexpr int NRED = (MODE == MPX_MODE_FG) ? 1 : (MODE == MPX_MODE_FGJ ? G::NRED : G::NHC);
  constexpr int NRED1 = NRED > 0 ? NRED : 1;
  // Jacobian slots of a node: D-blocks of the defect rows, variable entries, D-blocks of the
  // control-slope rows, interpolation blocks of the mid-point rows
  constexpr int NS_MAIN = NX * P1 + G::NJV + (G::DIFF_U ? NU * P1 : 0);
  constexpr int NS_MID = G::MIDU ? NU * P1 : 0;
  double(*sXU)[NX + NU][SLOTS] = reinterpret_cast<double(*)[NX + NU][SLOTS]>(mpx_lds<2 * (NX + NU) * SLOTS, 0>());  // [2][NX + NU][SLOTS]
  constexpr int NRED_LDS = 12;  // (one block for every mode and phase: NRED differs between them, the block does not)
  double(*sRed)[MPX_TILE / 64][NRED1] = reinterpret_cast<double(*)[MPX_TILE / 64][NRED1]>(mpx_lds<2 * (MPX_TILE / 64) * (NRED1 > NRED_LDS ? NRED1 : NRED_LDS), 1>());
  extern __shared__ double sAbs[];  // absorbing tiles: [row slot][abs_cap] g / grad_f values of the tile's node span

  // Workgroup -> (tile, batch chunk).  Hardware hands consecutive workgroup ids to the 8 XCDs round-robin;
  // here every XCD (id = linear id mod 8) walks a CONTIGUOUS range of (chunk, tile) items, so each XCD's L2
  // streams one contiguous eighth of the output arrays.  Measured on MI355X (config 2, B=4096): +3.4 % over the
  // natural mapping at equal chunk size; chunk-fastest order: -6 %; rotating tiles with the chunk: +-0; starting
  // every XCD at a different phase of its range: +-0.
  // (MPX_MAP_NATURAL restores blockIdx.x = tile, blockIdx.y = chunk for A/B runs.)
#if defined(MPX_MAP_NATURAL)
  const unsigned bx_ = item_bx >= 0 ? (unsigned)item_bx : res_bx >= 0 ? (unsigned)res_bx : blockIdx.x, by_ = item_bx >= 0 ? item_by : res_bx >= 0 ? 0u : blockIdx.y;
#else
  const unsigned lin_ = blockIdx.y * gridDim.x + blockIdx.x, tot_ = gridDim.x * gridDim.y;
  const unsigned xcd_ = lin_ % 8, q_ = tot_ / 8, r_ = tot_ % 8;  // XCD j owns q_ + (j < r_) items
  const unsigned item_ = xcd_ * q_ + (xcd_ < r_ ? xcd_ : r_) + lin_ / 8;  // bijection [0, tot) -> [0, tot)
  const unsigned bx_ = item_bx >= 0 ? (unsigned)item_bx : res_bx >= 0 ? (unsigned)res_bx : item_ % gridDim.x;
  const unsigned by_ = item_bx >= 0 ? item_by : res_bx >= 0 ? 0u : item_ / gridDim.x;
#endif
#ifdef MPX_NO_REGULAR
  const bool regular = false;
#else
  const bool regular = A.regular != 0;
#endif
  // The tile descriptor: a table load -- or, for regular buckets, arithmetic, so that the loads of z (and of the lane's table
  // rows) are the FIRST memory operations of the workgroup instead of the third level of a dependent chain.  Measured on MI355X
  // (one evaluation point per workgroup, same-process A/B): headline kernel 914 -> 863 us, hypersensitive 4000x3 1003 -> 908 us.
  MpxTile T{};  // (span_* / f_* stay zero for regular buckets: they never absorb)
  if (regular) {
    const int t = A.tile_first + (int)bx_ - A.reg_first_tile;  // index in the bucket
    const int w = t == 0 ? 0 : (t == A.reg_last ? 2 : 1);
    const int64_t step = t > 1 && w == 1 ? (int64_t)(t - 1) : 0;
    T.m0 = t == 0 ? 0 : 1 + (t - 1) * A.reg_lanes;
    T.n = t == 0 ? P + 1 : (t == A.reg_last ? A.reg_last_lanes : A.reg_lanes);
    T.n_own = t == 0 ? 1 : T.n;
    T.node0 = t == 0;
    T.tile_id = A.reg_first_tile + t;
    T.seg0 = t == 0 ? 0 : (t - 1) * (A.reg_lanes / P);
    T.jac_base = A.reg_jac_base[w] + step * A.reg_jac_size;
    T.hess_base = A.reg_hess_base[w] + step * A.reg_hess_size;
    T.g_base = A.reg_g_base[w] + step * A.reg_g_size;
  } else {
    T = A.tiles[A.tile_first + bx_];
  }
  const int l = threadIdx.x;
  const bool act = l < T.n;      // stages a node in LDS
  const bool own = l < T.n_own;  // owns output rows / entries
  const int m = T.m0 + (act ? l : 0);
  // node -> (phase node index, segment, point).  When every segment of the phase has this degree the bucket's node list is the
  // phase's node list (node 0, then points 1..P of every segment): plain arithmetic instead of two table loads that the loads of z
  // would have to wait for (one level less in the dependent-load chain at the head of every workgroup).
  const int i = regular ? m : A.node_i[m];
  const int sk = regular ? (m > 0 ? ((((m - 1) / P) << 8) | ((m - 1) % P + 1)) : 0) : A.node_sk[m];
  const int s = sk >> 8, k = sk & 255;
  // LDS slot of the lane's segment: tiles hold whole segments of one degree, P lanes each
  const int base = ((k == 0) ? 0 : (l - T.node0) / P) * P1;
  const bool halo = act && (k == 1);
  const int lane = l & 63, wave = l >> 6;
  const int N = A.N;
  const int64_t n = T.n_own;
#ifdef MPX_NO_VEC  // A/B switch for tools/ab.py: force 8-byte stores
  const bool vec = false, vech = false;
#else
  const bool vec = (T.jac_base & 1) == 0, vech = (T.hess_base & 1) == 0;
#endif

  // The lane's row of D and of the mid-point interpolation matrix, resident for the batch loop:
  // in VGPRs for low degrees; for high degrees (2*(P+1) doubles per lane would cost > 100 VGPRs and
  // halve the occupancy) the two tables are staged once per workgroup in LDS and read per use.  Rows of
  // different points are 2*(P+1) dwords apart: for odd P+1 the b64 reads of a wavefront hit distinct
  // banks, equal points broadcast.
  // (the hess_l kernels use the tables for the mid-point residuals only, and those exist for register tables only: above that
  // degree a hess_l kernel neither stages nor reads a table)
  constexpr bool TAB_REG = (P <= MPX_TABLES_IN_LDS_ABOVE);
  constexpr bool TAB_GLB = !TAB_REG && MODE != MPX_MODE_HESS && (P > MPX_TABLES_STREAM_ABOVE);
  constexpr bool TAB_LDS = !TAB_REG && MODE != MPX_MODE_HESS && !TAB_GLB;
  constexpr int NREG = TAB_REG ? P1 : 1;
  double* const sD = mpx_lds<(TAB_LDS ? P1 * P1 : 1), 2>();
  double* const sC = mpx_lds<(TAB_LDS ? P * P1 : 1), 3>();
  double Drow_[NREG], Crow_[NREG], Dmrow_[NREG];
  const int drow = k * P1, crow = (k >= 1 ? k - 1 : 0) * P1;
  // MPX_MID_RESID (hess_l pass only, degrees with the tables in registers): residual of the dynamics at the mid-point before node k
  const bool midres = MODE == MPX_MODE_HESS && TAB_REG && A.io.mid_resid != nullptr;
  if constexpr (TAB_LDS) {
    for (int e = l; e < P1 * P1; e += MPX_TILE) sD[e] = A.Dmat[e];
    for (int e = l; e < P * P1; e += MPX_TILE) sC[e] = A.Cmid[e];
    __syncthreads();
  } else if constexpr (TAB_REG) {
    // hess_l passes, tables of at most 64 entries: ONE load per table, lane and wavefront (lane e holds entry e), the lane's rows are
    // fetched from the lanes that hold them (ds_bpermute, no memory) instead of 2-3 (P + 1) loads per lane in front of every
    // workgroup -- an ablation without the table loads gains 3-8 % on these short workgroups (profiles/r3_soak.md).  Same box,
    // -DMPX_NO_TAB_SHUFFLE against this: hess_l config 2 296 -> 268 us, config 5 680 -> 639 us, config 4 260 -> 250 us; the
    // loop's hess_l + residual pass 107.6 -> 104.0 us.  A loss for the first-order kernels (their 18 fetches per lane cost more
    // than they save: 1002 -> 1032 us), which keep the loads.
#ifndef MPX_NO_TAB_SHUFFLE
    if constexpr (MODE == MPX_MODE_HESS && P1 * P1 <= 64) {
      // (the hess_l kernels use the tables for the mid-point residuals only: a pass without MPX_MID_RESID fetches none)
#pragma unroll
      for (int j = 0; j < P1; ++j) Drow_[j] = 0.0, Crow_[j] = 0.0, Dmrow_[j] = 0.0;
      if (midres) {
        const int ln = l & 63;
        const double tC = A.Cmid[ln < P * P1 ? ln : 0], tM = A.Dmid[ln < P * P1 ? ln : 0];
#pragma unroll
        for (int j = 0; j < P1; ++j) {
          const double c = __shfl(tC, crow + j, 64), m = __shfl(tM, crow + j, 64);
          Crow_[j] = (k >= 1) ? c : 0.0;
          Dmrow_[j] = (k >= 1) ? m : 0.0;
        }
      }
    } else
#endif
    {
#pragma unroll
    for (int j = 0; j < P1; ++j) {
      Drow_[j] = A.Dmat[drow + j];
      Crow_[j] = (k >= 1) ? A.Cmid[crow + j] : 0.0;
      Dmrow_[j] = (midres && k >= 1) ? A.Dmid[crow + j] : 0.0;
    }
    }
  }
  const double tkm = (midres && k >= 1) ? A.tkm[k - 1] : 0.0;
  auto Drow = [&](int j) -> double {
    if constexpr (TAB_LDS) return sD[drow + j];
    else return Drow_[j];
  };
  auto Crow = [&](int j) -> double {
    if constexpr (TAB_LDS) return sC[crow + j];
    else return Crow_[j];
  };
  const double tkk = A.tk[k];
  const double Wn = A.Wnode[i];

  const MpxIO& io = A.io;
  // hess_l passes: ONE evaluation point per workgroup as a compile-time fact (the host launches them so).  Without the state of the
  // software-pipelined batch loop the kernel needs 52 instead of 85 VGPRs (degree 5) and runs 3-5 % faster than the loop at its
  // best setting (config 2: 279 -> 265 us, config 5: 643 -> 624 us, the config-5 loop's pass 105.7 -> 102.8 us;
  // tools/r3_bpb1_ab.py).  Not for the first-order kernels: at twice the occupancy they are SLOWER (948 -> 1236 us at config 2 --
  // more workgroups per compute unit writing at once is what one point per workgroup was chosen against).
#ifndef MPX_HESS_ONE_POINT
#define MPX_HESS_ONE_POINT 1
#endif
#ifdef MPX_ABL_BPB1  // (experiment switch of tools/r3_bpb1_ab.py: every mode)
  constexpr bool ONE_POINT = true;
#else
  constexpr bool ONE_POINT = MODE == MPX_MODE_HESS && MPX_HESS_ONE_POINT;
#endif
  const int b0 = io.b_first + (ONE_POINT ? by_ : by_ * io.b_per_block);
  const int b1 = ONE_POINT ? (b0 + 1 < io.B ? b0 + 1 : io.B) : ((b0 + io.b_per_block < io.B) ? b0 + io.b_per_block : io.B);
  const int64_t tslot = (int64_t)T.tile_id * io.nred;
  // Absorbing tile (mixed-degree phases, MpxNodeArgs::abs_cap): row slots as in the packed staging block -- defect, path, DU, mU
  // rows, then the grad_f entries of the node.
  constexpr int SG_C = NX, SG_DU = NX + NC, SG_MU = SG_DU + (G::DIFF_U ? NU : 0), SG_Q = SG_MU + (G::MIDU ? NU : 0), NSG = SG_Q + NX + NU;
  const bool absorb = MODE != MPX_MODE_HESS && A.abs_cap > 0 && T.span_len > 0 && io.gtmp != nullptr;
  const int cap = A.abs_cap;
  const int ia = i - T.span_lo;  // the lane's node in the span
  int fpos = -1, fn = 0;         // the foreign node this lane fetches from the staging block
  int64_t fstage = 0;
  if (absorb && l < T.f_count) fpos = A.abs_fpos[T.f_first + l], fstage = A.abs_fstage[T.f_first + l], fn = A.abs_fn[T.f_first + l];

  // One evaluation point's inputs of this lane.  The batch loop is software pipelined: the loads of
  // point b+1 are issued BEFORE the stores of point b.  vmcnt retires in order, so a wait for loads
  // that were issued after a burst of stores would drain the whole store queue every iteration
  // (measured: the un-pipelined loop was 10 % slower and preferred tiny batch chunks).
  struct In {
    Vec<NX> Xs;
    Vec<NU> Us;
    Vec<NX + NU> Hl;  // first node of the segment (only lanes with k == 1 use it)
    Vec<NA> As;
    double t0v, tfv, ws, wc;
  };
  auto load_point = [&](int b, In& q) {
#ifdef MPX_ABL_SAME_Z  // ablation: every evaluation point reads the first point's z (no HBM reads to speak of)
    const double* __restrict__ zb = io.z + A.z_off;
#else
    const double* __restrict__ zb = io.z + (int64_t)b * io.z_stride + A.z_off;
#endif
    // (a streaming hint on these loads, __builtin_nontemporal_load, measured 7 % slower)
#pragma unroll
    for (int a = 0; a < NX; ++a) q.Xs[a] = (zb + (int64_t)a * N)[i];
#pragma unroll
    for (int c = 0; c < NU; ++c) q.Us[c] = (zb + (int64_t)(NX + c) * N)[i];
    if (MODE != MPX_MODE_HESS || midres) {
      if (halo) {  // first node of the segment belongs to the previous segment (mpopt.py:190-195)
#pragma unroll
        for (int a = 0; a < NX + NU; ++a) q.Hl[a] = (zb + (int64_t)a * N)[i - 1];
      }
    }
    const double* __restrict__ zt = zb + (int64_t)(NX + NU) * N;
    q.t0v = zt[0];
    q.tfv = zt[1];
#pragma unroll
    for (int c = 0; c < NA; ++c) q.As[c] = zt[2 + c];
    const int64_t woff = (int64_t)b * io.w_stride + A.seg_off + s;
    q.ws = io.w[woff];
    q.wc = io.wcum[woff];
  };
  // Absorbing tile: after the own lanes put their values in the span buffer, the foreign nodes' staged values join them and the
  // whole workgroup stores every row of the span as one contiguous run (full cache lines except at the two ends).  The barrier
  // waits for LDS only: a __syncthreads() here would also drain the store queue of the previous point's Jacobian block.
  double fv[MODE == MPX_MODE_HESS ? 1 : NSG];  // staged values of the lane's foreign node (loaded at the top of the iteration)
  int b_cur = 0;
  auto span_store = [&]() {
    constexpr int NSL = (MODE == MPX_MODE_FGJ) ? NSG : SG_Q;
    if (fpos >= 0) {
#pragma unroll
      for (int sl = 0; sl < NSL; ++sl) sAbs[sl * cap + fpos] = fv[sl];
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const int len = T.span_len, lo = T.span_lo;
    auto put_row = [&](int sl, double* __restrict__ row, int e0) {  // row[e] <- slot sl, e = e0 .. len - 1
      for (int e = e0 + l; e < len; e += MPX_TILE) row[e] = sAbs[sl * cap + e];
    };
    if (io.g) {
      double* __restrict__ gb = io.g + (int64_t)b_cur * io.g_stride;
#pragma unroll
      for (int a = 0; a < NX; ++a) put_row(a, gb + (A.g_off_F + (int64_t)a * N) + lo, 0);
#pragma unroll
      for (int j = 0; j < NC; ++j) put_row(SG_C + j, gb + (A.g_off_C + (int64_t)j * N) + lo, 0);
      if constexpr (G::DIFF_U) {
#pragma unroll
        for (int c = 0; c < NU; ++c) put_row(SG_DU + c, gb + (A.g_off_DU + (int64_t)c * N) + lo, 0);
      }
      if constexpr (G::MIDU) {  // the row of node i is i - 1; node 0 of the phase has none
#pragma unroll
        for (int c = 0; c < NU; ++c) put_row(SG_MU + c, gb + (A.g_off_mU + (int64_t)c * (N - 1)) + (lo - 1), lo == 0 ? 1 : 0);
      }
    }
    if constexpr (MODE == MPX_MODE_FGJ) {
      if (io.grad) {
        double* __restrict__ qb = io.grad + (int64_t)b_cur * io.grad_stride + A.z_off;
#pragma unroll
        for (int a = 0; a < NX + NU; ++a) put_row(SG_Q + a, qb + (int64_t)a * N + lo, 0);
      }
    }
  };
  In cur, nxt;
  if (b0 < b1) load_point(b0, cur);
  int it = 0;
  for (int b = b0; b < b1; ++b, ++it) {
    Vec<NX>& Xs = cur.Xs;
    Vec<NU>& Us = cur.Us;
    Vec<NA>& As = cur.As;
    const double t0v = cur.t0v, tfv = cur.tfv;
    const double kap = cur.ws * A.inv_dtau;    // h = (tf - t0) * kap            (mpopt.py:184)
    const double th = cur.wc + cur.ws * tkk;   // t = t0 + (tf - t0) * th        (mpopt.py:192, 198)
    const int buf = it & 1;
    if (MODE != MPX_MODE_HESS || midres) {
      if (act) {
#pragma unroll
        for (int a = 0; a < NX; ++a) sXU[buf][a][base + k] = Xs[a];
#pragma unroll
        for (int c = 0; c < NU; ++c) sXU[buf][NX + c][base + k] = Us[c];
        if (halo) {
#pragma unroll
          for (int a = 0; a < NX + NU; ++a) sXU[buf][a][base] = cur.Hl[a];
        }
      }
    }
    if (b + 1 < b1) load_point(b + 1, nxt);
    b_cur = b;
    if constexpr (MODE != MPX_MODE_HESS) {
      if (fpos >= 0) {  // staged values of the lane's foreign node: in flight during the node's own work
        const double* __restrict__ gt = io.gtmp + (int64_t)b * io.gtmp_stride + fstage;
#pragma unroll
        for (int sl = 0; sl < (MODE == MPX_MODE_FGJ ? NSG : SG_Q); ++sl) fv[sl] = gt[(int64_t)sl * fn];
      }
    }
#ifndef MPX_ABL_NO_BARRIER
    __syncthreads();
#endif
    if (it > 0 && l < NRED) {  // publish the previous point's tile sums
      double v = 0;
#pragma unroll
      for (int w = 0; w < MPX_TILE / 64; ++w) v += sRed[buf ^ 1][w][l];
      io.partial[((int64_t)(b - 1) * io.n_tiles_total) * io.nred + tslot + l] = v;
    }

    Vec<NRED> red;
    if constexpr (MODE == MPX_MODE_HESS) {
      Vec<NX> lF;
      Vec<NC> lC;
      const double* __restrict__ lb = io.lam_g + (int64_t)b * io.lam_stride;
#pragma unroll
      for (int a = 0; a < NX; ++a) lF[a] = (lb + (A.g_off_F + (int64_t)a * N))[i];
#pragma unroll
      for (int j = 0; j < NC; ++j) lC[j] = (lb + (A.g_off_C + (int64_t)j * N))[i];
      Vec<G::NHN> hn;
      G::hess(Xs, Us, t0v, tfv, As, kap, th, Wn, io.sigma[b], lF, lC, hn, red);
      scatter_slots<G::NHN>(io.hess + (int64_t)b * io.hess_stride + T.hess_base, n, l, own, vech, [&](int q) { return hn[q]; });
      if constexpr (TAB_REG) {
      if (midres && own && k >= 1) {
        // D_mid.X - h Sx dyn(I_mid.X, I_mid.U, t_mid, a) at the mid-point between nodes i - 1 and i: the same fma chains, in the
        // same order, as mpx_resid_<ph>_<deg> runs for that target point (mpopt.py:1466-1481), from the segment's X / U in LDS
        Vec<NX> Xi, DXi, fxm;
        Vec<NU> Ui;
        Vec<NC> ccm;
#pragma unroll
        for (int a = 0; a < NX; ++a) {
          double v = 0, d = 0;
#pragma unroll
          for (int j = 0; j < P1; ++j) {
            const double x = sXU[buf][a][base + j];
            v = fma(Crow_[j], x, v);
            d = fma(Dmrow_[j], x, d);
          }
          Xi[a] = v, DXi[a] = d;
        }
#pragma unroll
        for (int c = 0; c < NU; ++c) {
          double v = 0;
#pragma unroll
          for (int j = 0; j < P1; ++j) v = fma(Crow_[j], sXU[buf][NX + c][base + j], v);
          Ui[c] = v;
        }
        double qWm;
#ifdef MPX_ABL_MID_NOCOMPUTE  // ablation (tools/r3_midres_ab.py): the stores without the interpolation and the dynamics
#pragma unroll
        for (int a = 0; a < NX; ++a) fxm[a] = Xs[a], DXi[a] = 0;
#else
        G::fg(Xi, Ui, t0v, tfv, As, kap, cur.wc + cur.ws * tkm, 0.0, fxm, ccm, qWm);
#endif
        double* __restrict__ rb = io.mid_resid + (int64_t)b * io.mid_stride + ((int64_t)A.phase * (N - 1) + (i - 1)) * NX;
#pragma unroll
        for (int a = 0; a < NX; ++a)
#ifdef MPX_ABL_MID_NOSTORE  // ablation: the arithmetic without the stores
          if (DXi[a] - fxm[a] == 1.2345e300)
#endif
            rb[a] = DXi[a] - fxm[a];
      }
      }
    } else {
      Vec<NX> fx;
      Vec<NC> cc;
      Vec<NX> dd;
      Vec<G::NJV> jv;
      Vec<NX + NU> gn;
      if constexpr (MODE == MPX_MODE_FGJ) {
        G::fgj(Xs, Us, t0v, tfv, As, kap, th, Wn, fx, cc, dd, jv, gn, red);
      } else {
        G::fg(Xs, Us, t0v, tfv, As, kap, th, Wn, fx, cc, red[0]);
      }
      // Streamed tables (TAB_GLB): ONE walk over the lane's rows of D and of the mid-point matrix serves the contractions
      // (sequential fma chains over j, the bits of the other two table modes) and the constant rows of the lane's Jacobian block.
      // Slot q of a lane lives at pair q / 2 (scatter_slots' layout); a row that starts on an odd slot opens and closes every
      // chunk with an 8-byte half of a pair, the rest are 16-byte stores.  Addresses: uniform pointer (SGPRs) + the lane's byte
      // offset, for the table loads and for the stores.
      double aX[NX], aDU[G::DIFF_U ? NU : 1], aMU[G::MIDU ? NU : 1];
      if constexpr (TAB_GLB) {
#pragma unroll
        for (int a = 0; a < NX; ++a) aX[a] = 0;
#pragma unroll
        for (int c = 0; c < (G::DIFF_U ? NU : 1); ++c) aDU[c] = 0;
#pragma unroll
        for (int c = 0; c < (G::MIDU ? NU : 1); ++c) aMU[c] = 0;
        const bool want_g = own && io.g != nullptr;
        const bool want_j = MODE == MPX_MODE_FGJ && own && io.jac != nullptr;
        using std::integral_constant;
        // NSR: slots of a lane of this tile, VEC2: 16-byte stores (compile time: three instantiations below)
        auto walk = [&](auto nsr_c, auto vec_c) {
          constexpr int NSR = decltype(nsr_c)::value;
          constexpr bool VEC2 = decltype(vec_c)::value;
          constexpr int LASTQ = (NSR & 1) ? NSR - 1 : -1;  // an unpaired last slot is plain (scatter_slots)
          const bool vo = io.jac_variable_only != 0;
          const bool mid = G::MIDU && k >= 1;  // the lane has mid-point rows (every owning lane but node 0 of the phase)
          char* const jbase = reinterpret_cast<char*>(io.jac + (int64_t)b * io.jac_stride + T.jac_base);
          const uint32_t nb = (uint32_t)n * 8u, l16 = (uint32_t)l * 16u, l8 = (uint32_t)l * 8u;
          // slot q0 + j (q0 compile time, j a uniform runtime value or 0)
          auto put1 = [&](auto q0_c, auto maybe_last_c, uint32_t j, double v) {
            constexpr int Q0 = decltype(q0_c)::value;
            const uint32_t q = (uint32_t)Q0 + j;
            if (decltype(maybe_last_c)::value && LASTQ >= 0 && q == (uint32_t)LASTQ) {
              uint32_t o = l8;
              asm volatile("" : "+v"(o));
              *reinterpret_cast<double*>(jbase + (size_t)q * nb + o) = v;
            } else {
              uint32_t o = l16;
              asm volatile("" : "+v"(o));  // (pinned: uniform pointer in SGPRs + one 32-bit lane offset, no 64-bit address per store)
              *reinterpret_cast<double*>(jbase + ((size_t)(q >> 1) * 2u * nb + (q & 1u) * 8u) + o) = v;
            }
          };
          auto put2 = [&](auto q0_c, uint32_t j, double v0, double v1) {  // Q0 + j even, both slots exist
            char* const pu = jbase + (size_t)(((uint32_t)decltype(q0_c)::value + j) >> 1) * 2u * nb;
            uint32_t o = l16;
            asm volatile("" : "+v"(o));
            if constexpr (VEC2) {
              mpx_d2 w;
              w.x = v0, w.y = v1;
              *reinterpret_cast<mpx_d2*>(pu + o) = w;
            } else {
              *reinterpret_cast<double*>(pu + o) = v0;
              *reinterpret_cast<double*>(pu + 8 + o) = v1;
            }
          };
          // val(0 .. CNT - 1) to the slots Q0 + j ..; rows start at compile-time slots and chunks at even j, so the parity of the
          // first slot is a compile-time fact.  ML: the values may include the lane's last slot
          auto emit = [&](auto q0_c, auto cnt_c, auto ml_c, uint32_t j, auto val) {
            constexpr int Q0 = decltype(q0_c)::value, PAR = Q0 & 1, CNT = decltype(cnt_c)::value;
            if constexpr (PAR == 1 && CNT > 0) put1(q0_c, ml_c, j, val(0));
#pragma unroll
            for (int c = PAR; c + 1 < CNT; c += 2) put2(q0_c, j + (uint32_t)c, val(c), val(c + 1));
            if constexpr (CNT > PAR && ((CNT - PAR) & 1) != 0) put1(q0_c, ml_c, j + (uint32_t)(CNT - 1), val(CNT - 1));
          };
          constexpr int CH = MPX_STREAM_CH, NFULL = P1 / CH, REM = P1 % CH;
          static_assert(CH % 2 == 0, "chunks start at even columns");
          const uint32_t kd8 = 8u * (uint32_t)k, kc8 = 8u * (uint32_t)(k >= 1 ? k - 1 : 0);
          auto fetch = [&](int j0, double (&d)[CH], double (&cm)[CH]) {
            // (unconditional, clamped: a guarded prefetch meets the old values in phi nodes and is waited for at once)
#pragma unroll
            for (int c = 0; c < CH; ++c) {
              const int j = (j0 + c < P1) ? j0 + c : P;
              uint32_t od = kd8, oc = kc8;
              asm volatile("" : "+v"(od), "+v"(oc));
              d[c] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(A.Dmat + j * P1) + od);  // TRANSPOSED tables: DT[j][k] = D[k][j]
              if constexpr (G::MIDU) cm[c] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(A.Cmid + j * P) + oc);  // CT[j][k - 1] = C_mid[k - 1][j]
              else cm[c] = 0.0;
            }
          };
          auto work = [&](auto cnt_c, auto ml_c, int j0, const double (&d)[CH], const double (&cm)[CH]) {
            constexpr int CNT = decltype(cnt_c)::value;
            if (want_g) {
#pragma unroll
              for (int c = 0; c < CNT; ++c) {
#pragma unroll
                for (int a = 0; a < NX; ++a) aX[a] = fma(d[c], sXU[buf][a][base + j0 + c], aX[a]);
                if constexpr (G::DIFF_U || G::MIDU) {
#pragma unroll
                  for (int cu = 0; cu < NU; ++cu) {
                    const double u = sXU[buf][NX + cu][base + j0 + c];
                    if constexpr (G::DIFF_U) aDU[cu] = fma(d[c], u, aDU[cu]);
                    if constexpr (G::MIDU) aMU[cu] = fma(cm[c], u, aMU[cu]);
                  }
                }
              }
            }
            if constexpr (MODE == MPX_MODE_FGJ) {
              if (want_j) {
                static_for_n<NX>([&](auto a_c) {
                  constexpr int a = decltype(a_c)::value;
                  if (!vo || G::DD_VARIABLE[a])
                    emit(integral_constant<int, a * P1>{}, cnt_c, ml_c, (uint32_t)j0, [&](int c) { return (j0 + c == k) ? d[c] - dd[a] : d[c]; });
                });
                if (!vo) {
                  if constexpr (G::DIFF_U)
                    static_for_n<NU>([&](auto c_c) {
                      emit(integral_constant<int, NX * P1 + G::NJV + decltype(c_c)::value * P1>{}, cnt_c, ml_c, (uint32_t)j0, [&](int c) { return d[c]; });
                    });
                  if constexpr (NS_MID > 0 && NSR > NS_MAIN) {
                    if (mid)
                      static_for_n<NU>([&](auto c_c) {
                        emit(integral_constant<int, NS_MAIN + decltype(c_c)::value * P1>{}, cnt_c, ml_c, (uint32_t)j0, [&](int c) { return cm[c]; });
                      });
                  }
                }
              }
            }
          };
          if constexpr (MODE == MPX_MODE_FGJ && G::NJV > 0) {
            if (want_j) emit(integral_constant<int, NX * P1>{}, integral_constant<int, G::NJV>{}, std::true_type{}, 0u, [&](int q) { return jv[q]; });
          }
          double d0[CH], c0[CH], d1[CH], c1[CH];
          fetch(0, d0, c0);
#pragma unroll 2
          for (int ci = 0; ci < NFULL; ++ci) {
            fetch((ci + 1) * CH, d1, c1);  // the next chunk is requested before the stores of this one (vmcnt retires in order)
            // (column P, and with it possibly the lane's last slot, lies in a full chunk only when P + 1 is a multiple of CH)
            work(integral_constant<int, CH>{}, integral_constant<bool, REM == 0>{}, ci * CH, d0, c0);
#pragma unroll
            for (int c = 0; c < CH; ++c) d0[c] = d1[c], c0[c] = c1[c];
          }
          if constexpr (REM > 0) work(integral_constant<int, REM>{}, std::true_type{}, NFULL * CH, d0, c0);
        };
        if (want_g || want_j) {
          if (T.node0) walk(integral_constant<int, NS_MAIN>{}, std::false_type{});
          else if (vec) walk(integral_constant<int, NS_MAIN + NS_MID>{}, std::true_type{});
          else walk(integral_constant<int, NS_MAIN + NS_MID>{}, std::false_type{});
        }
      }
#ifdef MPX_ABL_NO_G
      if (false) {
#else
      if (own && io.g) {
#endif
        double* __restrict__ gb = io.g + (int64_t)b * io.g_stride;
        // destination of a g value: its row, or (mixed-degree phases) slot `sl` of the tile's packed block
        double* __restrict__ gp = io.gtmp ? io.gtmp + (int64_t)b * io.gtmp_stride + T.g_base + l : nullptr;
        // (absorbing tiles: slot `sl` of the span buffer in LDS)
        auto gput = [&](int sl, double* row, double v) {
          if (absorb)
            sAbs[sl * cap + ia] = v;
          else
            *(gp ? gp + (int64_t)sl * n : row) = v;
        };
#pragma unroll
        for (int a = 0; a < NX; ++a) {  // defect  F = D.X - h*Sx*dyn      (mpopt.py:227-232)
          double acc = 0;
          if constexpr (TAB_GLB) acc = aX[a];
          else {
#pragma unroll(TAB_LDS ? 4 : P1)
            for (int j = 0; j < P1; ++j) acc = fma(Drow(j), sXU[buf][a][base + j], acc);
          }
          gput(a, gb + (A.g_off_F + (int64_t)a * N) + i, acc - fx[a]);
        }
#pragma unroll
        for (int j = 0; j < NC; ++j) gput(NX + j, gb + (A.g_off_C + (int64_t)j * N) + i, cc[j]);  // mpopt.py:204, 255
        if constexpr (G::DIFF_U) {  // DU = D.U                              (mpopt.py:315-324)
#pragma unroll
          for (int c = 0; c < NU; ++c) {
            double acc = 0;
            if constexpr (TAB_GLB) acc = aDU[c];
            else {
#pragma unroll(TAB_LDS ? 4 : P1)
              for (int j = 0; j < P1; ++j) acc = fma(Drow(j), sXU[buf][NX + c][base + j], acc);
            }
            gput(NX + NC + c, gb + (A.g_off_DU + (int64_t)c * N) + i, acc);
          }
        }
        if constexpr (G::MIDU) {  // control at the mid-points of the nodes  (mpopt.py:350-369)
          if (k >= 1) {
#pragma unroll
            for (int c = 0; c < NU; ++c) {
              double acc = 0;
              if constexpr (TAB_GLB) acc = aMU[c];
              else {
#pragma unroll(TAB_LDS ? 4 : P1)
                for (int j = 0; j < P1; ++j) acc = fma(Crow(j), sXU[buf][NX + c][base + j], acc);
              }
              gput(NX + NC + (G::DIFF_U ? NU : 0) + c, gb + (A.g_off_mU + (int64_t)c * (N - 1)) + (i - 1), acc);
            }
          }
        }
      }
      if constexpr (MODE == MPX_MODE_FG) {
        if (absorb) span_store();
      }
      if constexpr (MODE == MPX_MODE_FGJ) {
#ifdef MPX_ABL_NO_G
        if (false) {
#else
        if (own && io.grad) {
#endif
          double* __restrict__ qb = io.grad + (int64_t)b * io.grad_stride + A.z_off;
          constexpr int SQ = NX + NC + (G::DIFF_U ? NU : 0) + (G::MIDU ? NU : 0);  // first grad_f slot of the packed block
          double* __restrict__ qp = io.gtmp ? io.gtmp + (int64_t)b * io.gtmp_stride + T.g_base + l : nullptr;
#pragma unroll
          for (int a = 0; a < NX + NU; ++a) {
            if (absorb)
              sAbs[(SQ + a) * cap + ia] = gn[a];
            else
              *(qp ? qp + (int64_t)(SQ + a) * n : qb + (int64_t)a * N + i) = gn[a];
          }
        }
        if (absorb) span_store();
#ifdef MPX_ABL_NO_JAC
        if (false) {
#else
        if (io.jac) {
#endif
          double* __restrict__ jb = io.jac + (int64_t)b * io.jac_stride + T.jac_base;
          // slot q -> value, evaluated lazily (q is a compile-time constant after unrolling).  The
          // table rows are loop invariant; the empty asm keeps the compiler from hoisting every
          // lane-exchanged pair out of the batch loop (registers are worth more than ~100 VALU ops
          // next to the stores).
          auto opaque = [](double v) {
            asm volatile("" : "+v"(v));
            return v;
          };
          auto sv = [&](int q) -> double {
            if (q < NX * P1) {
              const int a = q / P1, j = q % P1;
              const double d = opaque(Drow(j));
              return (j == k) ? d - dd[a] : d;
            }
            q -= NX * P1;
            if (q < G::NJV) return jv[q];
            q -= G::NJV;
            if (G::DIFF_U) {
              if (q < NU * P1) return opaque(Drow(q % P1));
              q -= NU * P1;
            }
            return opaque(Crow(q % P1));
          };
          auto variable = [](int q) -> bool {  // slot depends on (z, p)?
            if (q < NX * P1) return G::DD_VARIABLE[q / P1];
            return q < NX * P1 + G::NJV;
          };
          if constexpr (TAB_GLB) {
            // (written by the walk over the streamed tables above)
          } else if (io.jac_variable_only) {
            if (T.node0)
              scatter_slots<NS_MAIN, 0>(jb, n, l, own, false, sv, variable);
            else
              scatter_slots<NS_MAIN + NS_MID, 0>(jb, n, l, own, vec, sv, variable);
          } else if constexpr (TAB_LDS) {
            // high degree: stream the table rows (see SlotStream); slot order as everywhere: D rows of the states, variable
            // entries, D rows of the controls (DIFF_U), mid-point rows of the controls (not for node 0)
            if (own) {
              auto emit = [&](auto& S, const bool with_mid) {
                constexpr int LD = NX * P1, LV = G::NJV, LU = G::DIFF_U ? NU * P1 : 0;
                constexpr bool H1 = (LD & 1) != 0, H2 = ((LD + LV) & 1) != 0, H3 = ((LD + LV + LU) & 1) != 0;
                stream_rows<0, NX, false, P1>(S, [&](int a, int j) {
                  const double d = sD[drow + j];
                  return (j == k) ? d - dd[a] : d;
                });
                stream_regs<0, LV, H1>(S, [&](int q) { return jv[q]; });
                if constexpr (G::DIFF_U) stream_rows<0, NU, H2, P1>(S, [&](int, int j) { return sD[drow + j]; });
                if (with_mid) {
                  if constexpr (NS_MID > 0) stream_rows<0, NU, H3, P1>(S, [&](int, int j) { return sC[crow + j]; });
                  if constexpr (((NS_MAIN + NS_MID) & 1) != 0)
                    *reinterpret_cast<double*>(S.base + (uint32_t)(NS_MAIN + NS_MID - 1) * ((uint32_t)n * 8u) + (uint32_t)l * 8u) = S.carry;
                } else if constexpr ((NS_MAIN & 1) != 0) {
                  *reinterpret_cast<double*>(S.base + (uint32_t)(NS_MAIN - 1) * ((uint32_t)n * 8u) + (uint32_t)l * 8u) = S.carry;
                }
              };
              const uint32_t step = 2u * (uint32_t)n * 8u;
              if (T.node0) {
                SlotStream<false> S{reinterpret_cast<char*>(jb), (uint32_t)l * 16u, step, 0.0};
                emit(S, false);
              } else if (vec) {
                SlotStream<true> S{reinterpret_cast<char*>(jb), (uint32_t)l * 16u, step, 0.0};
                emit(S, true);
              } else {
                SlotStream<false> S{reinterpret_cast<char*>(jb), (uint32_t)l * 16u, step, 0.0};
                emit(S, true);
              }
            }
          } else if (T.node0) {  // node 0 owns no mid-point row: its block ends after the main slots
            scatter_slots<NS_MAIN, 0>(jb, n, l, own, false, sv);
          } else {
            scatter_slots<NS_MAIN + NS_MID, 0>(jb, n, l, own, vec, sv);
          }
        }
      }
    }
    // tile sums
#ifndef MPX_ABL_NO_RED
#pragma unroll
    for (int r = 0; r < NRED; ++r) {
      double v = wave_sum(own ? red[r] : 0.0);
      if (lane == 0) sRed[buf][wave][r] = v;
    }
#endif
    cur = nxt;
  }
  __syncthreads();
  if (it > 0 && l < NRED) {
    double v = 0;
#pragma unroll
    for (int w = 0; w < MPX_TILE / 64; ++w) v += sRed[(it - 1) & 1][w][l];
    io.partial[((int64_t)(b1 - 1) * io.n_tiles_total) * io.nred + tslot + l] = v;
  }
}

// All phases of a single-degree grid in one launch (MpxNodeMultiArgs): the grid's x dimension holds the tiles of every phase one
// after the other, the XCD-blocked item walk runs over the WHOLE grid (so an XCD still streams one contiguous share of the
// outputs), and a workgroup dispatches on the phase its tile belongs to -- uniform per workgroup; the node functions of the
// phases are different generated code, the register budget of the kernel is the largest of them.  Bit-identical to the
// per-phase launches (same node_body, same tile, same slots).
template <int P, int MODE, int PH = 0>
__device__ __forceinline__ void node_all_dispatch(const MpxNodeMultiArgs& M, const int ph, const int bx, const unsigned by) {
  if constexpr (PH < MPX_NPH) {
    if (ph == PH) node_body<PH, P, MODE>(M.a[PH], -1, bx, by);
    else node_all_dispatch<P, MODE, PH + 1>(M, ph, bx, by);
  }
}
template <int P, int MODE>
__device__ __forceinline__ void node_all(const MpxNodeMultiArgs& M) {
#if defined(MPX_MAP_NATURAL)
  const unsigned gt = blockIdx.x, by = blockIdx.y;
#else
  const unsigned lin_ = blockIdx.y * gridDim.x + blockIdx.x, tot_ = gridDim.x * gridDim.y;
  const unsigned xcd_ = lin_ % 8, q_ = tot_ / 8, r_ = tot_ % 8;
  const unsigned item_ = xcd_ * q_ + (xcd_ < r_ ? xcd_ : r_) + lin_ / 8;
  const unsigned gt = item_ % gridDim.x, by = item_ / gridDim.x;
#endif
  int ph = 0;
#pragma unroll
  for (int q = 1; q < MPX_NPH; ++q) ph += (q < M.n_ph && (int)gt >= M.tile_cum[q]) ? 1 : 0;
  node_all_dispatch<P, MODE>(M, ph, (int)gt - M.tile_cum[ph], by);
}

// ---------------------------------------------------------------------------------------------------------------------
// Light passes -- f, g and grad_f WITHOUT the Jacobian values, what a line search calls (nlp_f / nlp_g alone) -- of grids with a high
// degree (MPX_TABLES_IN_LDS_ABOVE < P <= 31).  In node_body such a pass is bound by the LDS pipe, not by HBM: every lane re-reads its
// row of D and the segment's X from LDS for each of the 3 (P + 1) products of a node, 186 ds_read_b64 per lane and point at degree 30
// -- counters of mpx_node_fg_0_30 on config 3 (profiles/r4_c3_fg): LDS busy 67 %, VALU busy 9 %, 318 us for 0.5 GB.  This is the case
// north_star reserves MFMA for ("only if the D.X defect contraction proves dense enough"):
//   * a WAVEFRONT works on a group: up to 16 consecutive segments of the high degree at one evaluation point, plus the low-degree
//     segments between them (config 3: [3, 30, 3] ...) -- ONE contiguous span of the phase's nodes.  The span's X / U rows enter LDS
//     with fully coalesced loads (512 B per instruction), the rows of g / grad_f leave the same way: no partial lines, no staging
//     block, no second bucket launch (the heavy passes need row spans + a staging round trip for that).
//   * D.X_a, D.U_c, C_mid.U_c of the 16 segments are (P+1)x(P+1) by (P+1)x16 products on the matrix cores (v_mfma_f64_16x16x4_f64): A
//     operand = the table (the lane's entries stay in registers for the life of the wavefront), B operand column n = segment n.
//     C/D layout (row = (lane >> 4) + 4 reg, col = lane & 15; tools/mfma_f64_probe.hip): lane (n, q) ends up with the rows
//     k = q + 4 e of segment n -- exactly the nodes whose X / U it supplied as B operand (k = 4 ks + q), so the node functions are
//     evaluated in place on the same registers.
//   * The matrix core accumulates the four products of an instruction in order, fused (probe: 256 / 256 results bit-equal to the
//     sequential fma chain), and K runs over the nodes in order: g is BIT-IDENTICAL to node_body's (tested).  The padding column
//     (node P + 1) multiplies a zero of the table.
//   * The low-degree nodes of the span: one lane each, the same fma chains as node_body over the span in LDS.
//   * Wavefronts are persistent and independent (no barrier): each walks the (group, point) items with the stride of the grid.  The
//     per-point sums (f; for the grad_f pass d/dt0, d/dtf, d/dA) of a group go to the partial-sum slot slot_first + group -- lane,
//     lane groups, segments, low-degree nodes, in that order: fixed, independent of the batch; the boundary kernel is unchanged.
//     f of a light pass and f of a heavy pass (node_body's tiles) round differently in the last bit; g and the node entries of
//     grad_f do not.
// History of the design (profiles/r4_c3_fg/README.md): 16 evaluation points as the 16 columns, loads straight from global memory:
// every access a 32-byte granule, the memory system delivered 3 TB/s at most and the low-degree bucket's own launch cost 86 us.
// ---------------------------------------------------------------------------------------------------------------------
typedef double mpx_d4 __attribute__((ext_vector_type(4)));

#ifndef MPX_LIGHT_XCD_BLOCKED
#define MPX_LIGHT_XCD_BLOCKED 0
#endif
#ifndef MPX_LOW_TOTALS_GROUPED
#define MPX_LOW_TOTALS_GROUPED 1  // 0: one wave_total_dpp per chunk where the chunk is worked on (the form of most of round 4, A/B)
#endif
#ifndef MPX_LOW_TOTALS_GROUP
#define MPX_LOW_TOTALS_GROUP 12
#endif
#ifndef MPX_LIGHT_DESC_LDS
#define MPX_LIGHT_DESC_LDS 1
#endif
// Workgroup i of a launch runs on XCD i % 8: with MPX_LIGHT_XCD_BLOCKED=1 the workgroups of one XCD take CONSECUTIVE items of every
// round of the persistent loop (each L2 streams a contiguous eighth of the round's rows instead of every eighth group of items), as
// node_body's XCD-blocked walk does for the tiles.  Measured in process (tools/r4_light_ab.py): 1.5 % slower at config 2 (g: 208.5
// against 205.1 us), 3 - 5 % slower at config 3 -- off.
// The wavefront index through readfirstlane (round 5): the compiler then knows that the item of a wavefront -- group, evaluation point,
// every base address derived from them -- is uniform: scalar registers and scalar loads (the group descriptor) instead of per-lane
// 64-bit address pairs in VGPRs.  -DMPX_LIGHT_SCALAR_WAVE=0: the per-lane form of round 4 (A/B).
#ifndef MPX_LIGHT_SCALAR_WAVE
#define MPX_LIGHT_SCALAR_WAVE 1
#endif
#if MPX_LIGHT_SCALAR_WAVE
#define MPX_LIGHT_UNIFORM_WAVE(w) __builtin_amdgcn_readfirstlane(w)
#else
#define MPX_LIGHT_UNIFORM_WAVE(w) (w)
#endif
__device__ __forceinline__ unsigned light_block_xcd() {
  if (MPX_LIGHT_XCD_BLOCKED && gridDim.x % 8 == 0) return (blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8;
  return blockIdx.x;
}

// PF (round 5): the ONE low degree of the grid when there is exactly one (config 3: [3, 30, 3] -> 3), else 0.  With it the low-degree
// nodes' contractions are straight-line code -- table rows and span values requested together, fma chains unrolled -- where the
// run-time degree made every term a dependent LDS round trip inside a loop (4 terms x 3 chains x 2 turns: 16 of 151 us, r4_light_chains).
#ifndef MPX_LIGHT_STATIC_LOW
#define MPX_LIGHT_STATIC_LOW 1
#endif
// Measured with it, in process on config 3 at B = 512 (tools/r4_light_ab.py, profiles/r5_c3_light/ab.txt; nlp_g / nlp_f / f + grad_f,
// us per whole pass; round-4 form 150.2 / 64.5 / 134.6): the static low degree 142.8 / 64.5 / 135.2 (kept); rows leaving the span buffer
// 16 bytes per lane +-0.5 % (MPX_LIGHT_ROWS16, off); whole 128-node chunks loaded under a UNIFORM branch without per-lane bounds
// 187 / 80 / 167 -- the branch splits the loads of a span over basic blocks and the compiler waits for each block's loads before the
// next: the one property this kernel lives on is that ALL loads of an item are in flight together (MPX_LIGHT_FAST_LOAD, off).
#ifndef MPX_LIGHT_ROWS16
#define MPX_LIGHT_ROWS16 0  // 1: rows of g / grad_f leave the span buffer 16 bytes per lane (1 KB per wavefront store) instead of 8
#endif
#ifndef MPX_LIGHT_FAST_LOAD
#define MPX_LIGHT_FAST_LOAD 0  // 1: whole 128-node chunks of a span are loaded without per-lane bounds (uniform branch) -- 25 % SLOWER
#endif
template <int PH, int P, int MODE, int PF = 0>
__device__ __forceinline__ void light_body(const MpxLightArgs& L) {
  using G = mpxgen::Phase<PH>;
  const MpxNodeArgs& A = L.node;
  constexpr int NX = G::NX, NU = G::NU, NA = G::NA, NC = G::NC;
  constexpr int NIN = NX + NU;
  constexpr int P1 = P + 1, MT = (P1 + 15) / 16, KS = 4 * MT;
  static_assert(MT <= 2, "light_body: degrees up to 31");
  constexpr int NRED = (MODE == MPX_MODE_FG) ? 1 : G::NRED;
  // rows of g a node writes: defect, path, DU, mU (the mid-point row before node i is row i - 1 of its block)
  constexpr int R_C = NX, R_DU = NX + NC, R_MU = R_DU + (G::DIFF_U ? NU : 0), NG = R_MU + (G::MIDU ? NU : 0);
  extern __shared__ double sBuf[];  // [wavefront][NIN][span_cap]: the span's inputs, then (in turns of NIN rows) its outputs; the low-degree tables
  const int t = threadIdx.x, wave = MPX_LIGHT_UNIFORM_WAVE(t >> 6), l = t & 63, n = l & 15, q = l >> 4;
  const int N = A.N, cap = L.span_cap;
  double* __restrict__ sW = sBuf + (size_t)wave * NIN * cap;
  const double* __restrict__ sTab = sBuf + (size_t)MPX_LIGHT_WAVES * NIN * cap;
  for (int e = t; e < L.ftab_n; e += 64 * MPX_LIGHT_WAVES) sBuf[(size_t)MPX_LIGHT_WAVES * NIN * cap + e] = L.ftab[e];
  const MpxIO& io = A.io;
  // A operands: lane l supplies row (l & 15) of an M tile and column (l >> 4) of a K step.  The differentiation table stays in
  // registers; the mid-point table (one chain per control) and the per-point constants are read from LDS where they are used --
  // with everything in registers a wavefront needed 336 of them and a compute unit held four wavefronts.
  double AD[MT][KS];
  __shared__ double sAC[MT][KS][64];
  __shared__ double sTk[4 * KS], sWt[4 * KS];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int k = 16 * mt + n, j = 4 * ks + q;
      const bool in = k <= P && j <= P;
      AD[mt][ks] = in ? A.Dmat[(in ? k : 0) * P1 + (in ? j : 0)] : 0.0;
      if (wave == 0) sAC[mt][ks][l] = (G::MIDU && in && k >= 1) ? A.Cmid[((in && k >= 1) ? k - 1 : 0) * P1 + (in ? j : 0)] : 0.0;
    }
  // (the quadrature weight of point k of a segment is w_k of its degree: the composite vector repeats the table, mpopt.py:4060-4062)
  if (t < 4 * KS) sTk[t] = A.tk[t <= P ? t : P], sWt[t] = L.wdeg[t <= P ? t : P];
  // (degree and table offsets of the low-degree buckets, indexed by a lane-dependent value: out of the kernel arguments that was a
  // dependent round trip to memory -- global_load_dword, s_waitcnt vmcnt -- in front of every turn of low-degree nodes; from LDS
  // config 3 nlp_g 162.6 -> 154.1 us in process, -DMPX_LIGHT_DESC_LDS=0 restores it.  Measured with it and not kept
  // (profiles/r4_light_chains): the segment widths requested before the wait for the span (+-0), the low-degree contractions four
  // terms at a time (spills: f + grad_f 138 -> 159 us), the row stores four LDS reads at a time (+1 %).)
  __shared__ int sFdeg[MPX_LIGHT_MAXDEG], sFD[MPX_LIGHT_MAXDEG], sFC[MPX_LIGHT_MAXDEG];
  if (t < MPX_LIGHT_MAXDEG) sFdeg[t] = L.fdeg[t], sFD[t] = L.fD_off[t], sFC[t] = L.fC_off[t];
  __syncthreads();  // (the only barrier of the kernel: the tables are in place)
  const bool want_g = io.g != nullptr, want_q = MODE == MPX_MODE_FGJ && io.grad != nullptr;
  const int64_t total = (int64_t)L.n_groups * (io.B - io.b_first), stride = (int64_t)gridDim.x * MPX_LIGHT_WAVES;
  auto lds_sync = [] { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };  // (one wavefront: LDS operations complete in order)
#ifdef MPX_LIGHT_STAMPS  // phase stamps of one wavefront's third item (wall_clock64: 100 MHz), forced waits at the phase ends
  int it_ = 0;
#define MPX_LSTAMP(k) if (L.dbg && blockIdx.x == 1 && wave == 1 && it_ == 2) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); if (l == 0) L.dbg[k] = wall_clock64(); }
#else
#define MPX_LSTAMP(k)
#endif
  // Every wavefront of the chip starts at the same time and the phases of an item take the same time everywhere: left alone, all of
  // them sit in the load phase together, then in the matrix phase, then in the store phase (stamps: 7 + 6.5 + 3.5 us per item), and
  // the memory system idles two thirds of the time.  Half of the wavefronts start half an item late.
#ifndef MPX_LIGHT_STAGGER
#define MPX_LIGHT_STAGGER 0
#endif
  if (wave & 1)
    for (int k = 0; k < MPX_LIGHT_STAGGER; ++k) __builtin_amdgcn_s_sleep(127);
  for (int64_t item = (int64_t)light_block_xcd() * MPX_LIGHT_WAVES + wave; item < total; item += stride) {
    MPX_LSTAMP(0)
    const int gi = (int)(item % L.n_groups), b = io.b_first + (int)(item / L.n_groups);
    const MpxLightGroup Gp = L.groups[gi];
    const double* __restrict__ zb = io.z + (int64_t)b * io.z_stride + A.z_off;
    const double* __restrict__ zt = zb + (int64_t)NIN * N;
    const double t0v = zt[0], tfv = zt[1];
    Vec<NA> As;
#pragma unroll
    for (int c = 0; c < NA; ++c) As[c] = zt[2 + c];
    constexpr int FMAX = 2;  // turns of 64 low-degree nodes a lane can hold (the host keeps f_count <= 64 * FMAX)
    // descriptors first: the loads that depend on them (widths of the segments) are then in flight together with the span's
    MpxLightForeign Fd[FMAX];
#pragma unroll
    for (int u = 0; u < FMAX; ++u) Fd[u] = L.foreign[Gp.f_first + (64 * u + l < Gp.f_count ? 64 * u + l : 0)];
    const bool col = n < Gp.n_light;  // the lane's high-degree segment (column n of the products)
    const int m0 = L.first_node + (Gp.seg_first + (col ? n : 0)) * P;
    const int st = A.node_i[m0] - 1, s = A.node_sk[m0] >> 8;  // point 0 of the segment in the phase; the segment
    const int64_t woff = (int64_t)b * io.w_stride + A.seg_off;
    // (1) the span's rows of X / U: coalesced loads, into LDS
    lds_sync();  // (the previous item's output reads of this buffer are done)
    {  // (ALL loads of the span are issued before the first LDS write: one round trip to memory per item, not one per batch)
      // (16 bytes per lane; rows are only 8-byte aligned -- N is odd as often as not --, an odd row length ends with an 8-byte load)
      constexpr int CH2 = (MPX_LIGHT_CHUNKS + 1) / 2;  // 128-node chunks a span can have (the host caps the span at 64 * MPX_LIGHT_CHUNKS nodes)
      typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
      d2u v[NIN][CH2];
      const int len_r_u = MPX_LIGHT_SCALAR_WAVE ? __builtin_amdgcn_readfirstlane(Gp.len_r) : Gp.len_r;
#pragma unroll
      for (int u = 0; u < CH2; ++u) {
        const int idx = 128 * u + 2 * l;
        if (MPX_LIGHT_FAST_LOAD && MPX_LIGHT_SCALAR_WAVE && 128 * (u + 1) <= len_r_u) {  // (uniform: a whole chunk)
#pragma unroll
          for (int a = 0; a < NIN; ++a) v[a][u] = *(const d2u*)(zb + (int64_t)a * N + Gp.lo_r + idx);
        } else {
#pragma unroll
          for (int a = 0; a < NIN; ++a) {
            const double* __restrict__ src = zb + (int64_t)a * N + Gp.lo_r + idx;
            if (idx + 1 < Gp.len_r) v[a][u] = *(const d2u*)src;
            else v[a][u] = d2u{idx < Gp.len_r ? src[0] : 0.0, 0.0};
          }
        }
      }
#pragma unroll
      for (int u = 0; u < CH2; ++u) {
        const int idx = 128 * u + 2 * l;
        if (MPX_LIGHT_FAST_LOAD && MPX_LIGHT_SCALAR_WAVE && 128 * (u + 1) <= len_r_u) {
#pragma unroll
          for (int a = 0; a < NIN; ++a) *(d2u*)&sW[a * cap + idx] = v[a][u];  // (cap is even, idx is even: 16-byte aligned in LDS)
        } else if (idx < Gp.len_r) {
#pragma unroll
          for (int a = 0; a < NIN; ++a) {
            sW[a * cap + idx] = v[a][u].x;
            if (idx + 1 < Gp.len_r) sW[a * cap + idx + 1] = v[a][u].y;
          }
        }
      }
    }
    const int sp = st - Gp.lo_r;  // point 0 of the lane's segment in the span
    const double ws = io.w[woff + s], wc = io.wcum[woff + s];
    double wsf[FMAX], wcf[FMAX];
#pragma unroll
    for (int u = 0; u < FMAX; ++u) wsf[u] = io.w[woff + Fd[u].s], wcf[u] = io.wcum[woff + Fd[u].s];
    lds_sync();
    MPX_LSTAMP(1)
    double zX[NX][KS], zU[NU > 0 ? NU : 1][KS];
#pragma unroll
    for (int e = 0; e < KS; ++e) {
      const int j = q + 4 * e <= P ? q + 4 * e : P;  // (the padding node P + 1 meets a zero of the table)
#pragma unroll
      for (int a = 0; a < NX; ++a) zX[a][e] = sW[a * cap + sp + j];
#pragma unroll
      for (int c = 0; c < NU; ++c) zU[c][e] = sW[(NX + c) * cap + sp + j];
    }
    // (2) the contractions on the matrix core
    mpx_d4 aX[NX][MT], aDU[NU > 0 ? NU : 1][MT], aCU[NU > 0 ? NU : 1][MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int a = 0; a < NX; ++a) aX[a][mt] = mpx_d4{0, 0, 0, 0};
#pragma unroll
      for (int c = 0; c < NU; ++c) aDU[c][mt] = mpx_d4{0, 0, 0, 0}, aCU[c][mt] = mpx_d4{0, 0, 0, 0};
    }
#ifdef MPX_ABL_LIGHT_NO_MFMA  // ablation: no matrix instructions
    if (false) {
#else
    if (want_g) {
#endif
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
          for (int a = 0; a < NX; ++a) aX[a][mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(AD[mt][ks], zX[a][ks], aX[a][mt], 0, 0, 0);
          if constexpr (G::DIFF_U) {
#pragma unroll
            for (int c = 0; c < NU; ++c) aDU[c][mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(AD[mt][ks], zU[c][ks], aDU[c][mt], 0, 0, 0);
          }
          if constexpr (G::MIDU) {
#pragma unroll
            for (int c = 0; c < NU; ++c) aCU[c][mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(sAC[mt][ks][l], zU[c][ks], aCU[c][mt], 0, 0, 0);
          }
        }
    }
    // (3) the low-degree nodes of the span, one lane each (in turns of 64): node_body's fma chains over the span in LDS
    double fval[FMAX][NG > 0 ? NG : 1], fgrd[FMAX][MODE == MPX_MODE_FGJ ? NIN : 1];
    int fpos[FMAX];
    Vec<NRED> fred;
#pragma unroll
    for (int r = 0; r < NRED; ++r) fred[r] = 0.0;
#pragma unroll
    for (int u = 0; u < FMAX; ++u) {
      fpos[u] = -1;
      const int fi = 64 * u + l;
#ifdef MPX_ABL_LIGHT_NO_FOREIGN  // ablation: the low-degree nodes are skipped
      if (false) {
#else
      if (fi < Gp.f_count) {
#endif
        const MpxLightForeign F = Fd[u];
        constexpr bool SLOW = PF > 0 && MPX_LIGHT_STATIC_LOW;
#if MPX_LIGHT_DESC_LDS
        // (one low degree, compile time: its two tables are the whole of sTab -- D at 0, C_mid behind it, mpx_layout.cpp: lplan.ftab)
        const int di = F.dk >> 8, k = F.dk & 255, pf = SLOW ? PF : sFdeg[di], p1f = pf + 1;
        const double* __restrict__ Dr = sTab + (SLOW ? 0 : sFD[di]) + k * p1f;
        const double* __restrict__ Cr = sTab + (SLOW ? (PF + 1) * (PF + 1) : sFC[di]) + (k >= 1 ? k - 1 : 0) * p1f;
#else
        const int di = F.dk >> 8, k = F.dk & 255, pf = L.fdeg[di], p1f = pf + 1;
        const double* __restrict__ Dr = sTab + L.fD_off[di] + k * p1f;
        const double* __restrict__ Cr = sTab + L.fC_off[di] + (k >= 1 ? k - 1 : 0) * p1f;
#endif
        fpos[u] = F.pos;
        Vec<NX> Xs, fx;
        Vec<NU> Us;
        Vec<NC> cc;
#pragma unroll
        for (int a = 0; a < NX; ++a) Xs[a] = sW[a * cap + F.pos];
#pragma unroll
        for (int c = 0; c < NU; ++c) Us[c] = sW[(NX + c) * cap + F.pos];
        const double kapf = wsf[u] * A.inv_dtau, thf = wcf[u] + wsf[u] * F.tk;
        Vec<NRED> gr;
        if constexpr (MODE == MPX_MODE_FG) {
          G::fg(Xs, Us, t0v, tfv, As, kapf, thf, F.w, fx, cc, gr[0]);
        } else {
          Vec<NX> dd;
          Vec<G::NJV> jv;
          Vec<NIN> gn;
          G::fgj(Xs, Us, t0v, tfv, As, kapf, thf, F.w, fx, cc, dd, jv, gn, gr);
#pragma unroll
          for (int a = 0; a < NIN; ++a) fgrd[u][a] = gn[a];
        }
#pragma unroll
        for (int r = 0; r < NRED; ++r) fred[r] += gr[r];
        if constexpr (SLOW) {
          if (want_g) {  // the same fma chains, same order; every operand requested before the first is used
            constexpr int PF1 = PF + 1;
            double dr[PF1], cr[PF1], xv[NIN][PF1];
#pragma unroll
            for (int j = 0; j < PF1; ++j) {
              dr[j] = Dr[j];
              if constexpr (G::MIDU) cr[j] = Cr[j];
#pragma unroll
              for (int a = 0; a < NIN; ++a)
                if (a < NX || G::DIFF_U || G::MIDU) xv[a][j] = sW[a * cap + F.pos0 + j];
            }
#pragma unroll
            for (int a = 0; a < NX; ++a) {
              double acc = 0;
#pragma unroll
              for (int j = 0; j < PF1; ++j) acc = fma(dr[j], xv[a][j], acc);
              fval[u][a] = acc - fx[a];
            }
#pragma unroll
            for (int jj = 0; jj < NC; ++jj) fval[u][R_C + jj] = cc[jj];
            if constexpr (G::DIFF_U) {
#pragma unroll
              for (int c = 0; c < NU; ++c) {
                double acc = 0;
#pragma unroll
                for (int j = 0; j < PF1; ++j) acc = fma(dr[j], xv[NX + c][j], acc);
                fval[u][R_DU + c] = acc;
              }
            }
            if constexpr (G::MIDU) {
#pragma unroll
              for (int c = 0; c < NU; ++c) {
                double acc = 0;
#pragma unroll
                for (int j = 0; j < PF1; ++j) acc = fma(cr[j], xv[NX + c][j], acc);
                fval[u][R_MU + c] = k >= 1 ? acc : 0.0;
              }
            }
          }
        } else
        if (want_g) {
#pragma unroll
          for (int a = 0; a < NX; ++a) {
            double acc = 0;
            for (int j = 0; j < p1f; ++j) acc = fma(Dr[j], sW[a * cap + F.pos0 + j], acc);
            fval[u][a] = acc - fx[a];
          }
#pragma unroll
          for (int jj = 0; jj < NC; ++jj) fval[u][R_C + jj] = cc[jj];
          if constexpr (G::DIFF_U) {
#pragma unroll
            for (int c = 0; c < NU; ++c) {
              double acc = 0;
              for (int j = 0; j < p1f; ++j) acc = fma(Dr[j], sW[(NX + c) * cap + F.pos0 + j], acc);
              fval[u][R_DU + c] = acc;
            }
          }
          if constexpr (G::MIDU) {
#pragma unroll
            for (int c = 0; c < NU; ++c) {
              double acc = 0;
              if (k >= 1)
                for (int j = 0; j < p1f; ++j) acc = fma(Cr[j], sW[(NX + c) * cap + F.pos0 + j], acc);
              fval[u][R_MU + c] = acc;
            }
          }
        }
      }
    }
    // (4) node functions of the lane's high-degree nodes
    const double kap = ws * A.inv_dtau;
    Vec<NRED> red;
#pragma unroll
    for (int r = 0; r < NRED; ++r) red[r] = 0.0;
    double fxs[KS][NX], ccs[KS][NC > 0 ? NC : 1], ogrd[KS][MODE == MPX_MODE_FGJ ? NIN : 1];
#pragma unroll
    for (int e = 0; e < KS; ++e) {
      const int j = q + 4 * e;
      const bool valid = col && j <= P && (j >= 1 || s == 0);  // (point 0 of a segment belongs to the previous one -- except node 0 of the phase)
      const double th = wc + ws * sTk[j];
      Vec<NX> Xs, fx;
      Vec<NU> Us;
      Vec<NC> cc;
#pragma unroll
      for (int a = 0; a < NX; ++a) Xs[a] = zX[a][e];
#pragma unroll
      for (int c = 0; c < NU; ++c) Us[c] = zU[c][e];
      Vec<NRED> gr;
      if constexpr (MODE == MPX_MODE_FG) {
        G::fg(Xs, Us, t0v, tfv, As, kap, th, sWt[j], fx, cc, gr[0]);
      } else {
        Vec<NX> dd;
        Vec<G::NJV> jv;
        Vec<NIN> gn;
        G::fgj(Xs, Us, t0v, tfv, As, kap, th, sWt[j], fx, cc, dd, jv, gn, gr);
#pragma unroll
        for (int a = 0; a < NIN; ++a) ogrd[e][a] = gn[a];
      }
      if (valid) {
#pragma unroll
        for (int r = 0; r < NRED; ++r) red[r] += gr[r];
      }
#pragma unroll
      for (int a = 0; a < NX; ++a) fxs[e][a] = fx[a];
#pragma unroll
      for (int jj = 0; jj < NC; ++jj) ccs[e][jj] = cc[jj];
    }
    // (value of row r of g at the lane's e-th node: compile-time indices after unrolling)
    auto oval = [&](int e, int r) -> double {
      if (r < R_C) return aX[r < NX ? r : 0][e / 4][e % 4] - fxs[e][r < NX ? r : 0];
      if (r < R_DU) return ccs[e][NC > 0 ? r - R_C : 0];
      if (r < R_MU) return aDU[NU > 0 ? r - R_DU : 0][e / 4][e % 4];
      return aCU[NU > 0 ? r - R_MU : 0][e / 4][e % 4];
    };
    MPX_LSTAMP(2)
    // (5) outputs: NIN rows at a time through the LDS buffer (every input read is done), coalesced stores of the owned span.
    // Row r of the span buffer holds positions relative to lo_r; a mid-point row holds the row of node i at position i - 1.
    const int w0 = Gp.lo_w - Gp.lo_r;  // first owned position
    auto put_rows = [&](auto value_light, auto value_foreign, int r0, int nr, auto row_ptr, auto shift) {
      lds_sync();
#pragma unroll
      for (int e = 0; e < KS; ++e) {
        const int j = q + 4 * e;
        if (col && j <= P && (j >= 1 || s == 0)) {
#pragma unroll
          for (int r = 0; r < NIN; ++r)
            if (r < nr && sp + j - shift(r0 + r) >= 0) sW[r * cap + sp + j - shift(r0 + r)] = value_light(e, r0 + r);  // (node 0 has no mid-point row)
        }
      }
#pragma unroll
      for (int u = 0; u < FMAX; ++u)
        if (fpos[u] >= 0) {
#pragma unroll
          for (int r = 0; r < NIN; ++r)
            if (r < nr && fpos[u] - shift(r0 + r) >= 0) sW[r * cap + fpos[u] - shift(r0 + r)] = value_foreign(u, r0 + r);
        }
      lds_sync();
#pragma unroll
      for (int r = 0; r < NIN; ++r)
        if (r < nr) {
          double* __restrict__ dst = row_ptr(r0 + r);  // address of the row's entry of node lo_r (the shift is in the positions)
          const int sh = shift(r0 + r);
          // owned positions [w0, w0 + len_w); a shifted row has no entry for node 0 of the phase
          const int p_lo = w0 - sh < 0 ? 0 : w0 - sh, p_hi = w0 + Gp.len_w - sh;
          if constexpr (MPX_LIGHT_ROWS16) {  // pairs (even positions: 16-byte aligned in LDS; 8-byte aligned in memory), ragged ends singly
            typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
            const int e_lo = (p_lo + 1) & ~1, e_hi = p_hi & ~1;
            if (l == 0 && p_lo < e_lo && p_lo < p_hi) dst[p_lo] = sW[r * cap + p_lo];
            if (l == 1 && e_hi < p_hi && e_hi >= e_lo) dst[e_hi] = sW[r * cap + e_hi];
            for (int idx = e_lo + 2 * l; idx < e_hi; idx += 128) *(d2u*)(dst + idx) = *(const d2u*)&sW[r * cap + idx];
          } else {
            for (int idx = p_lo + l; idx < p_hi; idx += 64) dst[idx] = sW[r * cap + idx];
          }
        }
    };
    if (want_g) {
      double* __restrict__ gb = io.g + (int64_t)b * io.g_stride;
      for (int r0 = 0; r0 < NG; r0 += NIN) {
        put_rows([&](int e, int r) { return oval(e, r); }, [&](int u, int r) { return fval[u][r]; }, r0, NG - r0 < NIN ? NG - r0 : NIN,
                 [&](int r) -> double* {
                   if (r < R_C) return gb + A.g_off_F + (int64_t)r * N + Gp.lo_r;
                   if (r < R_DU) return gb + A.g_off_C + (int64_t)(r - R_C) * N + Gp.lo_r;
                   if (r < R_MU) return gb + A.g_off_DU + (int64_t)(r - R_DU) * N + Gp.lo_r;
                   return gb + A.g_off_mU + (int64_t)(r - R_MU) * (N - 1) + Gp.lo_r;
                 },
                 [&](int r) { return r >= R_MU ? 1 : 0; });
      }
    }
    if constexpr (MODE == MPX_MODE_FGJ) {
      if (want_q) {
        double* __restrict__ qb = io.grad + (int64_t)b * io.grad_stride + A.z_off;
        put_rows([&](int e, int r) { return ogrd[e][r]; }, [&](int u, int r) { return fgrd[u][r]; }, 0, NIN,
                 [&](int r) -> double* { return qb + (int64_t)r * N + Gp.lo_r; }, [&](int) { return 0; });
      }
    }
    // (6) the point's sums over this group: the lane's nodes in order, then the wavefront's fixed tree (wave_sum) over the lanes of
    // the high-degree nodes and over those of the low-degree nodes
#pragma unroll
    for (int r = 0; r < NRED; ++r) {
      const double v = wave_sum(red[r]);
      const double fs = wave_sum(fred[r]);
      if (l == 0) io.partial[((int64_t)b * io.n_tiles_total + L.slot_first + gi) * io.nred + r] = v + fs;
    }
    MPX_LSTAMP(3)
#ifdef MPX_LIGHT_STAMPS
    ++it_;
#endif
  }
#undef MPX_LSTAMP
}

// ---------------------------------------------------------------------------------------------------------------------
// Light passes of SINGLE-DEGREE grids of low degree (P <= 12; BASELINE configs 1, 2, 4, 5): the same persistent, span-coalesced
// scheme as light_body without the matrix cores -- a wavefront takes a span of 64 * CHL consecutive nodes of one evaluation point
// (plus the few nodes that complete its first and last segment), stages the span's X / U rows in LDS (16-byte loads, all of them in
// flight before the first LDS write), evaluates its nodes lane by lane (64 at a time) with node_body's fma chains over LDS, and
// stores every row of g / grad_f straight from the registers: the lanes of a chunk hold consecutive nodes, so a row leaves as one
// 512-byte store per chunk.  node_body's light passes were bound by the lifetime of a workgroup, not by HBM (0.38 - 0.53 of peak at
// config 2 even with 16 evaluation points per workgroup: barrier per point, tile descriptors, 250-node tiles); here nothing but
// one table barrier at kernel start.  Measured at config 2, B = 4096 (profiles/r4_lightlow): nlp_f 80.5 us (6.5 TB/s, 0.81 of
// peak), nlp_g 182.6 us (0.70), nlp_f + nlp_grad_f 181.2 us (0.70).  What did NOT help: LDS-DMA double buffering of 256-node spans
// (98 / 216 us), staging the outputs through LDS for 16-byte stores (83 / 197), three workgroups per compute unit (91 / 197).
// g and the node entries of grad_f: bit-identical to node_body's.  Sums: wavefront total per 64-node chunk, the chunks of a long span
// added in order, the spans added by the boundary pass in order -- and the same grouping when every chunk is its own span, so that
// SMALL batches can run the same arithmetic with one chunk per wavefront (SMALL = true: a single evaluation keeps 79 wavefronts busy at config 2 instead of 10; with the long
// spans a single nlp_f + nlp_g + nlp_grad_f pass took 33 instead of 22 us) and still give the bits of a large batch.
// ---------------------------------------------------------------------------------------------------------------------

// One implementation for the per-phase kernels (PH0 = the phase, NPHK = 1) and the all-phases kernels (PH0 = 0, NPHK = MPX_NPH;
// round 5): there an item is (span, phase, evaluation point) -- the phases of a grid share its nodes and tables and differ in the
// offsets of MpxLightPhase and in the generated functions; a wavefront dispatches on the phase of its item.  One launch, one
// table prologue and one tail for all phases (config 4: 8 items per wavefront instead of two launches of 4).
template <int P, int MODE, bool SMALL, int PH0, int NPHK>
__device__ __forceinline__ void light_low_run(const MpxLightArgs& L, const MpxLightPhase* __restrict__ phases) {
  using G0 = mpxgen::Phase<PH0>;
  const MpxNodeArgs& A = L.node;
  constexpr int NX = G0::NX, NU = G0::NU, NA = G0::NA;
  constexpr int NIN = NX + NU, P1 = P + 1;
  // span geometry, compile time: rows of CAP doubles per wavefront in 52 KB of LDS per workgroup; CHL chunks of 64 owned nodes
  constexpr int CAP0 = 53248 / (8 * MPX_LIGHT_WAVES * NIN);
  constexpr int CHL = SMALL ? 1 : ((CAP0 - 2 * P - 8) / 64 > MPX_LOW_MAX_CHUNKS ? MPX_LOW_MAX_CHUNKS : ((CAP0 - 2 * P - 8) / 64 < 1 ? 1 : (CAP0 - 2 * P - 8) / 64));
  constexpr int OWN = 64 * CHL, CAP = (OWN + 2 * P + 8 + 1) & ~1;
  typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
  static_assert(MPX_LIGHT_WAVES * NIN * CAP * 8 <= 56 * 1024, "light_low_body: span rows do not fit LDS");
  __shared__ double sBufL[MPX_LIGHT_WAVES][NIN][CAP];
  __shared__ double sD[P1 * P1], sC[P * P1], sTk[P1], sWt[P1];
  const int t = threadIdx.x, wave = MPX_LIGHT_UNIFORM_WAVE(t >> 6), l = t & 63;
  const int N = A.N;
  double(*sW)[CAP] = sBufL[wave];
  const MpxIO& io = A.io;
  for (int e = t; e < P1 * P1; e += 64 * MPX_LIGHT_WAVES) sD[e] = A.Dmat[e];
  for (int e = t; e < P * P1; e += 64 * MPX_LIGHT_WAVES) sC[e] = A.Cmid[e];
  if (t < P1) sTk[t] = A.tk[t], sWt[t] = L.wdeg[t];
  __syncthreads();  // (the only barrier of the kernel)
  const bool want_g = io.g != nullptr, want_q = MODE == MPX_MODE_FGJ && io.grad != nullptr;
  const int n_groups = (N + OWN - 1) / OWN;
  const int64_t total = (int64_t)n_groups * NPHK * (io.B - io.b_first), stride = (int64_t)gridDim.x * MPX_LIGHT_WAVES;
  auto lds_sync = [] { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
#ifdef MPX_LIGHT_STAMPS
  int it_ = 0;
#define MPX_LSTAMP(k) if (L.dbg && blockIdx.x == 1 && wave == 1 && it_ == 2) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); if (l == 0) L.dbg[k] = wall_clock64(); }
#else
#define MPX_LSTAMP(k)
#endif
  // one item: span gi of phase PHc::value (offsets Q) at evaluation point b
  auto run_item = [&](auto PHc, const MpxLightPhase& Q, const int gi, const int b) {
    using G = mpxgen::Phase<decltype(PHc)::value>;
    static_assert(G::NX == NX && G::NU == NU && G::NA == NA, "the phases of an OCP have the same numbers of states, controls and parameters");
    constexpr int NC = G::NC;
    constexpr int NRED = (MODE == MPX_MODE_FG) ? 1 : G::NRED;
    MPX_LSTAMP(0)
    const int lo_w = gi * OWN, len_w = N - lo_w < OWN ? N - lo_w : OWN;
    const int lo_r = lo_w == 0 ? 0 : ((lo_w - 1) / P) * P;                 // point 0 of the first owned node's segment
    const int i_last = lo_w + len_w - 1;
    const int hi_r = i_last == 0 ? 1 : ((i_last - 1) / P + 1) * P + 1;    // one past the last node of the last owned node's segment
    const int len_r = hi_r - lo_r;
    const double* __restrict__ zb = io.z + (int64_t)b * io.z_stride + Q.z_off;
    const double* __restrict__ zt = zb + (int64_t)NIN * N;
    const double t0v = zt[0], tfv = zt[1];
    Vec<NA> As;
#pragma unroll
    for (int c = 0; c < NA; ++c) As[c] = zt[2 + c];
    const int64_t woff = (int64_t)b * io.w_stride + Q.seg_off;
    // the lane's nodes and the widths of their segments (loads in flight together with the span's)
    int sg[CHL], kk[CHL];
    double wsv[CHL], wcv[CHL];
#pragma unroll
    for (int u = 0; u < CHL; ++u) {
      const int i = lo_w + 64 * u + l < N ? lo_w + 64 * u + l : N - 1;
      sg[u] = i == 0 ? 0 : (i - 1) / P, kk[u] = i == 0 ? 0 : (i - 1) % P + 1;
      wsv[u] = io.w[woff + sg[u]], wcv[u] = io.wcum[woff + sg[u]];
    }
    lds_sync();  // (the previous item's output reads of this buffer are done)
    {  // 16 bytes per lane (rows are 8-byte aligned: N is odd as often as not), every load of the span in flight before the first LDS write
      constexpr int CH2 = (CAP + 127) / 128;
      d2u v[NIN][CH2];
#pragma unroll
      for (int u = 0; u < CH2; ++u) {
        const int idx = 128 * u + 2 * l;
#pragma unroll
        for (int a = 0; a < NIN; ++a) {
          const double* __restrict__ src = zb + (int64_t)a * N + lo_r + idx;
          if (idx + 1 < len_r) v[a][u] = *(const d2u*)src;
          else v[a][u] = d2u{idx < len_r ? src[0] : 0.0, 0.0};
        }
      }
#pragma unroll
      for (int u = 0; u < CH2; ++u) {
        const int idx = 128 * u + 2 * l;
        if (idx < len_r) {
#pragma unroll
          for (int a = 0; a < NIN; ++a) sW[a][idx] = v[a][u].x, sW[a][idx + 1] = v[a][u].y;
        }
      }
    }
    lds_sync();
    MPX_LSTAMP(1)
    // Sums: wavefront total of every 64-node chunk (fixed tree); the long spans add their chunks' totals in chunk order into one
    // partial-sum slot per span, the one-chunk spans write a slot per chunk and the boundary pass adds them in the same groups
    // (MpxBoundArgs::part_group): both give the same bits.
    Vec<NRED> tot;
#pragma unroll
    for (int r = 0; r < NRED; ++r) tot[r] = 0.0;
    // chunks whose wavefront totals are formed together: all of the span's in the passes without gradients (one sum per chunk);
    // with the three and more sums per chunk of the gradient passes the grouped form measured 2 % slower (registers), so those keep
    // one wave_total_dpp per chunk and sum.  In process, B = 4096 (tools/r4_light_ab.py, -DMPX_LOW_TOTALS_GROUPED=0 against the
    // default): nlp_f config 5 181.8 -> 168.5 us, config 4 72.2 -> 67.3, config 2 97.4 -> 93.9; nlp_g unchanged; bit-identical.
    constexpr bool GROUPED = MPX_LOW_TOTALS_GROUPED && !SMALL && MODE == MPX_MODE_FG;
    constexpr int CG = !GROUPED ? 1 : (CHL < MPX_LOW_TOTALS_GROUP ? CHL : MPX_LOW_TOTALS_GROUP);
    double gk[CG][NRED];
    // The lanes of a chunk hold CONSECUTIVE nodes, so every row of g / grad_f leaves as one 512-byte store per chunk straight from
    // the registers (no staging; the mid-point rows are the same run shifted by one node)
    double* __restrict__ gb = want_g ? io.g + (int64_t)b * io.g_stride : nullptr;
    double* __restrict__ qb = want_q ? io.grad + (int64_t)b * io.grad_stride + Q.z_off : nullptr;
#pragma unroll
    for (int u = 0; u < CHL; ++u) {
      const int i = lo_w + 64 * u + l;
      const bool valid = 64 * u + l < len_w;
      const int k = kk[u], pos = (valid ? i : N - 1) - lo_r, pos0 = sg[u] * P - lo_r;  // the node and point 0 of its segment in the span
      Vec<NX> Xs, fx;
      Vec<NU> Us;
      Vec<NC> cc;
#pragma unroll
      for (int a = 0; a < NX; ++a) Xs[a] = sW[a][pos];
#pragma unroll
      for (int c = 0; c < NU; ++c) Us[c] = sW[NX + c][pos];
      const double kap = wsv[u] * A.inv_dtau, th = wcv[u] + wsv[u] * sTk[k];
      Vec<NRED> gr;
      if constexpr (MODE == MPX_MODE_FG) {
        G::fg(Xs, Us, t0v, tfv, As, kap, th, sWt[k], fx, cc, gr[0]);
      } else {
        Vec<NX> dd;
        Vec<G::NJV> jv;
        Vec<NIN> gn;
        G::fgj(Xs, Us, t0v, tfv, As, kap, th, sWt[k], fx, cc, dd, jv, gn, gr);
        if (want_q && valid) {
#pragma unroll
          for (int a = 0; a < NIN; ++a) qb[(int64_t)a * N + i] = gn[a];
        }
      }
      // sums: one partial-sum slot per CHUNK of 64 nodes (chunk index in the phase), wavefront tree over the chunk's nodes -- the
      // same slots and the same additions for every span length, so the short-span and the long-span kernels agree bit for bit
      if constexpr (SMALL || !GROUPED) {
        if (64 * u < len_w) {
#pragma unroll
          for (int r = 0; r < NRED; ++r) {
            const double v = wave_total_dpp(valid ? gr[r] : 0.0);
            if constexpr (SMALL) {
              if (l == 0) io.partial[((int64_t)b * io.n_tiles_total + Q.slot_first + (lo_w >> 6)) * io.nred + r] = v;
            } else {
              tot[r] = u == 0 ? 0.0 + v : tot[r] + v;
            }
          }
        }
      } else {  // the totals of CG chunks together (wave_total_dpp_n), added to the span's sums in chunk order as before
#pragma unroll
        for (int r = 0; r < NRED; ++r) gk[u % CG][r] = valid ? gr[r] : 0.0;
        if (u % CG == CG - 1 || u == CHL - 1) {
          const int u0 = u - u % CG;
#pragma unroll
          for (int r = 0; r < NRED; ++r) {
            double x[CG];
#pragma unroll
            for (int m = 0; m < CG; ++m) x[m] = m <= u % CG ? gk[m][r] : 0.0;
            wave_total_dpp_n<CG>(x);
#pragma unroll
            for (int m = 0; m < CG; ++m)
              if (m <= u % CG && 64 * (u0 + m) < len_w) tot[r] = u0 + m == 0 ? 0.0 + x[m] : tot[r] + x[m];
          }
        }
      }
      if (want_g) {
        double dx[NX > 0 ? NX : 1], du[NU > 0 ? NU : 1], mu[NU > 0 ? NU : 1];
#pragma unroll
        for (int a = 0; a < NX; ++a) {
          double acc = 0;
#pragma unroll
          for (int j = 0; j < P1; ++j) acc = fma(sD[k * P1 + j], sW[a][pos0 + j], acc);
          dx[a] = acc - fx[a];
        }
        if constexpr (G::DIFF_U) {
#pragma unroll
          for (int c = 0; c < NU; ++c) {
            double acc = 0;
#pragma unroll
            for (int j = 0; j < P1; ++j) acc = fma(sD[k * P1 + j], sW[NX + c][pos0 + j], acc);
            du[c] = acc;
          }
        }
        if constexpr (G::MIDU) {
#pragma unroll
          for (int c = 0; c < NU; ++c) {
            double acc = 0;
            if (k >= 1) {
#pragma unroll
              for (int j = 0; j < P1; ++j) acc = fma(sC[(k - 1) * P1 + j], sW[NX + c][pos0 + j], acc);
            }
            mu[c] = acc;
          }
        }
        if (valid) {
#pragma unroll
          for (int a = 0; a < NX; ++a) gb[Q.g_off_F + (int64_t)a * N + i] = dx[a];
#pragma unroll
          for (int jj = 0; jj < NC; ++jj) gb[Q.g_off_C + (int64_t)jj * N + i] = cc[jj];
          if constexpr (G::DIFF_U) {
#pragma unroll
            for (int c = 0; c < NU; ++c) gb[Q.g_off_DU + (int64_t)c * N + i] = du[c];
          }
          if constexpr (G::MIDU) {
            if (k >= 1) {  // (node 0 has no mid-point row)
#pragma unroll
              for (int c = 0; c < NU; ++c) gb[Q.g_off_mU + (int64_t)c * (N - 1) + i - 1] = mu[c];
            }
          }
        }
      }
    }
    MPX_LSTAMP(2)
    if constexpr (!SMALL) {
      if (l == 0) {
#pragma unroll
        for (int r = 0; r < NRED; ++r) io.partial[((int64_t)b * io.n_tiles_total + Q.slot_first + gi) * io.nred + r] = tot[r];
      }
    }
    MPX_LSTAMP(3)
#ifdef MPX_LIGHT_STAMPS
    ++it_;
#endif
  };
  for (int64_t item = (int64_t)light_block_xcd() * MPX_LIGHT_WAVES + wave; item < total; item += stride) {
    const int gi = (int)(item % n_groups);
    if constexpr (NPHK == 1) {
      run_item(std::integral_constant<int, PH0>{}, phases[0], gi, io.b_first + (int)(item / n_groups));
    } else {  // (the phases of one evaluation point next to each other: its z is read once through the caches)
      const int64_t rest = item / n_groups;
      const int ph = (int)(rest % NPHK), b = io.b_first + (int)(rest / NPHK);
      static_for_n<NPHK>([&](auto K) {
        if (ph == decltype(K)::value) run_item(std::integral_constant<int, PH0 + decltype(K)::value>{}, phases[decltype(K)::value], gi, b);
      });
    }
  }
#undef MPX_LSTAMP
}

template <int PH, int P, int MODE, bool SMALL>
__device__ __forceinline__ void light_low_body(const MpxLightArgs& L) {
  const MpxNodeArgs& A = L.node;
  const MpxLightPhase Q{A.z_off, A.g_off_F, A.g_off_C, A.g_off_DU, A.g_off_mU, A.seg_off, L.slot_first};
  light_low_run<P, MODE, SMALL, PH, 1>(L, &Q);
}
template <int P, int MODE, bool SMALL>
__device__ __forceinline__ void light_low_all(const MpxLightMultiArgs& M) {
  light_low_run<P, MODE, SMALL, 0, MPX_NPH>(M.base, M.ph);
}

// ---------------------------------------------------------------------------------------------
// hess_l node pass over node-ordered tiles (mixed-degree grids, MpxHessNodeArgs): lane <-> node i0 + l.  Same arithmetic as the
// MODE_HESS branch of node_body (G::hess, slot layout of scatter_slots, fixed-order tile sums), without anything that depends
// on the degree; segment and reference position of a node come from two per-node tables instead of the bucket's node list.
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------------------------
// Light passes of SINGLE-DEGREE grids of HIGH degree (32 <= P <= 255; round 6): mpx_lighthigh_{fg,fgq}_<ph>_<deg>.
// node_body serves these degrees one evaluation point per walk over the tables: 2 (P + 1)^2 table values from L2 per tile and point
// for (nx + nu) (P + 1) outputs -- nlp_g of moon lander 50 x 100 at B = 512: 319 us, 0.05 of the HBM roofline (profiles/r6_final).
// Here the contraction of a segment IS a matrix product with the EVALUATION POINTS as one dimension:
//     G^T[b][i] = sum_k X^T[b][k] * D^T[k][i]        b: 16 evaluation points, i: the segment's points, k: the segment's points
// on v_mfma_f64_16x16x4_f64 -- A operand = the segment's X / U values of 16 evaluation points (an LDS tile [input][k][point],
// padded to 17), B operand = a 4 x 16 block of the TRANSPOSED differentiation / mid-point table straight from L2 (the lanes of a
// 16-lane row read 16 consecutive entries), accumulated over k in order: the sequential fused chain of node_body, bit for bit
// (tools/mfma_f64_probe.hip; products commute).  A workgroup = (segment, block of 16 evaluation points); its four wavefronts share
// the input tile and take the 16-node column tiles in turn.  In the C/D layout a lane holds ONE node for four evaluation points
// (row = point q + 4 r, column = node n), so the node functions run in place and every row of g / grad_f leaves from registers
// as 128-byte runs (16 consecutive nodes of one point per 16-lane row) -- no output staging.  Each table entry is read once per 16
// points instead of once per point.  g and the node entries of grad_f have node_body's bits; f and the (t0, tf, a) sums are
// added in another fixed order (lane: its nodes in order; 16-lane row by DPP; wavefronts in order; one partial-sum slot per
// segment, the boundary pass adds the segments in order).
// ---------------------------------------------------------------------------------------------------------------------
#ifndef MPX_LH_UNROLL
#define MPX_LH_UNROLL 4  // K-steps whose table loads are requested together
#endif
template <int PH, int P, int MODE>
__device__ __forceinline__ void light_high_body(const MpxLightArgs& L) {
  using G = mpxgen::Phase<PH>;
  const MpxNodeArgs& A = L.node;
  const MpxIO& io = A.io;
  constexpr int NX = G::NX, NU = G::NU, NA = G::NA, NC = G::NC, NIN = NX + NU;
  constexpr int P1 = P + 1, NTN = (P1 + 15) / 16, KS = (P1 + 3) / 4, KP = 4 * KS, LDB = 17;
  constexpr int NRED = (MODE == MPX_MODE_FG) ? 1 : G::NRED;
  extern __shared__ double sT[];  // [NIN][KP][LDB]: X / U of the segment's points k for 16 evaluation points (rows k > P: zero)
  __shared__ double sRed[4][16][NRED];
  __shared__ double sTk[NTN * 16], sWt[NTN * 16];
  const int t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t >> 6), l = t & 63, n = l & 15, q = l >> 4;
  const int s = blockIdx.x, N = A.N, st = s * P;
  const int b0 = io.b_first + 16 * (int)blockIdx.y;
  const int nb = io.B - b0 < 16 ? io.B - b0 : 16;
  // (1) the tile: thread (point b = t % 16, k-pair kc = t / 16) moves two consecutive k of one evaluation point per step
  {
    const int b = t & 15, kc = t >> 4;
    const double* __restrict__ zb = io.z + (int64_t)(b0 + (b < nb ? b : nb - 1)) * io.z_stride + A.z_off + st;
    typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
#pragma unroll
    for (int a = 0; a < NIN; ++a) {
      constexpr int STEPS = (KP + 31) / 32;
      d2u v[STEPS];
#pragma unroll
      // (UNCONDITIONAL loads at clamped addresses, the padding selected afterwards: a guarded load is its own basic block, and the
      // compiler waits for a block's loads before the next block's -- the tile arrived one round trip per step instead of one in all)
      for (int u = 0; u < STEPS; ++u) {
        const int k = 32 * u + 2 * kc, kl = k < P ? k : P - 1;  // (k is even; the pair kl, kl + 1 lies inside the segment)
        v[u] = *(const d2u*)(zb + (int64_t)a * N + kl);
      }
#pragma unroll
      for (int u = 0; u < STEPS; ++u) {
        const int k = 32 * u + 2 * kc;
        const double x0 = k < P ? v[u].x : (k == P ? v[u].y : 0.0), x1 = k + 1 <= P ? v[u].y : 0.0;
        if (k < KP) sT[(a * KP + k) * LDB + b] = x0, sT[(a * KP + k + 1) * LDB + b] = x1;
      }
    }
  }
  if (t < NTN * 16) sTk[t] = A.tk[t <= P ? t : P], sWt[t] = L.wdeg[t <= P ? t : P];
  // the lane's four evaluation points (rows q + 4 r of the products): scalars of the point
  double t0v[4], tfv[4], ws[4], wc[4];
  Vec<NA> As[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int b = q + 4 * r, bb = b0 + (b < nb ? b : nb - 1);
    const double* __restrict__ zt = io.z + (int64_t)bb * io.z_stride + A.z_off + (int64_t)NIN * N;
    t0v[r] = zt[0], tfv[r] = zt[1];
#pragma unroll
    for (int c = 0; c < NA; ++c) As[r][c] = zt[2 + c];
    const int64_t woff = (int64_t)bb * io.w_stride + A.seg_off + s;
    ws[r] = io.w[woff], wc[r] = io.wcum[woff];
  }
  __syncthreads();
  const bool want_g = io.g != nullptr, want_q = MODE == MPX_MODE_FGJ && io.grad != nullptr;
  double racc[4][NRED];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int k = 0; k < NRED; ++k) racc[r][k] = 0.0;
  const double* __restrict__ DT = A.Dmat;  // transposed: DT[k * P1 + i] = D[i][k]
  const double* __restrict__ CT = A.Cmid;  //             CT[k * P + (i - 1)] = C_mid[i - 1][k]
  // The wavefront's column tiles nt = wave, wave + 4, ... (NTW of them): K outermost, so that one LDS read of the A operands serves all
  // of them and the table loads of 4 K-steps x NTW tiles are in flight together (with the tiles outermost every K-step of every tile
  // was its own round trip to L2: 91 us for nlp_g at 50 x 100, B = 512; now the chain is KS / 4 round trips per wavefront).
  constexpr int NTW = (NTN + 3) / 4;
  mpx_d4 aX[NTW][NX], aDU[NTW][NU > 0 ? NU : 1], aCU[NTW][NU > 0 ? NU : 1];
#pragma unroll
  for (int tw = 0; tw < NTW; ++tw) {
#pragma unroll
    for (int a = 0; a < NX; ++a) aX[tw][a] = mpx_d4{0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < (NU > 0 ? NU : 1); ++c) aDU[tw][c] = mpx_d4{0, 0, 0, 0}, aCU[tw][c] = mpx_d4{0, 0, 0, 0};
  }
  // (the number of the wavefront's tiles as a compile-time fact of the loop -- only the last one depends on the wavefront: a uniform
  // branch INSIDE the K loop put the accumulators through phi nodes, and the compiler moved them between the accumulation and the
  // vector registers every K-step, waiting for the matrix pipe to drain each time)
  auto kloop = [&](auto nta_c) {
    constexpr int NTA = decltype(nta_c)::value;
#pragma unroll MPX_LH_UNROLL
    for (int ks = 0; ks < KS; ++ks) {
      const int kk = 4 * ks + q;
      double bD[NTA > 0 ? NTA : 1], bC[NTA > 0 ? NTA : 1];
#pragma unroll
      for (int tw = 0; tw < NTA; ++tw) {
        // (unconditional, clamped: rows k > P of the A tile are zero, so what multiplies them is irrelevant; columns i > P -- and the
        // mid-point column of point 0 -- are never stored)
        const int i = 16 * (wave + 4 * tw) + n, il = i <= P ? i : P, kl = kk <= P ? kk : P;
        bD[tw] = DT[kl * P1 + il];
        bC[tw] = 0.0;
        if constexpr (G::MIDU) bC[tw] = CT[kl * P + (il >= 1 ? il - 1 : 0)];
      }
      double xa[NX], ua[NU > 0 ? NU : 1];
#pragma unroll
      for (int a = 0; a < NX; ++a) xa[a] = sT[(a * KP + kk) * LDB + n];
#pragma unroll
      for (int c = 0; c < NU; ++c) ua[c] = sT[((NX + c) * KP + kk) * LDB + n];
#ifndef MPX_ABL_LH_NO_MFMA  // ablation (wrong results): no matrix instructions
#pragma unroll
      for (int tw = 0; tw < NTA; ++tw) {
#pragma unroll
        for (int a = 0; a < NX; ++a) aX[tw][a] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[a], bD[tw], aX[tw][a], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < NU; ++c) {
          if constexpr (G::DIFF_U) aDU[tw][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(ua[c], bD[tw], aDU[tw][c], 0, 0, 0);
          if constexpr (G::MIDU) aCU[tw][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(ua[c], bC[tw], aCU[tw][c], 0, 0, 0);
        }
      }
#endif
    }
  };
  if (want_g) {
    if (wave + 4 * (NTW - 1) < NTN) kloop(std::integral_constant<int, NTW>{});
    else kloop(std::integral_constant<int, NTW - 1>{});
  }
#pragma unroll
  for (int tw = 0; tw < NTW; ++tw) {
    const int nt = wave + 4 * tw;
    if (nt >= NTN) continue;
    const int i = 16 * nt + n;  // the lane's node of the segment (column n of the products)
    const bool inode = i <= P;
    // (2) node functions in place: node i of evaluation points q, q + 4, q + 8, q + 12
    const bool vnode = inode && (i >= 1 || s == 0);  // (point 0 of a segment belongs to the previous one -- except node 0 of the phase)
    const int ic = inode ? i : P;
    const int64_t node = (int64_t)st + ic;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int b = q + 4 * r;
      const bool valid = vnode && b < nb;
      Vec<NX> Xs, fx;
      Vec<NU> Us;
      Vec<NC> cc;
#pragma unroll
      for (int a = 0; a < NX; ++a) Xs[a] = sT[(a * KP + ic) * LDB + b];
#pragma unroll
      for (int c = 0; c < NU; ++c) Us[c] = sT[((NX + c) * KP + ic) * LDB + b];
      const double kap = ws[r] * A.inv_dtau, th = wc[r] + ws[r] * sTk[ic];
      Vec<NRED> gr;
      Vec<NIN> gn;
      if constexpr (MODE == MPX_MODE_FG) {
        G::fg(Xs, Us, t0v[r], tfv[r], As[r], kap, th, sWt[ic], fx, cc, gr[0]);
      } else {
        Vec<NX> dd;
        Vec<G::NJV> jv;
        G::fgj(Xs, Us, t0v[r], tfv[r], As[r], kap, th, sWt[ic], fx, cc, dd, jv, gn, gr);
      }
      if (valid) {
#pragma unroll
        for (int k = 0; k < NRED; ++k) racc[r][k] += gr[k];
#ifdef MPX_ABL_LH_NO_STORE  // ablation: the rows of g are not stored (unless a value is an impossible one: the arithmetic stays)
        if (want_g && aX[tw][0][r] == 1.2345e300) {
#else
        if (want_g) {
#endif
          double* __restrict__ gb = io.g + (int64_t)(b0 + b) * io.g_stride;
#pragma unroll
          for (int a = 0; a < NX; ++a) gb[A.g_off_F + (int64_t)a * N + node] = aX[tw][a][r] - fx[a];
#pragma unroll
          for (int j = 0; j < NC; ++j) gb[A.g_off_C + (int64_t)j * N + node] = cc[j];
          if constexpr (G::DIFF_U) {
#pragma unroll
            for (int c = 0; c < NU; ++c) gb[A.g_off_DU + (int64_t)c * N + node] = aDU[tw][c][r];
          }
          if constexpr (G::MIDU) {
            if (i >= 1) {
#pragma unroll
              for (int c = 0; c < NU; ++c) gb[A.g_off_mU + (int64_t)c * (N - 1) + (node - 1)] = aCU[tw][c][r];
            }
          }
        }
        if constexpr (MODE == MPX_MODE_FGJ) {
          if (want_q) {
            double* __restrict__ qb = io.grad + (int64_t)(b0 + b) * io.grad_stride + A.z_off;
#pragma unroll
            for (int a = 0; a < NIN; ++a) qb[(int64_t)a * N + node] = gn[a];
          }
        }
      }
    }
  }
  // (3) the points' sums over this segment: 16-lane rows by DPP (row_shr 1, 2, 4, 8: lane 15 of a row holds its total), wavefronts in order
  auto row_total = [](double x) {
    auto dpp0 = [](double v, auto ctrl) {
      constexpr int C = decltype(ctrl)::value;
      const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), C, 0xf, 0xf, false);
      const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), C, 0xf, 0xf, false);
      return __hiloint2double(hi, lo);
    };
    using std::integral_constant;
    x += dpp0(x, integral_constant<int, 0x111>{});
    x += dpp0(x, integral_constant<int, 0x112>{});
    x += dpp0(x, integral_constant<int, 0x114>{});
    x += dpp0(x, integral_constant<int, 0x118>{});
    return x;
  };
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int k = 0; k < NRED; ++k) {
      const double v = row_total(racc[r][k]);
      if (n == 15) sRed[wave][q + 4 * r][k] = v;
    }
  __syncthreads();
  if (t < 16 * NRED) {
    const int b = t / NRED, k = t - b * NRED;
    if (b < nb) {
      double v = 0;
#pragma unroll
      for (int w = 0; w < 4; ++w) v += sRed[w][b][k];
      io.partial[((int64_t)(b0 + b) * io.n_tiles_total + L.slot_first + s) * io.nred + k] = v;
    }
  }
}

template <int PH>
__device__ __forceinline__ void hess_by_node_body(const MpxHessNodeArgs& A) {
  using G = mpxgen::Phase<PH>;
  constexpr int NX = G::NX, NU = G::NU, NA = G::NA, NC = G::NC;
  constexpr int NRED = G::NHC, NRED1 = NRED > 0 ? NRED : 1;
  __shared__ double sRed[2][MPX_TILE / 64][NRED1];
  const unsigned lin_ = blockIdx.y * gridDim.x + blockIdx.x, tot_ = gridDim.x * gridDim.y;  // XCD-blocked walk, as node_body
  const unsigned xcd_ = lin_ % 8, q_ = tot_ / 8, r_ = tot_ % 8;
  const unsigned item_ = xcd_ * q_ + (xcd_ < r_ ? xcd_ : r_) + lin_ / 8;
  const unsigned bx_ = item_ % gridDim.x, by_ = item_ / gridDim.x;
  const MpxHTile T = A.htiles[A.tile_first + bx_];
  const int l = threadIdx.x, lane = l & 63, wave = l >> 6;
  const bool own = l < T.n;
  const int i = T.i0 + (own ? l : 0);
  const int s = A.node_seg[i];
  const double tkk = A.node_tk[i], Wn = A.Wnode[i];
  const int N = A.N;
  const int64_t n = T.n;
  const bool vech = (T.hess_base & 1) == 0;
  const MpxIO& io = A.io;
  constexpr bool ONE_POINT = MPX_HESS_ONE_POINT;  // (one evaluation point per workgroup, compile-time: see node_body)
  const int b0 = io.b_first + (ONE_POINT ? by_ : by_ * io.b_per_block);
  const int b1 = ONE_POINT ? (b0 + 1 < io.B ? b0 + 1 : io.B) : ((b0 + io.b_per_block < io.B) ? b0 + io.b_per_block : io.B);
  const int64_t tslot = (int64_t)T.tile_id * io.nred;
  struct In {
    Vec<NX> Xs;
    Vec<NU> Us;
    Vec<NA> As;
    Vec<NX> lF;
    Vec<NC> lC;
    double t0v, tfv, ws, wc, sig;
  };
  auto load_point = [&](int b, In& q) {
    const double* __restrict__ zb = io.z + (int64_t)b * io.z_stride + A.z_off;
#pragma unroll
    for (int a = 0; a < NX; ++a) q.Xs[a] = (zb + (int64_t)a * N)[i];
#pragma unroll
    for (int c = 0; c < NU; ++c) q.Us[c] = (zb + (int64_t)(NX + c) * N)[i];
    const double* __restrict__ zt = zb + (int64_t)(NX + NU) * N;
    q.t0v = zt[0], q.tfv = zt[1];
#pragma unroll
    for (int c = 0; c < NA; ++c) q.As[c] = zt[2 + c];
    const int64_t woff = (int64_t)b * io.w_stride + A.seg_off + s;
    q.ws = io.w[woff], q.wc = io.wcum[woff];
    const double* __restrict__ lb = io.lam_g + (int64_t)b * io.lam_stride;
#pragma unroll
    for (int a = 0; a < NX; ++a) q.lF[a] = (lb + (A.g_off_F + (int64_t)a * N))[i];
#pragma unroll
    for (int j = 0; j < NC; ++j) q.lC[j] = (lb + (A.g_off_C + (int64_t)j * N))[i];
    q.sig = io.sigma[b];
  };
  In cur, nxt;
  if (b0 < b1) load_point(b0, cur);
  int it = 0;
  for (int b = b0; b < b1; ++b, ++it) {
    const int buf = it & 1;
    if (b + 1 < b1) load_point(b + 1, nxt);  // before this point's stores: memory operations retire in order
    __syncthreads();
    if (it > 0 && l < NRED) {  // publish the previous point's tile sums
      double v = 0;
#pragma unroll
      for (int w = 0; w < MPX_TILE / 64; ++w) v += sRed[buf ^ 1][w][l];
      io.partial[((int64_t)(b - 1) * io.n_tiles_total) * io.nred + tslot + l] = v;
    }
    const double kap = cur.ws * A.inv_dtau, th = cur.wc + cur.ws * tkk;
    Vec<G::NHN> hn;
    Vec<NRED> red;
    G::hess(cur.Xs, cur.Us, cur.t0v, cur.tfv, cur.As, kap, th, Wn, cur.sig, cur.lF, cur.lC, hn, red);
    scatter_slots<G::NHN>(io.hess + (int64_t)b * io.hess_stride + T.hess_base, n, l, own, vech, [&](int q) { return hn[q]; });
#pragma unroll
    for (int r = 0; r < NRED; ++r) {
      double v = wave_sum(own ? red[r] : 0.0);
      if (lane == 0) sRed[buf][wave][r] = v;
    }
    cur = nxt;
  }
  __syncthreads();
  if (it > 0 && l < NRED) {
    double v = 0;
#pragma unroll
    for (int w = 0; w < MPX_TILE / 64; ++w) v += sRed[(it - 1) & 1][w][l];
    io.partial[((int64_t)(b1 - 1) * io.n_tiles_total) * io.nred + tslot + l] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// Off-node evaluation: interpolated states/controls, their polynomial derivatives and the dynamics
// residual  DXi - h_s*Sx*dyn(Xi/Sx, Ui/Su, ti, a)  at arbitrary points of every segment
// (mpopt.interpolate_single_phase / get_dynamics_residuals_single_phase, mpopt.py:1428-1543).
// lane <-> target point; the point's interpolation and derivative rows stay in VGPRs over the batch.
// ---------------------------------------------------------------------------------------------
// Row-major output [row][W] of a wavefront whose 64 lanes hold 64 CONSECUTIVE rows: the wavefront's values are one contiguous
// run of 64*W doubles, so lane l writes elements l, l+64, ... of that run (fetched from the owning lanes by shuffles) instead of
// W stores of stride W*8 bytes per lane.  Anything else (ragged target grids, a partial last wavefront) stores directly.
template <int W>
__device__ __forceinline__ void store_rows(double* __restrict__ out, int64_t row, const double* v, bool valid, bool contig, int64_t row0) {
  if (!out) return;
  if (W == 1 || !contig) {
    if (valid) {
#pragma unroll
      for (int a = 0; a < W; ++a) out[row * W + a] = v[a];
    }
    return;
  }
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < W; ++j) {
    const int e = j * 64 + lane, src = e / W, comp = e % W;
    double x = 0;
#pragma unroll
    for (int c = 0; c < W; ++c) {
      const double t = __shfl(v[c], src, 64);
      if (c == comp) x = t;
    }
    out[row0 * W + e] = x;
  }
}

template <int PH, int P>
__device__ __forceinline__ void resid_body(const MpxResidArgs& A) {
  using G = mpxgen::Phase<PH>;
  constexpr int NX = G::NX, NU = G::NU, NA = G::NA, NC = G::NC, P1 = P + 1;
  const int m_ = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = m_ < A.n;
  const int m = valid ? m_ : A.n - 1;  // idle lanes shadow the last point (they take part in the shuffles of store_rows)
  const int row = A.pt_id[m], s = A.pt_seg[m];
  const int st = A.seg_start[s];
  const double tn = A.pt_tn[m];
  // the wavefront's rows are consecutive (single-degree grids: always, except in the last wavefront)?
  const int row0 = __shfl(row, 0, 64);
  const bool contig = __all(valid && row == row0 + (int)(threadIdx.x & 63));
  // the point's interpolation / derivative rows: registers for the batch loop -- up to degree 32 (2 (P + 1) doubles per lane); above,
  // they are read inside the contraction, one j for every row at once (the same fma chains)
  constexpr bool ROWS_REG = P <= 32;
  double Crow[ROWS_REG ? P1 : 1], Drow[ROWS_REG ? P1 : 1];
  if constexpr (ROWS_REG) {
#pragma unroll
    for (int j = 0; j < P1; ++j) {
      Crow[j] = A.Cmat[(int64_t)j * A.n + m];
      Drow[j] = A.Dmat[(int64_t)j * A.n + m];
    }
  }
  const int N = A.N;
  // Degrees from 8 on (round 6): the node values of the workgroup's segments are staged in LDS once per evaluation point.  Every lane of a
  // segment contracts the SAME P + 1 values per input: read from global memory by each lane itself that is (NX + NU) (P + 1) load instructions
  // per lane and point -- 93 at degree 30 --, and the pass was bound by them (configs[2]: 0.13 of the HBM peak, profiles/r6_resid).  The span
  // [first node of the lowest segment, last node of the highest] of the workgroup's points is a fact of the plan; spans that do not fit (plans
  // with few points per segment) keep the direct loads.  Same values, same fma chains: bit-identical.
  constexpr int NIN = NX + NU;
  constexpr bool STAGE = P >= 8;
  constexpr int ZCAP = STAGE ? (4096 / NIN < 512 ? (4096 / NIN < 64 ? 64 : 4096 / NIN) : 512) : 1;  // (a workgroup of the mid-point grid spans 256 + P nodes)
  __shared__ double sZ[STAGE ? NIN : 1][ZCAP];
  __shared__ int sSeg[2];
  int n0 = 0, span = 0;
  bool staged = false;
  if constexpr (STAGE) {
    if (threadIdx.x == 0) sSeg[0] = 0x7fffffff, sSeg[1] = -1;
    __syncthreads();
    if (valid) atomicMin(&sSeg[0], s), atomicMax(&sSeg[1], s);
    __syncthreads();
    n0 = A.seg_start[sSeg[0]];
    span = A.seg_start[sSeg[1] + 1] - n0 + 1;
    staged = span <= ZCAP;
  }
  const int so = st - n0;  // point 0 of the lane's segment in the staged span
  constexpr int ZQ = (ZCAP + MPX_TILE - 1) / MPX_TILE;
  double pf[STAGE ? NIN : 1][ZQ];
  auto fetch = [&](int bb) {
    const double* __restrict__ zq = A.z + (int64_t)bb * A.z_stride + A.z_off + n0;
#pragma unroll
    for (int a = 0; a < (STAGE ? NIN : 0); ++a)
#pragma unroll
      for (int q = 0; q < ZQ; ++q) {
        const int e = (int)threadIdx.x + q * MPX_TILE;
        pf[a][q] = (zq + (int64_t)a * N)[e < span ? e : span - 1];
      }
  };
  const int b0 = blockIdx.y * A.b_per_block;
  const int b1 = (b0 + A.b_per_block < A.B) ? b0 + A.b_per_block : A.B;
  for (int b = b0; b < b1; ++b) {
    const double* __restrict__ zb = A.z + (int64_t)b * A.z_stride + A.z_off;
    if constexpr (STAGE) {
      if (staged) {  // (uniform)
        // the NEXT point's values are requested before this point's contraction starts and written to LDS after it (unconditional, clamped
        // loads: a guarded prefetch meets its old value in a phi node and is waited for at once; the last point fetches itself again)
        if (b == b0) fetch(b0);
        __syncthreads();  // the previous point's contractions are done with the buffer
#pragma unroll
        for (int a = 0; a < NIN; ++a)
#pragma unroll
          for (int q = 0; q < ZQ; ++q)
            if ((int)threadIdx.x + q * MPX_TILE < span) sZ[a][threadIdx.x + q * MPX_TILE] = pf[a][q];
        __syncthreads();
        fetch(b + 1 < b1 ? b + 1 : b);
      }
    }
    Vec<NX> Xi, DXi, fx, res;
    Vec<NU> Ui, DUi;
    Vec<NA> As;
    Vec<NC> cc;
    // (the staged and the direct form as two instantiations under one uniform branch: with a select per value the compiler kept both
    // address streams alive through the unrolled chains -- 512 registers and scratch at degree 30)
    auto contract = [&](auto from_lds) {
      constexpr bool LDS = decltype(from_lds)::value;
      auto zval = [&](int a, int j) {
        if constexpr (LDS) return sZ[STAGE ? a : 0][so + j];
        else return (zb + (int64_t)a * N)[st + j];
      };
      // (degrees 8 ... 32, spans that do not fit the buffer -- rare: plans with a few points per segment --: the rows are read inside a
      // partially unrolled loop, as above degree 32; fully unrolled with 64-bit addresses per value, that form set the kernel's register count)
      if constexpr (ROWS_REG && (LDS || !STAGE)) {
        // (j outermost: 2 (NX + NU) independent chains advance together and a step needs NX + NU values, where the input-by-input order
        // ran two chains at a time behind P + 1 loads each -- the same chains, term by term)
#pragma unroll
        for (int a = 0; a < NX; ++a) Xi[a] = 0, DXi[a] = 0;
#pragma unroll
        for (int c = 0; c < NU; ++c) Ui[c] = 0, DUi[c] = 0;
#pragma unroll
        for (int j = 0; j < P1; ++j) {
#pragma unroll
          for (int a = 0; a < NX; ++a) {
            const double x = zval(a, j);
            Xi[a] = fma(Crow[j], x, Xi[a]);
            DXi[a] = fma(Drow[j], x, DXi[a]);
          }
#pragma unroll
          for (int c = 0; c < NU; ++c) {
            const double x = zval(NX + c, j);
            Ui[c] = fma(Crow[j], x, Ui[c]);
            DUi[c] = fma(Drow[j], x, DUi[c]);
          }
        }
      } else {
#pragma unroll
        for (int a = 0; a < NX; ++a) Xi[a] = 0, DXi[a] = 0;
#pragma unroll
        for (int c = 0; c < NU; ++c) Ui[c] = 0, DUi[c] = 0;
#pragma unroll 4
        for (int j = 0; j < P1; ++j) {
          const double cj = A.Cmat[(int64_t)j * A.n + m], dj = A.Dmat[(int64_t)j * A.n + m];
#pragma unroll
          for (int a = 0; a < NX; ++a) {
            const double x = zval(a, j);
            Xi[a] = fma(cj, x, Xi[a]);
            DXi[a] = fma(dj, x, DXi[a]);
          }
#pragma unroll
          for (int c = 0; c < NU; ++c) {
            const double x = zval(NX + c, j);
            Ui[c] = fma(cj, x, Ui[c]);
            DUi[c] = fma(dj, x, DUi[c]);
          }
        }
      }
    };
    if (STAGE && staged) contract(std::true_type{});
    else contract(std::false_type{});
    const double* __restrict__ zt = zb + (int64_t)(NX + NU) * N;
    const double t0v = zt[0], tfv = zt[1];
#pragma unroll
    for (int c = 0; c < NA; ++c) As[c] = zt[2 + c];
    const int64_t woff = (int64_t)b * A.w_stride + A.seg_off + s;
    const double ws = A.w[woff], wc = A.wcum[woff];
    const double kap = ws * A.inv_dtau, th = wc + ws * tn;
    double qW;
    G::fg(Xi, Ui, t0v, tfv, As, kap, th, 0.0, fx, cc, qW);
#pragma unroll
    for (int a = 0; a < NX; ++a) res[a] = DXi[a] - fx[a];
    const int64_t ob = (int64_t)b * A.n_pts;
    if (A.ti && valid) A.ti[ob + row] = G::node_time(t0v, tfv, th);
    store_rows<NX>(A.xi ? A.xi + ob * NX : nullptr, row, Xi, valid, contig, row0);
    store_rows<NX>(A.dxi ? A.dxi + ob * NX : nullptr, row, DXi, valid, contig, row0);
    store_rows<NX>(A.dyn ? A.dyn + ob * NX : nullptr, row, fx, valid, contig, row0);
    store_rows<NX>(A.resid ? A.resid + ob * NX : nullptr, row, res, valid, contig, row0);
    if constexpr (NU > 0) {
      store_rows<NU>(A.ui ? A.ui + ob * NU : nullptr, row, Ui, valid, contig, row0);
      store_rows<NU>(A.dui ? A.dui + ob * NU : nullptr, row, DUi, valid, contig, row0);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Boundary kernel: one workgroup per evaluation point.  Finishes the fixed-order reductions,
// evaluates Mayer cost and terminal constraints (mpopt.py:264-300), the linear rows that couple
// segments or phases (control-slope continuity mpopt.py:379-413, events mpopt.py:464-521) and
// scatters their derivative entries.
// ---------------------------------------------------------------------------------------------
constexpr int MAXRED = 64;

// What a phase of the boundary pass reads from memory, requested for EVERY phase before the first of them is waited for (round 5):
// the phase's first 256 partial-sum values (one per lane) and, on lane 0, the inputs of its terminal functions.  The pass is one
// short workgroup per evaluation point and nothing but a chain of dependent round trips -- partial sums, barrier, terminal inputs,
// per phase, one phase after the other: four in a row at config 4 (11 us of a 141 us nlp_g pass at B = 4096, and a quarter of a
// single evaluation); now one.  Same values, same additions, same order.
#ifndef MPX_BOUND_PREFETCH
#define MPX_BOUND_PREFETCH 1
#endif
template <int PH>
struct BoundPhaseIn {
  using G = mpxgen::Phase<PH>;
  Vec<G::NX> XF, X0;
  Vec<G::NA> As;
  double t0v, tfv, pv;
};
template <int PH>
struct BoundIn : BoundIn<PH - 1> {
  BoundPhaseIn<PH> me;
};
template <>
struct BoundIn<-1> {};

template <int PH, int MODE>
__device__ __forceinline__ void boundary_phase_load(const MpxBoundArgs& A, int b, int l, BoundPhaseIn<PH>& I) {
  using G = mpxgen::Phase<PH>;
  constexpr int NX = G::NX, NU = G::NU, NA = G::NA;
  const MpxIO& io = A.io;
  const MpxPhaseInfo& P = A.ph[PH];
  const double* __restrict__ pp = io.partial + ((int64_t)b * io.n_tiles_total + P.tile_first) * io.nred;
  const int nred = io.nred, total = (MODE == MPX_MODE_HESS ? P.tile_count_h : P.tile_count) * nred;
  const int per = (256 / nred) * nred;
  I.pv = l < (total < per ? total : per) ? pp[l] : 0.0;
  if (l == 0) {
    const int N = P.N;
    const double* __restrict__ zb = io.z + (int64_t)b * io.z_stride + P.z_off;
#pragma unroll
    for (int a = 0; a < NX; ++a) I.XF[a] = zb[(int64_t)a * N + N - 1], I.X0[a] = zb[(int64_t)a * N];
    const double* zt = zb + (int64_t)(NX + NU) * N;
    I.t0v = zt[0], I.tfv = zt[1];
#pragma unroll
    for (int c = 0; c < NA; ++c) I.As[c] = zt[2 + c];
  }
}

template <int PH, int MODE>
__device__ __forceinline__ void boundary_phase(const MpxBoundArgs& A, int b, int l, double* red, double& facc, const BoundPhaseIn<PH>* pre = nullptr) {
  using G = mpxgen::Phase<PH>;
  constexpr int NX = G::NX, NU = G::NU, NA = G::NA;
  constexpr int NRED = (MODE == MPX_MODE_FG) ? 1 : (MODE == MPX_MODE_FGJ ? G::NRED : G::NHC);
  static_assert(NRED <= MAXRED, "too many reduced quantities");
  const MpxIO& io = A.io;
  const MpxPhaseInfo& P = A.ph[PH];
  // Sum of the tile partials in tile order (the fixed order every result is defined by).  The loads are what costs: one lane
  // walking 21 ... 101 tiles with one load in flight took longer than the node kernel at a batch of one, so the workgroup fetches
  // the partials 256 values at a time (all lanes, coalesced: the block of a phase is contiguous) into LDS and the NRED summing
  // lanes add their column from there -- same additions, same order.
  {
    __shared__ double sPart[256];
    const double* __restrict__ pp = io.partial + ((int64_t)b * io.n_tiles_total + P.tile_first) * io.nred;
    const int nred = io.nred, total = (MODE == MPX_MODE_HESS ? P.tile_count_h : P.tile_count) * nred;
    const int per = (256 / nred) * nred;  // whole tiles per chunk
    double s = 0;
    // > 1: the slots are added in GROUPS of glen -- a short-span light pass (light_low_body): the sum of a group is what a long span
    // writes into its one slot, so both passes give the same bits.  The groups are independent: lane (group, column) adds its glen
    // slots, then the NRED summing lanes add the group sums in order (8 + 24 dependent additions instead of 188 at config 5).
    const int glen = A.part_group;
    const int n_grp = glen > 1 ? (total / nred + glen - 1) / glen : 0;
    if (glen > 1 && glen <= MPX_LOW_MAX_CHUNKS && n_grp * nred <= 256) {
      if (l < n_grp * nred) {
        const int g = l / nred, r = l - g * nred, cnt = total / nred - g * glen < glen ? total / nred - g * glen : glen;
        double v[MPX_LOW_MAX_CHUNKS];
#pragma unroll
        for (int c = 0; c < MPX_LOW_MAX_CHUNKS; ++c) v[c] = c < cnt ? pp[(g * glen + c) * nred + r] : 0.0;  // (all loads in flight together)
        double tg = 0;
#pragma unroll
        for (int c = 0; c < MPX_LOW_MAX_CHUNKS; ++c)
          if (c < cnt) tg += v[c];
        sPart[l] = tg;
      }
      __syncthreads();
      if (l < NRED)
        for (int g = 0; g < n_grp; ++g) s += sPart[g * nred + l];
    } else {
      double tg = 0;
      int in_group = 0;
      for (int c0 = 0; c0 < total; c0 += per) {
        const int cnt = total - c0 < per ? total - c0 : per;
        if (l < cnt) sPart[l] = (pre && c0 == 0) ? pre->pv : pp[c0 + l];
        __syncthreads();
        if (l < NRED) {
          if (glen > 1) {
            for (int e = l; e < cnt; e += nred) {
              tg += sPart[e];
              if (++in_group == glen) s += tg, tg = 0, in_group = 0;
            }
          } else {
            for (int e = l; e < cnt; e += nred) s += sPart[e];
          }
        }
        __syncthreads();
      }
      if (in_group) s += tg;
    }
    if (l < NRED) red[l] = s;
  }
  __syncthreads();
  if (l == 0) {
    const int N = P.N;
    const double* __restrict__ zb = io.z + (int64_t)b * io.z_stride + P.z_off;
    Vec<NX> XF, X0;
    Vec<NA> As;
    double t0v, tfv;
    if (pre) {
      XF = pre->XF, X0 = pre->X0, As = pre->As, t0v = pre->t0v, tfv = pre->tfv;
    } else {
#pragma unroll
      for (int a = 0; a < NX; ++a) {
        XF[a] = zb[(int64_t)a * N + N - 1];
        X0[a] = zb[(int64_t)a * N];
      }
      const double* zt = zb + (int64_t)(NX + NU) * N;
      t0v = zt[0], tfv = zt[1];
#pragma unroll
      for (int c = 0; c < NA; ++c) As[c] = zt[2 + c];
    }
    double M = 0;
    Vec<G::NTC> tc;
    if constexpr (MODE == MPX_MODE_FG) {
      G::term_fg(XF, tfv, X0, t0v, As, M, tc);
      facc += red[0] + M;
      if (io.g)
        for (int j = 0; j < G::NTC; ++j) io.g[(int64_t)b * io.g_stride + P.g_off_TC + j] = tc[j];
    } else if constexpr (MODE == MPX_MODE_FGJ) {
      Vec<G::NMG> mg;
      Vec<G::NTJ> tj;
      G::term_fgj(XF, tfv, X0, t0v, As, M, tc, mg, tj);
      facc += red[0] + M;
      if (io.g)
        for (int j = 0; j < G::NTC; ++j) io.g[(int64_t)b * io.g_stride + P.g_off_TC + j] = tc[j];
      if (io.grad) {
        double* qb = io.grad + (int64_t)b * io.grad_stride;
        double* qt = qb + P.z_off + (int64_t)(NX + NU) * N;
        for (int r = 0; r < 2 + NA; ++r) qt[r] = red[1 + r];
        for (int e = 0; e < G::NMG; ++e) qb[A.mg_dst[A.mg_off[PH] + e]] += mg[e];
      }
      if (io.jac) {
        double* jb = io.jac + (int64_t)b * io.jac_stride + P.jac_TC;
        for (int e = 0; e < G::NTJ; ++e) jb[e] = tj[e];
      }
    } else {
      Vec<G::NTC> lT;
      const double* lb = io.lam_g + (int64_t)b * io.lam_stride + P.g_off_TC;
      for (int j = 0; j < G::NTC; ++j) lT[j] = lb[j];
      Vec<G::NTH> th;
      G::term_hess(XF, tfv, X0, t0v, As, io.sigma[b], lT, th);
      double* hb = io.hess + (int64_t)b * io.hess_stride;
      for (int e = 0; e < G::NHC; ++e) hb[A.hc_dst[A.hc_off[PH] + e]] = red[e];
      for (int e = 0; e < G::NTH; ++e) {
        const int64_t d = A.th_dst[A.th_off[PH] + e];
        if (d & MPX_ACCUM_BIT)
          hb[d & ~MPX_ACCUM_BIT] += th[e];
        else
          hb[d] = th[e];
      }
    }
  }
  __syncthreads();
}

template <int PH, int MODE>
struct PhaseLoop {
  __device__ static __forceinline__ void run(const MpxBoundArgs& A, int b, int l, double* red, double& facc) {
    PhaseLoop<PH - 1, MODE>::run(A, b, l, red, facc);
    boundary_phase<PH, MODE>(A, b, l, red, facc);
  }
  __device__ static __forceinline__ void load(const MpxBoundArgs& A, int b, int l, BoundIn<PH>& I) {
    PhaseLoop<PH - 1, MODE>::load(A, b, l, I);
    boundary_phase_load<PH, MODE>(A, b, l, I.me);
  }
  __device__ static __forceinline__ void run_pre(const MpxBoundArgs& A, int b, int l, double* red, double& facc, const BoundIn<PH>& I) {
    PhaseLoop<PH - 1, MODE>::run_pre(A, b, l, red, facc, I);
    boundary_phase<PH, MODE>(A, b, l, red, facc, &I.me);
  }
};
template <int MODE>
struct PhaseLoop<-1, MODE> {
  __device__ static __forceinline__ void run(const MpxBoundArgs&, int, int, double*, double&) {}
  __device__ static __forceinline__ void load(const MpxBoundArgs&, int, int, BoundIn<-1>&) {}
  __device__ static __forceinline__ void run_pre(const MpxBoundArgs&, int, int, double*, double&, const BoundIn<-1>&) {}
};

template <int MODE>
__device__ __forceinline__ void boundary_body(const MpxBoundArgs& A, const int res_b = -1) {  // res_b >= 0: called by the resident kernel
  __shared__ double red[MAXRED];
  const int b = res_b >= 0 ? res_b : (int)blockIdx.x, l = threadIdx.x;
  const MpxIO& io = A.io;
  double facc = 0;
  if constexpr (MPX_BOUND_PREFETCH && MPX_NPH <= 4) {  // (registers: ~12 per phase on lane 0's wavefront)
    BoundIn<MPX_NPH - 1> I;
    PhaseLoop<MPX_NPH - 1, MODE>::load(A, b, l, I);
    PhaseLoop<MPX_NPH - 1, MODE>::run_pre(A, b, l, red, facc, I);
  } else {
    PhaseLoop<MPX_NPH - 1, MODE>::run(A, b, l, red, facc);
  }
  if constexpr (MODE != MPX_MODE_HESS) {
    const double* __restrict__ zb = io.z + (int64_t)b * io.z_stride;
    for (int r = l; r < A.n_lin; r += blockDim.x) {
      const int64_t e0 = A.lin_ptr[r], e1 = A.lin_ptr[r + 1];
      double s = 0;
      for (int64_t e = e0; e < e1; ++e) s = fma(A.lin_coef[e], zb[A.lin_idx[e]], s);
      if (io.g) io.g[(int64_t)b * io.g_stride + A.lin_row[r]] = s;
      if constexpr (MODE == MPX_MODE_FGJ) {
        if (io.jac && !io.jac_variable_only) {
          double* jb = io.jac + (int64_t)b * io.jac_stride + A.lin_jac;
          for (int64_t e = e0; e < e1; ++e) jb[e] = A.lin_coef[e];
        }
      }
    }
    if (l == 0 && io.f) io.f[b] = facc;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// nlp_grad: grad_gamma_x = sigma * grad_f + J^T lam_g and grad_gamma_p = d gamma / d (segment widths), gamma = sigma f + lam_g^T g
// -- the sixth oracle ca.nlpsol derives from mpopt's NLP (mpopt.py:757; evaluated once per solve, its -grad_gamma_p is the lam_p
// the solver returns).  No Jacobian is stored: per node
//   * the nonlinear part comes from the generated first derivatives of the node Lagrangian sig*qW - lF.fx + lC.c (G::gradl), which
//     also give d/dkap and d/dth -- the two ways a segment width enters a node (h = (tf - t0) kap, t = t0 + (tf - t0) th,
//     mpopt.py:184-198);
//   * the D / mid-point interpolation blocks are linear rows: column (segment s, point j) of z receives sum_k lam[s, k] * D[k][j],
//     a TRANSPOSED contraction over the segment's multipliers, which the workgroup stages in LDS like node_body stages X / U.  The
//     first point of a segment is the last node of the previous one (mpopt.py:189-195), i.e. another lane's entry: its sum goes to
//     the `halo` staging array and the finishing pass adds it (one addition, fixed order).
// grad_gamma_p[s] = sum_{i in s} (gk_i / (tau1 - tau0) + gth_i * tk_i) + sum_{i in later segments} gth_i: the node pass leaves the two
// per-SEGMENT sums in `pseg` (added from the segment's last node down; segment 0's by the mini-tile of node 0, which holds the whole
// segment), the finishing pass forms the suffix sums in a fixed order.
// Round 6: measured as a batched pass for the first time (tools/r6_nlp_grad_bench.py), configs[1] took 1166 us for 1.5 GB -- longer than the
// whole f + g + grad_f + jac_g pass -- and two thirds of it in the finishing kernel: it read-modify-wrote one `halo` entry per segment into
// grad_gamma_x (3 000 scattered 8-byte updates per evaluation point) and walked the per-node `pnode` pairs three times with a stride of one
// lane's chunk.  Now the lane that owns a segment's last node adds the next segment's column-0 sum itself whenever that segment's lanes
// follow in the same workgroup (same additions in the same order: bit-identical), and the per-segment sums are formed where the node
// terms are -- in the workgroup's LDS.
// ---------------------------------------------------------------------------------------------------------------------
template <int PH, int P>
__device__ __forceinline__ void gradl_body(const MpxGradlArgs& A) {
  using G = mpxgen::Phase<PH>;
  constexpr int NX = G::NX, NU = G::NU, NA = G::NA, NC = G::NC;
  constexpr int P1 = P + 1, SEGS = MPX_TILE / P, SLOTS = SEGS * P1;
  constexpr int NRED = 2 + NA;
  // multiplier rows staged per segment: defect rows, control-slope rows (DIFF_U), mid-point control rows (MIDU)
  constexpr int L_DU = NX, L_MU = NX + (G::DIFF_U ? NU : 0), NL = L_MU + (G::MIDU ? NU : 0);
  // (degrees above MPX_TABLES_STREAM_ABOVE: the tables stay in global memory -- column j = the lane's point k, so the lanes of a
  // segment read consecutive addresses of the ROW-major tables; node_body's streamed mode reads the transposed ones)
  constexpr bool TAB_GLB = P > MPX_TABLES_STREAM_ABOVE;
  __shared__ double sL[NL][SLOTS];
  __shared__ double sD_[TAB_GLB ? 1 : P1 * P1];
  __shared__ double sC_[TAB_GLB ? 1 : P * P1];
  __shared__ double sRed[MPX_TILE / 64][NRED];
  __shared__ double sPn[2][MPX_TILE];  // the lanes' (d gamma_i / d w_s, d gamma_i / d th): summed per segment below
  const double* __restrict__ const sD = TAB_GLB ? A.Dmat : sD_;
  const double* __restrict__ const sC = TAB_GLB ? A.Cmid : sC_;
  const unsigned lin_ = blockIdx.y * gridDim.x + blockIdx.x, tot_ = gridDim.x * gridDim.y;  // XCD-blocked walk, as node_body
  const unsigned xcd_ = lin_ % 8, q_ = tot_ / 8, r_ = tot_ % 8;
  const unsigned item_ = xcd_ * q_ + (xcd_ < r_ ? xcd_ : r_) + lin_ / 8;
  const unsigned bx_ = item_ % gridDim.x, by_ = item_ / gridDim.x;
  const MpxTile T = A.tiles[A.tile_first + bx_];
  const int l = threadIdx.x, lane = l & 63, wave = l >> 6;
  const bool act = l < T.n, own = l < T.n_own;
  const int m = T.m0 + (act ? l : 0);
  const int i = A.node_i[m], sk = A.node_sk[m];
  const int s = sk >> 8, k = sk & 255;
  const int base = ((k == 0) ? 0 : (l - T.node0) / P) * P1;
  const bool halo = act && k == 1 && !T.node0;  // loads the multipliers of the segment's first node (owned by the previous segment)
  // the lane's node is also point 0 of the NEXT segment: if that segment's lanes follow in this workgroup (regular tiles hold whole
  // segments in node order; in a mixed-degree grid the next lane may belong to a segment further on), this lane adds the next segment's
  // column-0 sums to its outputs itself; otherwise the next segment's first lane leaves them in `halo` for the finishing pass
  const bool fuse_next = act && !T.node0 && k == P && l + 1 < T.n && (A.node_sk[m + 1] >> 8) == s + 1;
  const bool halo_out = halo && s >= 1 && !(l >= 1 && (A.node_sk[m - (l >= 1 ? 1 : 0)] >> 8) == s - 1);
  const int N = A.N;
  if constexpr (!TAB_GLB) {
    for (int e = l; e < P1 * P1; e += MPX_TILE) sD_[e] = A.Dmat[e];
    for (int e = l; e < P * P1; e += MPX_TILE) sC_[e] = A.Cmid[e];
  }
  // A.bpb evaluation points per workgroup, one after the other (round 6: one point per workgroup left 86 000 workgroups of a few
  // microseconds each at configs[1], every one of them loading the tables and its tile descriptor again).  The LDS arrays are reused without
  // an extra barrier: sL is read only before the second barrier of a point, sPn / sRed are written only after the first barrier of the next.
  for (int bi = 0; bi < A.bpb; ++bi) {
  const int b = A.b_first + (int)by_ * A.bpb + bi;
  if (b >= A.B) break;  // (uniform)
  const double* __restrict__ zb = A.z + (int64_t)b * A.z_stride + A.z_off;
  const double* __restrict__ lb = A.lam_g + (int64_t)b * A.lam_stride;
  Vec<NX> Xs, lF;
  Vec<NU> Us;
  Vec<NA> As;
  Vec<NC> lC;
#pragma unroll
  for (int a = 0; a < NX; ++a) Xs[a] = (zb + (int64_t)a * N)[i], lF[a] = (lb + (A.g_off_F + (int64_t)a * N))[i];
#pragma unroll
  for (int c = 0; c < NU; ++c) Us[c] = (zb + (int64_t)(NX + c) * N)[i];
#pragma unroll
  for (int j = 0; j < NC; ++j) lC[j] = (lb + (A.g_off_C + (int64_t)j * N))[i];
  const double* __restrict__ zt = zb + (int64_t)(NX + NU) * N;
  const double t0v = zt[0], tfv = zt[1];
#pragma unroll
  for (int c = 0; c < NA; ++c) As[c] = zt[2 + c];
  const int64_t woff = (int64_t)b * A.w_stride + A.seg_off + s;
  const double ws = A.w[woff], wc = A.wcum[woff];
  const double tkk = A.tk[k], Wn = A.Wnode[i];
  if (act) {
#pragma unroll
    for (int a = 0; a < NX; ++a) sL[a][base + k] = lF[a];
    if constexpr (G::DIFF_U) {
#pragma unroll
      for (int c = 0; c < NU; ++c) sL[L_DU + c][base + k] = (lb + (A.g_off_DU + (int64_t)c * N))[i];
    }
    if constexpr (G::MIDU) {  // the row of the mid-point before node i is i - 1 (mpopt.py:350-369); node 0 has none
#pragma unroll
      for (int c = 0; c < NU; ++c) sL[L_MU + c][base + k] = k >= 1 ? (lb + (A.g_off_mU + (int64_t)c * (N - 1)))[i - 1] : 0.0;
    }
    if (halo) {
#pragma unroll
      for (int a = 0; a < NX; ++a) sL[a][base] = (lb + (A.g_off_F + (int64_t)a * N))[i - 1];
      if constexpr (G::DIFF_U) {
#pragma unroll
        for (int c = 0; c < NU; ++c) sL[L_DU + c][base] = (lb + (A.g_off_DU + (int64_t)c * N))[i - 1];
      }
      if constexpr (G::MIDU) {
#pragma unroll
        for (int c = 0; c < NU; ++c) sL[L_MU + c][base] = 0.0;
      }
    }
  }
  __syncthreads();
  const double kap = ws * A.inv_dtau, th = wc + ws * tkk;
  Vec<NX + NU> gx;
  Vec<NRED> gr;
  double gk, gth;
  G::gradl(Xs, Us, t0v, tfv, As, kap, th, Wn, A.sigma[b], lF, lC, gx, gr, gk, gth);
  // rows of the segment that reach column j: points 1..P, and point 0 in segment 0 only (node 0 owns the first row of D there;
  // in every later segment that row belongs to the previous segment's block, mpopt.py:4035-4038)
  const int klo = s == 0 ? 0 : 1;
  auto col_D = [&](int row, int j) {
    double acc = 0;
#pragma unroll 4
    for (int kp = klo; kp < P1; ++kp) acc = fma(sL[row][base + kp], sD[kp * P1 + j], acc);
    return acc;
  };
  auto col_C = [&](int row, int j) {
    double acc = 0;
#pragma unroll 4
    for (int kp = 1; kp < P1; ++kp) acc = fma(sL[row][base + kp], sC[(kp - 1) * P1 + j], acc);
    return acc;
  };
  // column 0 of the NEXT segment's block (its rows 1 ... P: the next segment is never segment 0): the same chains as that segment's
  // first lane runs for `halo`
  auto next_D = [&](int row) {
    double acc = 0;
#pragma unroll 4
    for (int kp = 1; kp < P1; ++kp) acc = fma(sL[row][base + P1 + kp], sD[kp * P1], acc);
    return acc;
  };
  auto next_C = [&](int row) {
    double acc = 0;
#pragma unroll 4
    for (int kp = 1; kp < P1; ++kp) acc = fma(sL[row][base + P1 + kp], sC[(kp - 1) * P1], acc);
    return acc;
  };
  if (own && A.gx) {
    double* __restrict__ gb = A.gx + (int64_t)b * A.gx_stride + A.z_off;
#pragma unroll
    for (int a = 0; a < NX; ++a) {
      double v = gx[a] + col_D(a, k);
      if (fuse_next) v += next_D(a);  // (what the finishing pass added from `halo`: one addition, after the node's own sum)
      gb[(int64_t)a * N + i] = v;
    }
#pragma unroll
    for (int c = 0; c < NU; ++c) {
      double v = gx[NX + c];
      if constexpr (G::DIFF_U) v += col_D(L_DU + c, k);
      if constexpr (G::MIDU) v += col_C(L_MU + c, k);
      if (fuse_next) {
        double h = 0;
        if constexpr (G::DIFF_U) h += next_D(L_DU + c);
        if constexpr (G::MIDU) h += next_C(L_MU + c);
        v += h;
      }
      gb[(int64_t)(NX + c) * N + i] = v;
    }
    if (halo_out) {
      double* __restrict__ hb = A.halo + ((int64_t)b * A.S * MPX_NPH + A.seg_off + s) * (NX + NU);
#pragma unroll
      for (int a = 0; a < NX; ++a) hb[a] = col_D(a, 0);
#pragma unroll
      for (int c = 0; c < NU; ++c) {
        double v = 0;
        if constexpr (G::DIFF_U) v += col_D(L_DU + c, 0);
        if constexpr (G::MIDU) v += col_C(L_MU + c, 0);
        hb[NX + c] = v;
      }
    }
  }
  sPn[0][l] = fma(gth, tkk, gk * A.inv_dtau);
  sPn[1][l] = gth;
#pragma unroll
  for (int r = 0; r < NRED; ++r) {
    const double v = wave_sum(own ? gr[r] : 0.0);
    if (lane == 0) sRed[wave][r] = v;
  }
  __syncthreads();
  // the segment's sums, from its last node down (then node 0 for segment 0: the order the finishing pass used to add the per-node values
  // in).  Segment 0 is summed by the mini-tile of node 0 -- lanes 0 ... P hold node 0 and the whole segment there --, every other
  // segment by the regular tile that owns its nodes.
  if (act && k == P && (T.node0 || s != 0)) {
    double v0 = 0, v1 = 0;
#pragma unroll 4
    for (int kk = 0; kk < P; ++kk) v0 += sPn[0][l - kk], v1 += sPn[1][l - kk];
    if (T.node0) v0 += sPn[0][0], v1 += sPn[1][0];
    double* __restrict__ ps = A.pseg + (((int64_t)b * MPX_NPH + A.phase) * A.S + s) * 2;
    ps[0] = v0, ps[1] = v1;
  }
  if (l < NRED) {
    double v = 0;
#pragma unroll
    for (int w = 0; w < MPX_TILE / 64; ++w) v += sRed[w][l];
    A.partial[((int64_t)b * A.n_tiles_total + T.tile_id) * A.nred + l] = v;
  }
  }  // (evaluation points of the workgroup)
}

template <int PH>
__device__ __forceinline__ void gradl_finish_phase(const MpxGradlFinArgs& A, int b, int l, double* red, double* sScan) {
  using G = mpxgen::Phase<PH>;
  constexpr int NX = G::NX, NU = G::NU, NA = G::NA;
  constexpr int NRED = 2 + NA;
  static_assert(NRED <= MAXRED, "too many reduced quantities");
  const MpxPhaseInfo& P = A.ph[PH];
  const int N = P.N, S = A.S;
  {  // (t0, tf, A) sums of the tile partials, in tile order (as boundary_phase)
    __shared__ double sPart[256];
    const double* __restrict__ pp = A.partial + ((int64_t)b * A.n_tiles_total + P.tile_first) * A.nred;
    const int nred = A.nred, total = P.tile_count * nred;
    const int per = (256 / nred) * nred;
    double s = 0;
    for (int c0 = 0; c0 < total; c0 += per) {
      const int cnt = total - c0 < per ? total - c0 : per;
      if (l < cnt) sPart[l] = pp[c0 + l];
      __syncthreads();
      if (l < NRED)
        for (int e = l; e < cnt; e += nred) s += sPart[e];
      __syncthreads();
    }
    if (l < NRED) red[l] = s;
  }
  __syncthreads();
  if (A.gx) {
    double* __restrict__ gb = A.gx + (int64_t)b * A.gx_stride + P.z_off;
    if (l == 0) {  // terminal cost / constraints (mpopt.py:277-298): d (sig M + lT . tc) / d (XF, tf, X0, t0, A)
      const double* __restrict__ zb = A.z + (int64_t)b * A.z_stride + P.z_off;
      Vec<NX> XF, X0;
      Vec<NA> As;
      Vec<G::NTC> lT;
#pragma unroll
      for (int a = 0; a < NX; ++a) XF[a] = zb[(int64_t)a * N + N - 1], X0[a] = zb[(int64_t)a * N];
      const double* zt = zb + (int64_t)(NX + NU) * N;
#pragma unroll
      for (int c = 0; c < NA; ++c) As[c] = zt[2 + c];
      const double* lb = A.lam_g + (int64_t)b * A.lam_stride + P.g_off_TC;
      for (int j = 0; j < G::NTC; ++j) lT[j] = lb[j];
      Vec<2 * NX + 2 + NA> tg;
      G::term_gradl(XF, zt[1], X0, zt[0], As, A.sigma[b], lT, tg);
      for (int a = 0; a < NX; ++a) gb[(int64_t)a * N + N - 1] += tg[a], gb[(int64_t)a * N] += tg[NX + 1 + a];
      double* gt = gb + (int64_t)(NX + NU) * N;
      gt[0] = red[0] + tg[2 * NX + 1];
      gt[1] = red[1] + tg[NX];
      for (int c = 0; c < NA; ++c) gt[2 + c] = red[2 + c] + tg[2 * NX + 2 + c];
    }
    // first node of segment s >= 1 = last node of segment s - 1: add what the rows of segment s contribute to its columns -- for the
    // segments whose predecessor's lanes were not in the same workgroup of the node pass (the others were added there)
    const double* __restrict__ hb = A.halo + ((int64_t)b * S * MPX_NPH + (int64_t)PH * S) * (NX + NU);
    const int h0 = A.halo_off[PH], nh = A.halo_off[PH + 1] - h0;
    for (int e = l; e < nh * (NX + NU); e += 256) {
      const int s = A.halo_seg[h0 + e / (NX + NU)], r = e % (NX + NU);
      gb[(int64_t)r * N + A.seg_start[s]] += hb[(int64_t)s * (NX + NU) + r];
    }
  }
  if (A.gp) {
    // grad_gamma_p[s] = (sum of pnode[i][0] over the nodes segment s owns) + (sum of pnode[i][1] over the nodes of all LATER segments):
    // a lane owns a contiguous chunk of segments, chunk totals meet in LDS, every sum runs from the last segment down
    const double* __restrict__ pn = A.pseg + ((int64_t)b * MPX_NPH + PH) * (int64_t)S * 2;
    double* __restrict__ gp = A.gp + (int64_t)b * A.gp_stride + (int64_t)PH * S;
    const int chunk = (S + 255) / 256, s0 = l * chunk < S ? l * chunk : S, s1 = s0 + chunk < S ? s0 + chunk : S;
    auto seg_sum = [&](int s, int which) { return pn[(int64_t)s * 2 + which]; };  // (formed by the node pass, from the segment's last node down)
    // (up to 16 segments per lane -- S <= 4096, every BASELINE grid --: the lane's pairs are read ONCE, 16 bytes at a time, and kept in
    // registers; read value by value in both loops, configs[4] (S = 4000) spent 885 us here on lines fetched eight times over)
    constexpr int CH = 16;
    typedef double d2a __attribute__((ext_vector_type(2), aligned(16)));
    const d2a* __restrict__ pn2 = reinterpret_cast<const d2a*>(pn);
    d2a v[CH];
    const bool cached = chunk <= CH;
    if (cached) {
#pragma unroll
      for (int j = 0; j < CH; ++j) v[j] = s0 + j < s1 ? pn2[s0 + j] : d2a{0.0, 0.0};
    }
    double tot = 0;
    if (cached) {
#pragma unroll
      for (int j = CH - 1; j >= 0; --j)
        if (s0 + j < s1) tot += v[j].y;
    } else {
      for (int s = s1 - 1; s >= s0; --s) tot += seg_sum(s, 1);
    }
    sScan[l] = tot;
    __syncthreads();
    double off = 0;
    for (int q = 255; q > l; --q) off += sScan[q];
    if (cached) {
#pragma unroll
      for (int j = CH - 1; j >= 0; --j)
        if (s0 + j < s1) {
          gp[s0 + j] = v[j].x + off;
          off += v[j].y;
        }
    } else {
      for (int s = s1 - 1; s >= s0; --s) {
        gp[s] = seg_sum(s, 0) + off;
        off += seg_sum(s, 1);
      }
    }
  }
  __syncthreads();
}

template <int PH>
struct GradlFinLoop {
  __device__ static __forceinline__ void run(const MpxGradlFinArgs& A, int b, int l, double* red, double* sScan) {
    GradlFinLoop<PH - 1>::run(A, b, l, red, sScan);
    gradl_finish_phase<PH>(A, b, l, red, sScan);
  }
};
template <>
struct GradlFinLoop<-1> {
  __device__ static __forceinline__ void run(const MpxGradlFinArgs&, int, int, double*, double*) {}
};

__device__ __forceinline__ void gradl_finish_body(const MpxGradlFinArgs& A) {
  __shared__ double red[MAXRED];
  __shared__ double sScan[256];
  const int b = blockIdx.x, l = threadIdx.x;
  GradlFinLoop<MPX_NPH - 1>::run(A, b, l, red, sScan);
  if (A.gx) {  // J^T lam_g of the linear rows (control-slope continuity, phase events), one lane per column they touch
    double* __restrict__ gb = A.gx + (int64_t)b * A.gx_stride;
    const double* __restrict__ lb = A.lam_g + (int64_t)b * A.lam_stride;
    for (int c = l; c < A.n_lt; c += 256) {
      double s = 0;
      for (int64_t e = A.lt_ptr[c]; e < A.lt_ptr[c + 1]; ++e) s = fma(lb[A.lt_row[e]], A.lt_coef[e], s);
      gb[A.lt_col[c]] += s;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Resident kernel (MpxResidentArgs, mpx_device.h): single evaluations without a launch.  `dispatch(A, mode, bx)` runs the node pass
// of tile bx of the bucket A describes (generated: a chain of node_body<PH, P, MODE> instantiations), `bound(G, mode)` the
// boundary pass.  All n_tiles workgroups are resident at once (the host launches only then).
// ---------------------------------------------------------------------------------------------------------------------
template <bool SYSTEM>
__device__ __forceinline__ void res_grid_sync(const MpxResidentArgs& R, unsigned long long& epoch) {
  if (SYSTEM) __threadfence_system();  // (the last barrier of a request: outputs may live in mapped host memory)
  else __threadfence();
  __syncthreads();
  ++epoch;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(R.sync_count, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long target = epoch * (unsigned long long)R.n_tiles;
    while (__hip_atomic_load(R.sync_count, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {}
  }
  __syncthreads();
}

template <class Dispatch, class Bound>
__device__ __forceinline__ void resident_loop(const MpxResidentArgs& R, Dispatch dispatch, Bound bound) {
  __shared__ unsigned long long s_seq;
  const int t = blockIdx.x;
  constexpr unsigned long long EXIT = ~0ull;
  unsigned long long seen = R.start_seq, epoch = 0;  // (request WORDS: number << 8 | slot << 1 | new)
  // the static arguments of this workgroup's bucket and of the boundary pass: fetched ONCE (a per-request copy of the 0.7 KB cost
  // 6.3 of the 20 us of a request, tools/r4_resident_stamps.py); the request's io comes through LDS
  MpxNodeArgs A = R.buckets[R.tile_bucket[t]];
  __shared__ MpxResRequest sQ;
  const long long t_start = wall_clock64();
  long long t_idle = t_start;
  for (;;) {
    if (threadIdx.x == 0) {
      unsigned long long q = seen;
      if (t == 0) {
        for (;;) {
          q = __hip_atomic_load(&R.box->seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
          if (q != seen) break;
          const long long now = wall_clock64();
          if (now - t_idle > R.idle_ticks || now - t_start > R.life_ticks ||
              __hip_atomic_load(&R.box->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) {
            // leave: tell the host first, then look once more (the host writes seq before it reads alive)
            __hip_atomic_store(&R.box->alive, 0u, __ATOMIC_SEQ_CST, __HIP_MEMORY_SCOPE_SYSTEM);
            __threadfence_system();
            q = __hip_atomic_load(&R.box->seq, __ATOMIC_SEQ_CST, __HIP_MEMORY_SCOPE_SYSTEM);
            if (q == seen) q = EXIT;
            else __hip_atomic_store(&R.box->alive, 1u, __ATOMIC_SEQ_CST, __HIP_MEMORY_SCOPE_SYSTEM);  // (a request came in after all: stay)
            break;
          }
          __builtin_amdgcn_s_sleep(2);
        }
        if (q != EXIT && (q & 1)) {  // new content of the slot: fetch it from the mailbox once
          const int sl = (int)((q >> 1) & 127);
          R.dev_slots[sl] = R.box->slots[sl];
          __threadfence();
        }
        __hip_atomic_store(R.dev_seq, q, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        while ((q = __hip_atomic_load(R.dev_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) == seen) __builtin_amdgcn_s_sleep(1);
      }
      s_seq = q;
    }
    __syncthreads();
    const unsigned long long q = s_seq;
    __syncthreads();
    if (q == EXIT) break;
    seen = q;
#ifdef MPX_RES_STAMPS  // phase stamps of workgroup 0 (wall_clock64, 100 MHz) into the mailbox's last slot (diagnostics builds)
#define MPX_RSTAMP(k) if (t == 0 && threadIdx.x == 0) reinterpret_cast<long long*>(&R.box->slots[MPX_RES_SLOTS - 1])[k] = wall_clock64();
#else
#define MPX_RSTAMP(k)
#endif
    MPX_RSTAMP(0)
    {
      const int* __restrict__ src = reinterpret_cast<const int*>(&R.dev_slots[(q >> 1) & 127]);
      int* dst = reinterpret_cast<int*>(&sQ);
      for (int e = threadIdx.x; e < (int)(sizeof(MpxResRequest) / 4); e += blockDim.x) dst[e] = src[e];
    }
    __syncthreads();
    const MpxResRequest& Q = sQ;
    const int mode = Q.mode, ccs = Q.ccs;
    A.io = Q.io;
    MPX_RSTAMP(1)
    dispatch(A, mode, t - A.tile_first);
    MPX_RSTAMP(2)
    res_grid_sync<false>(R, epoch);
    MPX_RSTAMP(3)
    if (t == 0) {
      __shared__ MpxBoundArgs sGb;
      {
        const int* __restrict__ src = reinterpret_cast<const int*>(R.bound);
        int* dst = reinterpret_cast<int*>(&sGb);
        for (int e = threadIdx.x; e < (int)(sizeof(MpxBoundArgs) / 4); e += blockDim.x) dst[e] = src[e];
      }
      __syncthreads();
      if (threadIdx.x == 0) sGb.io = A.io;
      __syncthreads();
      bound(sGb, mode);
    }
    MPX_RSTAMP(4)
    if (ccs) {  // values in compressed-column order: out[k] = native[perm[k]], every workgroup a slice (coalesced writes)
      res_grid_sync<false>(R, epoch);
      const bool hs = mode == MPX_MODE_HESS;
      const double* __restrict__ src = hs ? A.io.hess : A.io.jac;
      const int64_t* __restrict__ perm = hs ? R.perm_h : R.perm_j;
      const int64_t nnz = hs ? R.nnz_h : R.nnz_j;
      double* __restrict__ dst = Q.ccs_out;
      for (int64_t k = (int64_t)t * blockDim.x + threadIdx.x; k < nnz; k += (int64_t)R.n_tiles * blockDim.x) dst[k] = src[perm[k]];
    }
    res_grid_sync<true>(R, epoch);
    MPX_RSTAMP(5)
    if (t == 0 && threadIdx.x == 0) __hip_atomic_store(&R.box->done, q >> 8, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    t_idle = wall_clock64();
  }
  if (t == 0 && threadIdx.x == 0) {
    __threadfence_system();
    __hip_atomic_store(&R.box->exited, 1u, __ATOMIC_SEQ_CST, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
}  // namespace mpxk

#ifndef MPX_MIN_WAVES
#define MPX_MIN_WAVES 1
#endif
#ifndef MPX_MIN_WAVES_HIGH  // degrees with the tables in LDS: the register budget that gives 4 wavefronts per SIMD (<= 128 VGPRs)
#define MPX_MIN_WAVES_HIGH 4
#endif
#ifndef MPX_MIN_WAVES_STREAM  // streamed tables: two chunks of table values are live on top; 4 spilled 35 dwords, 3 (168 VGPRs) measured 8-19 % slower than 2 at degrees 72-80 (tools/r6_stream_ab.py)
#define MPX_MIN_WAVES_STREAM 2
#endif
#define MPX_WAVES_FOR(P) ((P) > MPX_TABLES_STREAM_ABOVE ? MPX_MIN_WAVES_STREAM : (P) > MPX_TABLES_IN_LDS_ABOVE ? MPX_MIN_WAVES_HIGH : MPX_MIN_WAVES)
#define MPX_INSTANTIATE_NODE(PH, P)                                                                         \
  extern "C" __global__ __launch_bounds__(MPX_TILE, MPX_MIN_WAVES) void mpx_node_fg_##PH##_##P(const MpxNodeArgs A) {      \
    mpxk::node_body<PH, P, MPX_MODE_FG>(A);                                                                 \
  }                                                                                                         \
  extern "C" __global__ __launch_bounds__(MPX_TILE, MPX_WAVES_FOR(P)) void mpx_node_fgj_##PH##_##P(const MpxNodeArgs A) {  \
    mpxk::node_body<PH, P, MPX_MODE_FGJ>(A);                                                                \
  }                                                                                                         \
  extern "C" __global__ __launch_bounds__(MPX_TILE, MPX_MIN_WAVES) void mpx_node_hess_##PH##_##P(const MpxNodeArgs A) {    \
    mpxk::node_body<PH, P, MPX_MODE_HESS>(A);                                                               \
  }                                                                                                         \
  extern "C" __global__ __launch_bounds__(MPX_TILE) void mpx_resid_##PH##_##P(const MpxResidArgs A) {       \
    mpxk::resid_body<PH, P>(A);                                                                             \
  }

#define MPX_INSTANTIATE_HESS_BY_NODE(PH)                                                                    \
  extern "C" __global__ __launch_bounds__(MPX_TILE, MPX_MIN_WAVES) void mpx_node_hessn_##PH(const MpxHessNodeArgs A) {  \
    mpxk::hess_by_node_body<PH>(A);                                                                         \
  }

#define MPX_INSTANTIATE_LIGHT(PH, P) MPX_INSTANTIATE_LIGHT_PF(PH, P, 0)
// (PF: the grid's one low degree, 0 if it has none or several -- light_body)
#ifndef MPX_LIGHT_MIN_WG
#define MPX_LIGHT_MIN_WG 2  // workgroups per compute unit the light kernels are compiled for (A/B with MPX_LIGHT_SEGS / MPX_LIGHT_PER_CU)
#endif
#define MPX_INSTANTIATE_LIGHT_PF(PH, P, PF)                                                                                   \
  extern "C" __global__ __launch_bounds__(64 * MPX_LIGHT_WAVES, MPX_LIGHT_MIN_WG) void mpx_light_fg_##PH##_##P(const MpxLightArgs A) {          \
    mpxk::light_body<PH, P, MPX_MODE_FG, PF>(A);                                                                              \
  }                                                                                                                           \
  extern "C" __global__ __launch_bounds__(64 * MPX_LIGHT_WAVES, MPX_LIGHT_MIN_WG) void mpx_light_fgq_##PH##_##P(const MpxLightArgs A) {         \
    mpxk::light_body<PH, P, MPX_MODE_FGJ, PF>(A);                                                                             \
  }

#define MPX_INSTANTIATE_LIGHT_HIGH(PH, P)                                                                                     \
  extern "C" __global__ __launch_bounds__(256) void mpx_lighthigh_fg_##PH##_##P(const MpxLightArgs A) {                        \
    mpxk::light_high_body<PH, P, MPX_MODE_FG>(A);                                                                             \
  }                                                                                                                           \
  extern "C" __global__ __launch_bounds__(256) void mpx_lighthigh_fgq_##PH##_##P(const MpxLightArgs A) {                       \
    mpxk::light_high_body<PH, P, MPX_MODE_FGJ>(A);                                                                            \
  }

// (two workgroups per compute unit: three measured 8 - 13 % slower at config 2, one 45 % -- profiles/r4_lightlow/README.md)
#define MPX_LOW_WPS(PH) 2
#define MPX_INSTANTIATE_LIGHT_LOW(PH, P)                                                                                      \
  extern "C" __global__ __launch_bounds__(64 * MPX_LIGHT_WAVES, MPX_LOW_WPS(PH)) void mpx_lightlow_fg_##PH##_##P(const MpxLightArgs A) {    \
    mpxk::light_low_body<PH, P, MPX_MODE_FG, false>(A);                                                                       \
  }                                                                                                                           \
  extern "C" __global__ __launch_bounds__(64 * MPX_LIGHT_WAVES, MPX_LOW_WPS(PH)) void mpx_lightlow_fgq_##PH##_##P(const MpxLightArgs A) {   \
    mpxk::light_low_body<PH, P, MPX_MODE_FGJ, false>(A);                                                                      \
  }                                                                                                                           \
  extern "C" __global__ __launch_bounds__(64 * MPX_LIGHT_WAVES, MPX_LOW_WPS(PH)) void mpx_lightlows_fg_##PH##_##P(const MpxLightArgs A) {   \
    mpxk::light_low_body<PH, P, MPX_MODE_FG, true>(A);                                                                        \
  }                                                                                                                           \
  extern "C" __global__ __launch_bounds__(64 * MPX_LIGHT_WAVES, MPX_LOW_WPS(PH)) void mpx_lightlows_fgq_##PH##_##P(const MpxLightArgs A) {  \
    mpxk::light_low_body<PH, P, MPX_MODE_FGJ, true>(A);                                                                       \
  }

// all phases of a single-degree grid in one launch (generated for OCPs with more than one phase; MpxNodeMultiArgs / MpxLightMultiArgs)
#define MPX_INSTANTIATE_NODE_ALL(P)                                                                                            \
  extern "C" __global__ __launch_bounds__(MPX_TILE, MPX_MIN_WAVES) void mpx_node_fg_all_##P(const MpxNodeMultiArgs M) {        \
    mpxk::node_all<P, MPX_MODE_FG>(M);                                                                                         \
  }                                                                                                                            \
  extern "C" __global__ __launch_bounds__(MPX_TILE, MPX_WAVES_FOR(P)) void mpx_node_fgj_all_##P(const MpxNodeMultiArgs M) {    \
    mpxk::node_all<P, MPX_MODE_FGJ>(M);                                                                                        \
  }                                                                                                                            \
  extern "C" __global__ __launch_bounds__(MPX_TILE, MPX_MIN_WAVES) void mpx_node_hess_all_##P(const MpxNodeMultiArgs M) {      \
    mpxk::node_all<P, MPX_MODE_HESS>(M);                                                                                       \
  }
#define MPX_INSTANTIATE_LIGHT_LOW_ALL(P)                                                                                       \
  extern "C" __global__ __launch_bounds__(64 * MPX_LIGHT_WAVES, MPX_LOW_WPS(0)) void mpx_lightlow_fg_all_##P(const MpxLightMultiArgs M) {   \
    mpxk::light_low_all<P, MPX_MODE_FG, false>(M);                                                                             \
  }                                                                                                                            \
  extern "C" __global__ __launch_bounds__(64 * MPX_LIGHT_WAVES, MPX_LOW_WPS(0)) void mpx_lightlow_fgq_all_##P(const MpxLightMultiArgs M) {  \
    mpxk::light_low_all<P, MPX_MODE_FGJ, false>(M);                                                                            \
  }                                                                                                                            \
  extern "C" __global__ __launch_bounds__(64 * MPX_LIGHT_WAVES, MPX_LOW_WPS(0)) void mpx_lightlows_fg_all_##P(const MpxLightMultiArgs M) {  \
    mpxk::light_low_all<P, MPX_MODE_FG, true>(M);                                                                              \
  }                                                                                                                            \
  extern "C" __global__ __launch_bounds__(64 * MPX_LIGHT_WAVES, MPX_LOW_WPS(0)) void mpx_lightlows_fgq_all_##P(const MpxLightMultiArgs M) { \
    mpxk::light_low_all<P, MPX_MODE_FGJ, true>(M);                                                                             \
  }

#define MPX_INSTANTIATE_GRADL(PH, P)                                                                        \
  extern "C" __global__ __launch_bounds__(MPX_TILE) void mpx_node_gradl_##PH##_##P(const MpxGradlArgs A) {  \
    mpxk::gradl_body<PH, P>(A);                                                                             \
  }

#define MPX_INSTANTIATE_BOUNDARY()                                                                          \
  extern "C" __global__ __launch_bounds__(256) void mpx_gradl_finish(const MpxGradlFinArgs A) {             \
    mpxk::gradl_finish_body(A);                                                                             \
  }                                                                                                         \
  extern "C" __global__ __launch_bounds__(256) void mpx_boundary_fg(const MpxBoundArgs A) {                 \
    mpxk::boundary_body<MPX_MODE_FG>(A);                                                                    \
  }                                                                                                         \
  extern "C" __global__ __launch_bounds__(256) void mpx_boundary_fgj(const MpxBoundArgs A) {                \
    mpxk::boundary_body<MPX_MODE_FGJ>(A);                                                                   \
  }                                                                                                         \
  extern "C" __global__ __launch_bounds__(256) void mpx_boundary_hess(const MpxBoundArgs A) {               \
    mpxk::boundary_body<MPX_MODE_HESS>(A);                                                                  \
  }

// (read by libmpx when it loads the code object: mpx_host.cpp, load_device)
extern "C" __device__ __attribute__((used)) const int mpx_tables_stream_above = MPX_TABLES_STREAM_ABOVE;
