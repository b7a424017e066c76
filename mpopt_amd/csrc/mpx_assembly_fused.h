// mpx_assembly_fused.h -- fused, persistent evaluation kernels of assembled contexts (include/mpx.h, mpx_create_assembled).
//
// Round 3 replacement of the two-pass scheme (mpx_assembly_kernels.h point kernels -> raw buffer in HBM -> mpx_gather_kernel)
// for batches: the counters of profiles/r3_adaptive show that pass bound by the instruction stream of a table interpreter
// (~80 instructions per stored wavefront-row, a three-level dependent load chain per row) with the raw values written to and
// read back from HBM on top (traffic 1.88x the algorithmic bytes).  Here
//   * one workgroup evaluates U evaluation points at a time, entirely out of LDS:  V[u] = [raw point values | z | 1.0],
//     the vector every output row is a fixed-order sum  sum_t coef_t * V[idx_t]  over (mpx_gather in mpx.h, sources remapped
//     to positions in V by the host).  Raw values never leave the compute unit;
//   * workgroups are persistent: a lane owns the same rows (row = k * NT + lane inside each output array) for every
//     evaluation point it sees, and keeps their first term (position, coefficient) in REGISTERS, loaded once per kernel
//     instead of once per evaluation -- the per-problem row counts are compile-time constants of the generated source
//     (MPX_FUSE_*), so the row loops unroll and the table is straight-line register code;
//   * the row phase is then nothing but  ds_read -> v_fma -> global_store  with a uniform base and a running lane offset:
//     each store instruction of a wavefront writes one contiguous 512-byte run;
//   * rows with 2 .. MT terms (a few hundred: defect rows = D.X - f, mid-point residual entries; MT = the ELL width the code
//     generator picks per pass) are summed by one lane each from an ELL table (L1 / L2 resident), rows with more terms
//     (objective, d/dt0, d/dtf, d/dwidths, the sum-to-one row of the widths) by one wavefront each with the same shuffle tree
//     as mpx_gather_kernel, which uses the same threshold (MpxGatherArgs::long_threshold).
// Every sum keeps the term order and the fma chain of the two-pass kernels: results are bit-identical to them (tested), so the
// host may pick either path by batch size.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "mpx_device.h"

#ifndef MPX_FUSE_LOC_G  // table entries of a local variable fetched together: 6 where the packed row registers leave room, else 4
#if defined(MPX_FUSE_NDICT_FGJ) && MPX_FUSE_NDICT_FGJ > 0 && MPX_FUSE_NDICT_FGJ <= 2048
#define MPX_FUSE_LOC_G 6
#else
#define MPX_FUSE_LOC_G 4
#endif
#endif
// Packed tables of the local variables / multipliers (MpxPtSet::loc_pack, mu_pack) when the generator found at most 1024 distinct
// coefficients (MPX_FUSE_LOC_DICT_MAX) in each family (MPX_FUSE_NDICT_LOC / _MU, exact counts): one 32-bit load per term instead of an index and a double,
// one register per entry in flight instead of three
#ifndef MPX_FUSE_NDICT_LOC
#define MPX_FUSE_NDICT_LOC 0
#endif
#ifndef MPX_FUSE_NDICT_MU
#define MPX_FUSE_NDICT_MU 0
#endif
#ifndef MPX_FUSE_LOC_DICT_MAX
#define MPX_FUSE_LOC_DICT_MAX 1024
#endif
#define MPX_FUSE_NLD ((MPX_FUSE_NDICT_LOC) > 0 && (MPX_FUSE_NDICT_LOC) <= MPX_FUSE_LOC_DICT_MAX ? (MPX_FUSE_NDICT_LOC) : 0)
#define MPX_FUSE_NMD ((MPX_FUSE_NDICT_MU) > 0 && (MPX_FUSE_NDICT_MU) <= MPX_FUSE_LOC_DICT_MAX ? (MPX_FUSE_NDICT_MU) : 0)
#ifndef MPX_FUSE_LOC_GP
#define MPX_FUSE_LOC_GP 5  // packed entries of a local variable fetched together (3 ... 20 measured: 5 completes the 6-term interpolation variables in one round)
#endif
#ifndef MPX_FUSE_TASK_PER_U
#define MPX_FUSE_TASK_PER_U 1
#endif
#ifndef MPX_FUSE_CHAINS
#define MPX_FUSE_CHAINS 0
#endif
#ifndef MPX_FUSE_CHAIN_MAX
#define MPX_FUSE_CHAIN_MAX 256  // LDS slots for the partial sums of chained local variables (all chains of the context together)
#endif
#ifndef MPX_FUSE_MULTI_EARLY
#define MPX_FUSE_MULTI_EARLY 0
#endif
#ifndef MPX_FUSE_PACKED_EARLY
#define MPX_FUSE_PACKED_EARLY 1
#endif
#ifndef MPX_FUSE_MROW_HES
#define MPX_FUSE_MROW_HES 0
#endif
#ifndef MPX_FUSE_MID_WAVE
#define MPX_FUSE_MID_WAVE 0
#endif
#ifndef MPX_FUSE_OUT_INDEXED
#define MPX_FUSE_OUT_INDEXED 0
#endif
#ifndef MPX_FUSE_NT
#define MPX_FUSE_NT 512  // lanes per workgroup
#endif

namespace mpxk {

#ifndef MPX_FUSE_PAIR_ROWS
#define MPX_FUSE_PAIR_ROWS 0  // (measured 6 % slower at moon lander 20x5, tools/r4_adaptive_ab.py: 60.2 against 56.8 us)
#endif
typedef double mpx_d2u __attribute__((ext_vector_type(2), aligned(8)));
// Row <-> (register k, lane l).  Paired (MPX_FUSE_PAIR_ROWS=1): registers 2j, 2j + 1 of lane l hold the ADJACENT rows 2 (j NT + l), + 1, so that the
// two values leave as one 16-byte store (a wavefront's store instruction then writes a 1 KB run); an odd last register holds row
// (K - 1) NT + l.  Unpaired (default): row k NT + l.
template <int K, int NT>
__device__ __forceinline__ int fused_row_of(int k, int l) {
  if (MPX_FUSE_PAIR_ROWS && k < (K & ~1)) return (k >> 1) * 2 * NT + 2 * l + (k & 1);
  return k * NT + l;
}
template <int NA, int NT>
struct RowRegs {  // first terms of the rows k * NT + lane, k < K, of one output array; idx < 0: not a single-term row of this lane
  static constexpr int K = (NA + NT - 1) / NT;
  int idx[K > 0 ? K : 1];
  double coef[K > 0 ? K : 1];
  __device__ __forceinline__ void load(const ::MpxFusedArgs& A, int base, int l) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int r = fused_row_of<K, NT>(k, l);
      const bool ok = r < NA && A.r_nt[base + (r < NA ? r : 0)] <= 1;
      idx[k] = ok ? A.r_idx[base + r] : -1;
      coef[k] = ok ? A.r_coef[base + r] : 0.0;
    }
  }
  // out[r] = fma(coef, V[idx], 0) for the lane's single-term rows (the fma with a zero addend is what the two-pass kernel
  // computes: it differs from a plain product in the sign of a zero result)
  __device__ __forceinline__ void store(const double* __restrict__ V, double* __restrict__ out, int l) const {
    if (!out) return;
#pragma unroll
    for (int k = 0; k < K; ++k)
      if (idx[k] >= 0) out[fused_row_of<K, NT>(k, l)] = fma(coef[k], V[idx[k]], 0.0);
  }
};

// The same with ONE register per row: position in V and a 16-bit code of the coefficient (MpxFusedArgs::r_pack / r_dict, the
// dictionary in LDS).  Frees two thirds of the row registers -- 36 of 128 VGPRs for moon lander 20x5 -- for the table entries the
// point tasks want in flight.
template <int NA, int NT>
struct RowRegsPacked {
  static constexpr int K = (NA + NT - 1) / NT;
  uint32_t pk[K > 0 ? K : 1];
  __device__ __forceinline__ void load(const ::MpxFusedArgs& A, int base, int l) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int r = fused_row_of<K, NT>(k, l);
      pk[k] = r < NA ? A.r_pack[base + r] : 0xffffffffu;
    }
  }
  __device__ __forceinline__ void store(const double* __restrict__ V, const double* __restrict__ dict, double* __restrict__ out, int l) const {
    if (!out) return;
    constexpr int KP = MPX_FUSE_PAIR_ROWS ? (K & ~1) : 0;
#pragma unroll
    for (int k = 0; k < KP; k += 2) {
      const bool ha = pk[k] != 0xffffffffu, hb = pk[k + 1] != 0xffffffffu;
      const double a = ha ? fma(dict[pk[k] >> 16], V[pk[k] & 0xffffu], 0.0) : 0.0, b = hb ? fma(dict[pk[k + 1] >> 16], V[pk[k + 1] & 0xffffu], 0.0) : 0.0;
      double* __restrict__ o = out + fused_row_of<K, NT>(k, l);
      if (ha && hb) *(mpx_d2u*)o = mpx_d2u{a, b};
      else if (ha) o[0] = a;
      else if (hb) o[1] = b;
    }
#pragma unroll
    for (int k = KP; k < K; ++k)
      if (pk[k] != 0xffffffffu) out[fused_row_of<K, NT>(k, l)] = fma(dict[pk[k] >> 16], V[pk[k] & 0xffffu], 0.0);
  }
};

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {  // f(integral_constant<int, I>) ... f(integral_constant<int, N - 1>)
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

#ifndef MPX_FUSE_SET_MAX_ENTRIES
#define MPX_FUSE_SET_MAX_ENTRIES 96  // table entries of a point (32-bit codes) a lane may hold at once (mpxgen::SetT path)
#endif

// point functions of one 64-point block for the U evaluation points of a chunk: a term's table entry (position, coefficient) is
// read once and applied to all of them (per evaluation point the terms are added in the stored order, as in the two-pass point
// kernels); local variables from z in LDS, results to raw in LDS
// SET >= 0 (round 4): the term counts of the set's local variables and multipliers are compile-time constants of the generated
// source (mpxgen::SetT<SET>::lt / mt -- the running offsets the host computes from the same tables), so EVERY table entry of the
// point is requested in one round trip and the sums are straight-line code; with run-time counts the loop of each variable was its
// own dependent round trip (ten in a row for a moon-lander mid-point: 4.5 of the 5 us of a point task).
template <int FID, int MODE, int U, int VN, int RAWN, int SET = -1>
__device__ __forceinline__ void fused_point(const MpxPtSet& S, const int* __restrict__ ltoff, const int* __restrict__ mtoff, const ::MpxFusedArgs& A, int blk, int lane,
                                            double (*V)[VN], int b0, int nu, const double* __restrict__ CH, const double* __restrict__ ldict,
                                            const double* __restrict__ mdict) {
  using F = mpxgen::Pt<FID>;
  constexpr int NLD = MPX_FUSE_NLD, NMD = MPX_FUSE_NMD;
  constexpr int NLOC = F::NLOC, NCST = F::NCST, NOUT = F::NOUT, NJ = F::NJ, NH = F::NH;
  if constexpr (MODE == MPX_MODE_HESS && NH == 0) return;
  const int p = blk * 64 + lane;
#ifdef MPX_FUSE_PT_STAMPS
  long long* dq = (A.dbg && blockIdx.x == 1 && threadIdx.x == 0 && CH[MPX_FUSE_CHAIN_MAX - 1] == 2.0) ? A.dbg + 16 : nullptr;
  if (dq) dq[0] = wall_clock64();
#endif
  if (p >= S.n) return;
  const int64_t n = S.n;
  double cst[NCST > 0 ? NCST : 1];
#pragma unroll
  for (int k = 0; k < NCST; ++k) cst[k] = S.cst[(int64_t)k * n + p];
  double loc[U][NLOC > 0 ? NLOC : 1];
  // Local variables: fixed-order sums over z (in LDS), terms added in stored order as in the two-pass point kernels.  Measured
  // (MPX_FUSE_PT_STAMPS, moon lander 20x5): this gather is 7.6 us of a 9 us point task -- one memory round trip per variable more
  // than per term: fetching 4 or 8 table entries of a variable together gains 15-25 % of it, a term-major loop over all variables
  // (one round trip per term index) or taking the 19-term running width sum out of it (MPX_FUSE_CHAINS) lose more to registers
  // and to the extra phase than they save.
  // Round 3, late: the FIRST table entry of every variable is requested before any is used (one memory round trip for all the
  // single-term variables -- states, controls, t0, tf, the segment's width -- instead of one each: the unrolled loop over v with a
  // run-time term loop inside would not let the compiler overlap them), the remaining terms of the long ones follow in groups of
  // MPX_FUSE_LOC_G.  Same terms, same order, same fma chains.
  constexpr bool SETC = []() {
    if constexpr (SET >= 0 && NLD > 0) return mpxgen::SetT<(SET >= 0 ? SET : 0)>::lt(NLOC) <= MPX_FUSE_SET_MAX_ENTRIES;
    else return false;
  }();
  if constexpr (SETC) {
    using ST = mpxgen::SetT<(SET >= 0 ? SET : 0)>;
    constexpr int NE = ST::lt(NLOC);
    uint32_t e[NE > 0 ? NE : 1];
#pragma unroll
    for (int t = 0; t < NE; ++t) e[t] = S.loc_pack[(int64_t)t * n + p];
    static_for<0, NLOC>([&](auto vc) {
      constexpr int v = decltype(vc)::value, ta = ST::lt(v), tb = ST::lt(v + 1);
      double acc[U];
      if (MPX_FUSE_CHAINS && v == S.chain_v) {
        const int slot = S.chain_pos[p];
#pragma unroll
        for (int u = 0; u < U; ++u) loc[u][v] = CH[u * MPX_FUSE_CHAIN_MAX + slot];
        return;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) acc[u] = 0.0;
      static_for<ta, tb>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        const double cq = ldict[e[t] >> 16];
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = fma(cq, V[u][RAWN + (int)(e[t] & 0xffffu)], acc[u]);  // (first term: fma(c, v, 0.0) as in the two-pass kernels)
      });
#pragma unroll
      for (int u = 0; u < U; ++u) loc[u][v] = acc[u];
    });
  } else if constexpr (NLD > 0) {
    // packed tables: the first entry of every variable in one round trip (one register each), the rest in groups of MPX_FUSE_LOC_GP
    uint32_t e0[NLOC > 0 ? NLOC : 1];
#pragma unroll
    for (int v = 0; v < NLOC; ++v) {
      const int t0 = ltoff[v] < ltoff[v + 1] ? ltoff[v] : (ltoff[v] > 0 ? ltoff[v] - 1 : 0);
      e0[v] = S.loc_pack[(int64_t)t0 * n + p];
    }
#pragma unroll
    for (int v = 0; v < NLOC; ++v) {
      double acc[U];
      if (MPX_FUSE_CHAINS && v == S.chain_v) {
        const int slot = S.chain_pos[p];
#pragma unroll
        for (int u = 0; u < U; ++u) loc[u][v] = CH[u * MPX_FUSE_CHAIN_MAX + slot];
        continue;
      }
      const int t1 = ltoff[v + 1];
      {
        const double c0 = ldict[e0[v] >> 16];
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = ltoff[v] < t1 ? fma(c0, V[u][RAWN + (int)(e0[v] & 0xffffu)], 0.0) : 0.0;
      }
      for (int t = ltoff[v] + 1; t < t1; t += MPX_FUSE_LOC_GP) {
        uint32_t pe[MPX_FUSE_LOC_GP];
#pragma unroll
        for (int q = 0; q < MPX_FUSE_LOC_GP; ++q) pe[q] = S.loc_pack[(int64_t)(t + q < t1 ? t + q : t1 - 1) * n + p];
#pragma unroll
        for (int q = 0; q < MPX_FUSE_LOC_GP; ++q)
          if (t + q < t1) {
            const double cq = ldict[pe[q] >> 16];
#pragma unroll
            for (int u = 0; u < U; ++u) acc[u] = fma(cq, V[u][RAWN + (int)(pe[q] & 0xffffu)], acc[u]);
          }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) loc[u][v] = acc[u];
    }
  } else {
  int ix0[NLOC > 0 ? NLOC : 1];
  double cf0[NLOC > 0 ? NLOC : 1];
#pragma unroll
  for (int v = 0; v < NLOC; ++v) {
    const int t0 = ltoff[v] < ltoff[v + 1] ? ltoff[v] : (ltoff[v] > 0 ? ltoff[v] - 1 : 0);  // (a variable without terms reads a valid entry it does not use)
    ix0[v] = S.loc_idx[(int64_t)t0 * n + p], cf0[v] = S.loc_coef[(int64_t)t0 * n + p];
  }
#pragma unroll
  for (int v = 0; v < NLOC; ++v) {
    double acc[U];
    if (MPX_FUSE_CHAINS && v == S.chain_v) {  // chained variable: the shared partial sums are in LDS (fused_body), this point reads its prefix
      const int slot = S.chain_pos[p];
#pragma unroll
      for (int u = 0; u < U; ++u) loc[u][v] = CH[u * MPX_FUSE_CHAIN_MAX + slot];
      continue;
    }
    const int t1 = ltoff[v + 1];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] = ltoff[v] < t1 ? fma(cf0[v], V[u][RAWN + ix0[v]], 0.0) : 0.0;
    for (int t = ltoff[v] + 1; t < t1; t += MPX_FUSE_LOC_G) {
      int ix[MPX_FUSE_LOC_G];
      double cf[MPX_FUSE_LOC_G];
#pragma unroll
      for (int q = 0; q < MPX_FUSE_LOC_G; ++q) {
        const int tt = t + q < t1 ? t + q : t1 - 1;  // (clamped: a valid address; the value is not used)
        ix[q] = S.loc_idx[(int64_t)tt * n + p], cf[q] = S.loc_coef[(int64_t)tt * n + p];
      }
#pragma unroll
      for (int q = 0; q < MPX_FUSE_LOC_G; ++q)
        if (t + q < t1) {
#pragma unroll
          for (int u = 0; u < U; ++u) acc[u] = fma(cf[q], V[u][RAWN + ix[q]], acc[u]);  // (slots past the batch hold stale values: never used)
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) loc[u][v] = acc[u];
  }
  }
#ifdef MPX_FUSE_PT_STAMPS
  if (dq) dq[1] = wall_clock64();
#endif
  if constexpr (MODE == MPX_MODE_HESS) {
    double mu[U][NOUT > 0 ? NOUT : 1];
    constexpr bool SETM = []() {
      if constexpr (SET >= 0) return mpxgen::SetT<(SET >= 0 ? SET : 0)>::mt(NOUT) <= 16;
      else return false;
    }();
    if constexpr (SETM) {  // every multiplier entry in one round trip, every multiplier in the next, then straight-line sums
      using ST = mpxgen::SetT<(SET >= 0 ? SET : 0)>;
      constexpr int NM = ST::mt(NOUT);
      int ix[NM > 0 ? NM : 1];
      double cf[NM > 0 ? NM : 1], lm[U][NM > 0 ? NM : 1];
#pragma unroll
      for (int t = 0; t < NM; ++t) {
        const int64_t tt = (int64_t)t * n + p;
        if constexpr (NMD > 0) {
          const uint32_t e = S.mu_pack[tt];
          ix[t] = (int)(e & 0xffffu), cf[t] = mdict[e >> 16];
        } else {
          ix[t] = S.mu_idx[tt], cf[t] = S.mu_coef[tt];
        }
      }
#pragma unroll
      for (int t = 0; t < NM; ++t)
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int b = b0 + (u < nu ? u : 0);
          lm[u][t] = ix[t] == A.n_g ? A.sigma[b] : A.lam[(int64_t)b * A.lam_stride + ix[t]];
        }
      static_for<0, NOUT>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        double acc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = 0;
        static_for<ST::mt(r), ST::mt(r + 1)>([&](auto tc) {
          constexpr int t = decltype(tc)::value;
#pragma unroll
          for (int u = 0; u < U; ++u) acc[u] = fma(cf[t], lm[u][t], acc[u]);
        });
#pragma unroll
        for (int u = 0; u < U; ++u) mu[u][r] = acc[u];
      });
    } else
#pragma unroll
    for (int r = 0; r < NOUT; ++r) {
      double acc[U];
#pragma unroll
      for (int u = 0; u < U; ++u) acc[u] = 0;
      // (four table entries and their multipliers in flight: one term at a time was a dependent pair of round trips per term)
      for (int t = mtoff[r]; t < mtoff[r + 1]; t += 4) {
        int ix[4];
        double cf[4], lm[U][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int64_t tt = (int64_t)(t + q < mtoff[r + 1] ? t + q : mtoff[r + 1] - 1) * n + p;
          if constexpr (NMD > 0) {
            const uint32_t e = S.mu_pack[tt];
            ix[q] = (int)(e & 0xffffu), cf[q] = mdict[e >> 16];
          } else {
            ix[q] = S.mu_idx[tt], cf[q] = S.mu_coef[tt];
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int b = b0 + (u < nu ? u : 0);
            lm[u][q] = ix[q] == A.n_g ? A.sigma[b] : A.lam[(int64_t)b * A.lam_stride + ix[q]];
          }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (t + q < mtoff[r + 1]) {
#pragma unroll
            for (int u = 0; u < U; ++u) acc[u] = fma(cf[q], lm[u][q], acc[u]);
          }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) mu[u][r] = acc[u];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (u >= nu) break;
      double H[NH > 0 ? NH : 1];
      F::hes(loc[u], cst, mu[u], H);
      double* __restrict__ rb = &V[u][0] + S.rawh_off;
#pragma unroll
      for (int q = 0; q < NH; ++q) rb[(int64_t)q * n + p] = H[q];
    }
  } else {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (u >= nu) break;
      double* __restrict__ rb = &V[u][0] + S.raw_off;
      double out[NOUT > 0 ? NOUT : 1];
      if constexpr (MODE == MPX_MODE_FGJ) {
        double J[NJ > 0 ? NJ : 1];
        F::jac(loc[u], cst, out, J);
#ifdef MPX_FUSE_PT_STAMPS
        if (dq) dq[2] = wall_clock64() + (long long)(J[0] == 1.2345e300);
#endif
#pragma unroll
        for (int q = 0; q < NJ; ++q) rb[(int64_t)(NOUT + q) * n + p] = J[q];
      } else {
        F::val(loc[u], cst, out);
      }
#pragma unroll
      for (int r = 0; r < NOUT; ++r) rb[(int64_t)r * n + p] = out[r];
    }
  }
#ifdef MPX_FUSE_PT_STAMPS
  if (dq) dq[3] = wall_clock64();
#endif
}

template <int MODE, int FID, int U, int VN, int RAWN>
struct FusedDispatch {
  __device__ static __forceinline__ void run(const MpxPtSet& S, const int* lt, const int* mt, const ::MpxFusedArgs& A, int blk, int lane, double (*V)[VN], int b0, int nu,
                                             const double* CH, const double* ld, const double* md) {
    if (S.fid == FID)
      fused_point<FID, MODE, U, VN, RAWN>(S, lt, mt, A, blk, lane, V, b0, nu, CH, ld, md);
    else
      FusedDispatch<MODE, FID - 1, U, VN, RAWN>::run(S, lt, mt, A, blk, lane, V, b0, nu, CH, ld, md);
  }
};
template <int MODE, int U, int VN, int RAWN>
struct FusedDispatch<MODE, -1, U, VN, RAWN> {
  __device__ static __forceinline__ void run(const MpxPtSet&, const int*, const int*, const ::MpxFusedArgs&, int, int, double (*)[VN], int, int, const double*, const double*,
                                             const double*) {}
};

#ifdef MPX_FUSE_SETS  // dispatch on the set index: term counts of the set are compile-time constants (mpxgen::SetT<SET>)
template <int MODE, int SET, int U, int VN, int RAWN>
struct FusedDispatchSet {
  __device__ static __forceinline__ void run(int k, const MpxPtSet& S, const int* lt, const int* mt, const ::MpxFusedArgs& A, int blk, int lane, double (*V)[VN], int b0, int nu,
                                             const double* CH, const double* ld, const double* md) {
    if (k == SET)
      fused_point<mpxgen::SetT<SET>::FID, MODE, U, VN, RAWN, SET>(S, lt, mt, A, blk, lane, V, b0, nu, CH, ld, md);
    else
      FusedDispatchSet<MODE, SET - 1, U, VN, RAWN>::run(k, S, lt, mt, A, blk, lane, V, b0, nu, CH, ld, md);
  }
};
template <int MODE, int U, int VN, int RAWN>
struct FusedDispatchSet<MODE, -1, U, VN, RAWN> {
  __device__ static __forceinline__ void run(int, const MpxPtSet&, const int*, const int*, const ::MpxFusedArgs&, int, int, double (*)[VN], int, int, const double*, const double*,
                                             const double*) {}
};
#endif

// MODE_FG / MODE_FGJ: arrays f (1 row), g (NG), grad_f (NZ), jac_val (NNZJ); MODE_HESS: hess_val (NNZH).
// MT: ELL width of the multi-term rows (rows with 2 .. MT terms); RL x TL: long rows per wavefront x 64-term rounds
// per long row when their table fits the register budget (RL == 0: read from global memory per chunk).
template <int MODE, int NF, int NT, int U, int RAWN, int NZ, int N0, int N1, int N2, int N3, int MT, int RM, int RL, int TL, int NDICT>
__device__ __forceinline__ void fused_body(const ::MpxFusedArgs& A) {
  constexpr int VN = RAWN + NZ + 1;  // [raw | z | 1.0]
  __shared__ double V[U][VN];
  __shared__ double CH[2][U][MPX_FUSE_CHAIN_MAX];  // partial sums of the chained local variables (MpxPtSet::chain_v), this chunk's and the next one's
  // (the wavefront index through readfirstlane: the compiler then knows that everything derived from it -- the point set of a
  // task, its table pointers and term counts, the long row of a wavefront -- is uniform: scalar loads and branches, global_load
  // with a scalar base instead of flat_load with per-lane 64-bit addresses; without it the point phase was 2x slower)
  const int l = threadIdx.x, lane = l & 63, wave = __builtin_amdgcn_readfirstlane(l >> 6);
  constexpr int NW = NT / 64;
  // Set descriptors and their term-offset arrays, copied to LDS once per kernel: a point task reads its set, the offsets of its
  // local variables and of its multipliers from here.  From global memory each of these was a dependent round trip per local
  // variable (descriptor -> offsets -> table entries): ~10 us per task, the whole point phase of a chunk.
  constexpr int MAXS = 16, MAXV = 48;
  __shared__ MpxPtSet sS[MAXS];
  __shared__ int sLt[MAXS][MAXV], sMt[MAXS][MAXV];
  {
    const int ns = A.n_sets < MAXS ? A.n_sets : MAXS;
    for (int k = l; k < ns; k += NT) sS[k] = A.sets[k];
    __syncthreads();
    for (int e = l; e < ns * MAXV; e += NT) {
      const int k = e / MAXV, v = e - k * MAXV;
      // (offset arrays have n_loc + 1 / n_out + 1 entries; the generated functions know those counts, the kernel reads only them)
      sLt[k][v] = v <= sS[k].n_loc ? sS[k].loc_toff[v] : 0;
      sMt[k][v] = v <= sS[k].n_out ? sS[k].mu_toff[v] : 0;
    }
    __syncthreads();
  }
  __shared__ double sLDict[MPX_FUSE_NLD > 0 ? MPX_FUSE_NLD : 1], sMDict[MPX_FUSE_NMD > 0 ? MPX_FUSE_NMD : 1];  // ... of the packed local / multiplier tables
  if constexpr (MPX_FUSE_NLD > 0)
    for (int e = l; e < MPX_FUSE_NLD; e += NT) sLDict[e] = e < A.n_ldict ? A.l_dict[e] : 0.0;
  if constexpr (MPX_FUSE_NMD > 0)
    for (int e = l; e < MPX_FUSE_NMD; e += NT) sMDict[e] = e < A.n_mdict ? A.m_dict[e] : 0.0;
  if constexpr (MPX_FUSE_NLD > 0 || MPX_FUSE_NMD > 0) __syncthreads();
  __shared__ double sDict[NDICT > 0 ? NDICT : 1];  // coefficient dictionary of the packed single-term rows
  if constexpr (NDICT > 0) {
    for (int e = l; e < NDICT; e += NT) sDict[e] = e < A.n_dict ? A.r_dict[e] : 0.0;
    __syncthreads();
  }
  std::conditional_t<(NDICT > 0), RowRegsPacked<N0, NT>, RowRegs<N0, NT>> R0;
  std::conditional_t<(NDICT > 0), RowRegsPacked<N1, NT>, RowRegs<N1, NT>> R1;
  std::conditional_t<(NDICT > 0), RowRegsPacked<N2, NT>, RowRegs<N2, NT>> R2;
  std::conditional_t<(NDICT > 0), RowRegsPacked<N3, NT>, RowRegs<N3, NT>> R3;
  R0.load(A, 0, l);
  R1.load(A, N0, l);
  R2.load(A, N0 + N1, l);
  R3.load(A, N0 + N1 + N2, l);
  // (not in the Hessian kernel by default: four more registers there cost a workgroup per compute unit, MPX_FUSE_MROW_HES)
  constexpr bool MROW = RM > 0 && RM <= 4 && (MODE != MPX_MODE_HESS || MPX_FUSE_MROW_HES);
  int mrow_[MROW ? RM : 1], mnt_[MROW ? RM : 1];
  if constexpr (MROW) {
#pragma unroll
    for (int q = 0; q < RM; ++q) {
      const int m = q * NT + l;
      mrow_[q] = m < A.n_multi ? A.multi_rows[m] : 0;
      mnt_[q] = m < A.n_multi ? A.r_nt[mrow_[q]] : 0;
    }
  }
  // (the four bases and strides from LDS: lgkmcnt, not vmcnt)
  __shared__ double* sOb[4];
  __shared__ int64_t sOs[4];
  if (l < 4) sOb[l] = A.out[l], sOs[l] = A.out_stride[l];  // (visible behind the first barrier of the chunk loop)
  // (constant indices into the argument block only: A.out[a] with the lane's own `a` is not a scalar register but a load from the
  // argument block in memory -- and the wait for it, s_waitcnt vmcnt(0) in front of the row's store, also waited for every store
  // the wavefront had issued before: the multi-term and long rows drained the burst of single-term rows one row at a time)
  auto out_of = [&](int row, int64_t b) -> double* {
#if MPX_FUSE_OUT_INDEXED  // (the form of rounds 3 / 4, A/B)
    int a = 0, loc = row;
    if (loc >= N0) { loc -= N0, a = 1; if (loc >= N1) { loc -= N1, a = 2; if (loc >= N2) { loc -= N2, a = 3; } } }
    return A.out[a] ? A.out[a] + b * A.out_stride[a] + loc : nullptr;
#else
    int a = 0, loc = row;
    if (loc >= N0) { loc -= N0, a = 1; if (loc >= N1) { loc -= N1, a = 2; if (loc >= N2) { loc -= N2, a = 3; } } }
    double* base = sOb[a];
    return base ? base + b * sOs[a] + loc : nullptr;
#endif
  };
  // long rows of this wavefront (rows wave, wave + NW, ...): lane j holds terms j, j + 64, ... of each
  int lidx[RL > 0 ? RL : 1][TL > 0 ? TL : 1];
  double lcf[RL > 0 ? RL : 1][TL > 0 ? TL : 1];
  if constexpr (RL > 0) {
#pragma unroll
    for (int r = 0; r < RL; ++r) {
      const int w = wave + r * NW;
      const int64_t e0 = w < A.n_long ? A.ptr[A.long_rows[w]] : 0, e1 = w < A.n_long ? A.ptr[A.long_rows[w] + 1] : 0;
#pragma unroll
      for (int t = 0; t < TL; ++t) {
        const int64_t e = e0 + lane + 64 * t;
        lidx[r][t] = e < e1 ? A.idx[e] : -1;
        lcf[r][t] = e < e1 ? A.coef[e] : 0.0;
      }
    }
  }
  constexpr int ZR = (U * (NZ + 1) + NT - 1) / NT;  // z staging: elements per lane and chunk
  double zr[ZR];
  auto z_load = [&](int c) {  // z of chunk c (and the 1.0 closing every V[u]) into registers
    const int b0 = c * U, nu = (A.B - b0 < U) ? A.B - b0 : U;
#pragma unroll
    for (int q = 0; q < ZR; ++q) {
      const int e = q * NT + l, u = e / (NZ + 1), i = e - u * (NZ + 1);
      zr[q] = (u < nu && i < NZ) ? A.z[(int64_t)(b0 + u) * A.z_stride + i] : 1.0;
    }
  };
  // Chained local variables: lane (c, u) of the LAST wavefront runs chain c for evaluation point u of a chunk -- a sequential fma
  // chain over z read straight from global memory (all loads first), i.e. the partial sums every point would compute itself --
  // one chunk AHEAD of its use (double-buffered), so that it costs the point phase nothing.
  auto chains_of = [&](int c, int slot) {
    if (MPX_FUSE_CHAINS && l >= NT - 64 && lane < A.n_chains * U) {
      const int cch = lane / U, u = lane - cch * U, b0 = c * U;
      if (b0 + u < A.B) {
        const int e0 = A.ch_ptr[cch], e1 = A.ch_ptr[cch + 1];
        const double* __restrict__ zb = A.z + (int64_t)(b0 + u) * A.z_stride;
        double* __restrict__ o = &CH[slot][u][A.ch_slot[cch]];
        double acc = 0;
        o[0] = acc;
        for (int e = e0; e < e1; e += 8) {
          double zv[8], cv[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int ee = e + q < e1 ? e + q : e1 - 1;
            zv[q] = zb[A.ch_idx[ee]], cv[q] = A.ch_coef[ee];
          }
#pragma unroll
          for (int q = 0; q < 8; ++q)
            if (e + q < e1) acc = fma(cv[q], zv[q], acc), o[e + q - e0 + 1] = acc;
        }
      }
    }
  };
  constexpr int MAXT = 4;
  const int tq0 = A.task_ptr[wave], n_my = A.task_ptr[wave + 1] - tq0;
  int my_t[MAXT];
#pragma unroll
  for (int q = 0; q < MAXT; ++q) my_t[q] = q < n_my ? A.task_list[tq0 + q] : 0;
  const int n_chunks = (A.B + U - 1) / U;
#ifndef MPX_FUSE_XCD_BLOCKED
#define MPX_FUSE_XCD_BLOCKED 1
#endif
  // XCD-blocked walk (as node_body): workgroup i runs on XCD i % 8; the workgroups of one XCD take CONSECUTIVE chunks of every round,
  // so each L2 streams a contiguous eighth of the round's outputs instead of every eighth chunk
  const int first_chunk = (MPX_FUSE_XCD_BLOCKED && gridDim.x % 8 == 0) ? (int)((blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8) : (int)blockIdx.x;
  if (first_chunk < n_chunks) z_load(first_chunk), chains_of(first_chunk, 0);
  int it_ = 0;
#define MPX_FUSE_STAMP(k) do { if (A.dbg && blockIdx.x == 1 && l == 0 && it_ == 2) A.dbg[k] = wall_clock64(); } while (0)
  for (int c = first_chunk; c < n_chunks; c += gridDim.x, ++it_) {
    const int b0 = c * U, nu = (A.B - b0 < U) ? A.B - b0 : U;
    MPX_FUSE_STAMP(0);
#ifdef MPX_FUSE_PT_STAMPS
    if (l == 0) CH[it_ & 1][0][MPX_FUSE_CHAIN_MAX - 1] = (double)it_;
#endif
#pragma unroll
    for (int q = 0; q < ZR; ++q) {
      const int e = q * NT + l, u = e / (NZ + 1), i = e - u * (NZ + 1);
      if (u < U) V[u][RAWN + i] = zr[q];
    }
    __syncthreads();
    MPX_FUSE_STAMP(1);
#ifndef MPX_FUSE_Z_LATE
#define MPX_FUSE_Z_LATE 0
#endif
    // (the next chunk's z, in flight during this chunk's work.  Requesting it BEHIND the point tasks instead -- a wavefront's loads
    // return in order, the table entries of the point tasks are L2 hits -- measured 4 % slower in process, tools/r4_adaptive_ab.py)
#ifndef MPX_ABL_FUSE_NO_ZNEXT  // ablation (wrong results): every chunk of a workgroup works on the z of its first chunk -- what the z fetch costs
    if (!MPX_FUSE_Z_LATE && c + (int)gridDim.x < n_chunks) z_load(c + gridDim.x), chains_of(c + gridDim.x, (it_ + 1) & 1);
#endif
    // ---- point functions: wavefront <-> (64-point block, evaluation point) ----
    // wavefront <-> its scheduled (64-point block, evaluation point) tasks
    for (int q = 0; q < n_my; ++q) {
      // (the wavefront's schedule and the point set of every task: read once per kernel -- per chunk they were two dependent
      // scalar round trips and a search in front of every task, 2 us of a 5.4 us point phase)
      const int t = q < MAXT ? (q == 0 ? my_t[0] : q == 1 ? my_t[1] : q == 2 ? my_t[2] : my_t[3]) : A.task_list[tq0 + q];
      const int u = t / A.n_blocks, bx = t - u * A.n_blocks;
      if (u >= nu) continue;
      int k = 0;
      while (k + 1 < A.n_sets && bx >= sS[k + 1].block_first) ++k;
#if defined(MPX_FUSE_SETS) && !defined(MPX_FUSE_NO_SET_CONSTS)
      FusedDispatchSet<MODE, MPX_FUSE_SETS - 1, 1, VN, RAWN>::run(k, sS[k], sLt[k], sMt[k], A, bx - sS[k].block_first, lane, &V[u], b0 + u, 1, &CH[it_ & 1][u][0], sLDict, sMDict);
#else
      FusedDispatch<MODE, NF - 1, 1, VN, RAWN>::run(sS[k], sLt[k], sMt[k], A, bx - sS[k].block_first, lane, &V[u], b0 + u, 1, &CH[it_ & 1][u][0], sLDict, sMDict);
#endif
    }
    if (MPX_FUSE_Z_LATE && c + (int)gridDim.x < n_chunks) z_load(c + gridDim.x), chains_of(c + gridDim.x, (it_ + 1) & 1);
    // table entries of this lane's multi-term rows: loads issued before the barrier, used after the single-term rows
    MPX_FUSE_STAMP(2);
    __syncthreads();
    MPX_FUSE_STAMP(3);
#if MPX_FUSE_MULTI_EARLY
    // table entries of this lane's multi-term rows, requested BEFORE the burst of single-row stores: memory operations of a
    // wavefront retire in order, loads issued behind the stores would wait for the store queue to drain
    int eix[RM > 0 ? RM : 1][MT > 0 ? MT : 1], ent[RM > 0 ? RM : 1], erow[RM > 0 ? RM : 1];
    double ecf[RM > 0 ? RM : 1][MT > 0 ? MT : 1];
#pragma unroll
    for (int r = 0; r < RM; ++r) {
      const int m = r * NT + l, mm = m < A.n_multi ? m : 0;
      erow[r] = A.multi_rows[mm];
      ent[r] = m < A.n_multi ? A.r_nt[erow[r]] : 0;
#pragma unroll
      for (int t = 0; t < MT; ++t) eix[r][t] = A.m_idx[(int64_t)t * A.n_multi + mm], ecf[r][t] = A.m_coef[(int64_t)t * A.n_multi + mm];
    }
#endif
    // Packed ELL table (NDICT > 0): one 32-bit entry per term (position | coefficient code) -- a third of the registers and half
    // the load instructions of the unpacked table, so the entries of ALL the lane's multi-term rows can be requested BEFORE the
    // burst of single-term stores (memory operations of a wavefront retire in order: loads issued behind the stores wait for the
    // store queue to drain -- the multi-term phase was 5 us of a 15 us chunk) and used after it.
    constexpr bool PE = NDICT > 0 && MROW && MPX_FUSE_PACKED_EARLY && !MPX_FUSE_MULTI_EARLY;
    uint32_t epk[PE ? RM : 1][PE ? (MT > 0 ? MT : 1) : 1];
    if constexpr (PE) {
      int le = l;
      asm volatile("" : "+v"(le));  // (opaque per chunk: the table addresses are recomputed, not hoisted into registers)
#pragma unroll
      for (int r = 0; r < RM; ++r) {
        const int m = r * NT + le, mm = m < A.n_multi ? m : 0;
#pragma unroll
        for (int t = 0; t < MT; ++t) epk[r][t] = A.m_pack[(int64_t)t * A.n_multi + mm];
      }
    }
    // ---- rows with one term: registers -> LDS read -> store ----
    for (int u = 0; u < nu; ++u) {
      const double* __restrict__ Vu = V[u];
      const int64_t b = b0 + u;
      if constexpr (NDICT > 0) {
        R0.store(Vu, sDict, A.out[0] ? A.out[0] + b * A.out_stride[0] : nullptr, l);
        R1.store(Vu, sDict, A.out[1] ? A.out[1] + b * A.out_stride[1] : nullptr, l);
        R2.store(Vu, sDict, A.out[2] ? A.out[2] + b * A.out_stride[2] : nullptr, l);
        R3.store(Vu, sDict, A.out[3] ? A.out[3] + b * A.out_stride[3] : nullptr, l);
      } else {
        R0.store(Vu, A.out[0] ? A.out[0] + b * A.out_stride[0] : nullptr, l);
        R1.store(Vu, A.out[1] ? A.out[1] + b * A.out_stride[1] : nullptr, l);
        R2.store(Vu, A.out[2] ? A.out[2] + b * A.out_stride[2] : nullptr, l);
        R3.store(Vu, A.out[3] ? A.out[3] + b * A.out_stride[3] : nullptr, l);
      }
    }
    MPX_FUSE_STAMP(4);
    // ---- rows with 2 .. MT terms: one lane per row, ELL table [t][row] (the MT loads of a row are independent), added in stored order ----
#if MPX_FUSE_MULTI_EARLY
#pragma unroll
    for (int r = 0; r < RM; ++r) {
      if (ent[r] > 0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (u >= nu) break;
          double sm = 0;
#pragma unroll
          for (int t = 0; t < MT; ++t)
            if (t < ent[r]) sm = fma(ecf[r][t], V[u][eix[r][t]], sm);
          double* o = out_of(erow[r], b0 + u);
          if (o) *o = sm;
        }
      }
    }
    for (int m = A.n_multi; m < A.n_multi; m += NT) {
#else
    if constexpr (PE) {
#pragma unroll
      for (int r = 0; r < RM; ++r)
        if (mnt_[r] > 0) {
#pragma unroll
          for (int u = 0; u < U; ++u) {
            if (u >= nu) break;
            double sm = 0;
#pragma unroll
            for (int t = 0; t < MT; ++t)
              if (t < mnt_[r]) sm = fma(sDict[epk[r][t] >> 16], V[u][epk[r][t] & 0xffffu], sm);
            double* o = out_of(mrow_[r], b0 + u);
            if (o) *o = sm;
          }
        }
    }
    int lm = l;
    asm volatile("" : "+v"(lm));  // (opaque per chunk: else the 2 MT table addresses of every round are hoisted out of the chunk loop -- registers)
    for (int m = PE ? A.n_multi : lm, r_ = 0; m < A.n_multi; m += NT, ++r_) {
#endif
      // (row and term count of the lane's r_-th multi-term row: registers for the life of the workgroup where the row count is a
      // compile-time constant -- two dependent loads per round less)
      int row, nt;
      if constexpr (MROW) {
        row = mrow_[0], nt = mnt_[0];
#pragma unroll
        for (int q = 1; q < RM; ++q)
          if (r_ == q) row = mrow_[q], nt = mnt_[q];
      } else {
        row = A.multi_rows[m], nt = A.r_nt[row];
      }
      int ix[MT > 0 ? MT : 1];
      double cf[MT > 0 ? MT : 1];
      if constexpr (NDICT > 0) {
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          const uint32_t e = A.m_pack[(int64_t)t * A.n_multi + m];
          ix[t] = (int)(e & 0xffffu), cf[t] = sDict[e >> 16];
        }
      } else {
#pragma unroll
        for (int t = 0; t < MT; ++t) ix[t] = A.m_idx[(int64_t)t * A.n_multi + m], cf[t] = A.m_coef[(int64_t)t * A.n_multi + m];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (u >= nu) break;
        double s = 0;
#pragma unroll
        for (int t = 0; t < MT; ++t)
          if (t < nt) s = fma(cf[t], V[u][ix[t]], s);
        double* o = out_of(row, b0 + u);
        if (o) *o = s;
      }
    }
    MPX_FUSE_STAMP(5);
    MPX_FUSE_STAMP(6);
    // ---- long rows: one wavefront per row, lane j sums terms j, j + 64, ... in order, fixed shuffle tree (as mpx_gather_kernel) ----
    if constexpr (RL > 0) {
#pragma unroll
      for (int r = 0; r < RL; ++r) {
        const int w = wave + r * NW;
        if (w < A.n_long) {
          const int row = A.long_rows[w];
          for (int u = 0; u < nu; ++u) {
            double s = 0;
#pragma unroll
            for (int t = 0; t < TL; ++t)
              if (lidx[r][t] >= 0) s = fma(lcf[r][t], V[u][lidx[r][t]], s);
            s = mpx_wave_total(s);
            if (lane == 0) {
              double* op = out_of(row, b0 + u);
              if (op) *op = s;
            }
          }
        }
      }
      // (rows past the register table, should the host ever find more long rows than the generator counted: from global memory)
      for (int w = wave + RL * NW; w < A.n_long; w += NW) {
        const int row = A.long_rows[w];
        const int64_t e0 = A.ptr[row], e1 = A.ptr[row + 1];
        for (int u = 0; u < nu; ++u) {
          double s = 0;
          for (int64_t e = e0 + lane; e < e1; e += 64) s = fma(A.coef[e], V[u][A.idx[e]], s);
          s = mpx_wave_total(s);
          if (lane == 0) {
            double* op = out_of(row, b0 + u);
            if (op) *op = s;
          }
        }
      }
    } else {
      for (int w = wave; w < A.n_long * nu; w += NW) {
        const int u = w / A.n_long, row = A.long_rows[w - u * A.n_long];
        const int64_t e0 = A.ptr[row], e1 = A.ptr[row + 1];
        double s = 0;
        for (int64_t e = e0 + lane; e < e1; e += 64) s = fma(A.coef[e], V[u][A.idx[e]], s);
        s = mpx_wave_total(s);
        if (lane == 0) {
          double* op = out_of(row, b0 + u);
          if (op) *op = s;
        }
      }
    }
    MPX_FUSE_STAMP(7);
    __syncthreads();  // V is rewritten by the next chunk
    MPX_FUSE_STAMP(8);
  }
#undef MPX_FUSE_STAMP
}

}  // namespace mpxk

// LDS budget: U evaluation points of [raw | z | 1] must fit 60 KB (a launch gets 64 KB without raising the function's limit);
// register budget: a lane keeps 3 dwords per row it owns, at most MPX_FUSE_MAX_ROWS_PER_LANE rows (else the kernel is not built
// and the host keeps the two-pass path for this problem).
#ifndef MPX_FUSE_MIN_WAVES  // wavefronts per SIMD the register allocation must allow (these kernels live on occupancy)
#define MPX_FUSE_MIN_WAVES 4
#endif
#ifndef MPX_FUSE_LDS_BYTES
#define MPX_FUSE_LDS_BYTES 61440
#endif
#ifndef MPX_FUSE_MAX_U
#define MPX_FUSE_MAX_U 2
#endif
#ifndef MPX_FUSE_MAX_ROWS_PER_LANE
#define MPX_FUSE_MAX_ROWS_PER_LANE 48
#endif
#define MPX_FUSE_CEIL(n) (((n) + MPX_FUSE_NT - 1) / MPX_FUSE_NT)
#define MPX_FUSE_U_FOR(RAWN, ROWS) ((ROWS) > MPX_FUSE_MAX_ROWS_PER_LANE ? 0 : ((int)(MPX_FUSE_LDS_BYTES / (8 * ((RAWN) + MPX_FUSE_NZ + 1))) > MPX_FUSE_MAX_U ? MPX_FUSE_MAX_U : (int)(MPX_FUSE_LDS_BYTES / (8 * ((RAWN) + MPX_FUSE_NZ + 1)))))
#ifndef MPX_FUSE_U_FGJ
#define MPX_FUSE_U_FGJ MPX_FUSE_U_FOR(MPX_FUSE_RAW_N, 1 + MPX_FUSE_CEIL(MPX_FUSE_NG) + MPX_FUSE_CEIL(MPX_FUSE_NZ) + MPX_FUSE_CEIL(MPX_FUSE_NNZJ))
#endif
#ifndef MPX_FUSE_U_HES
#define MPX_FUSE_U_HES MPX_FUSE_U_FOR(MPX_FUSE_RAWH_N, MPX_FUSE_CEIL(MPX_FUSE_NNZH))
#endif

// Long-row tables in registers when a wavefront's share is at most 16 (position, coefficient) pairs per lane.
#define MPX_FUSE_RL(NLONG) (((NLONG) + MPX_FUSE_NT / 64 - 1) / (MPX_FUSE_NT / 64))
#define MPX_FUSE_TL(LT) (((LT) + 63) / 64)
#ifndef MPX_FUSE_LONG_REGS
#define MPX_FUSE_LONG_REGS 4  // (position, coefficient) pairs per lane a wavefront may keep of its long rows
#endif
#define MPX_FUSE_RL_OK(NLONG, LT) ((NLONG) > 0 && MPX_FUSE_RL(NLONG) * MPX_FUSE_TL(LT) <= MPX_FUSE_LONG_REGS ? MPX_FUSE_RL(NLONG) : 0)

// Packed single-term rows (RowRegsPacked) when the generator found few enough distinct coefficients (MPX_FUSE_NDICT_*, exact counts)
#ifndef MPX_FUSE_NDICT_FGJ
#define MPX_FUSE_NDICT_FGJ 0
#endif
#ifndef MPX_FUSE_NDICT_HES
#define MPX_FUSE_NDICT_HES 0
#endif
#ifndef MPX_FUSE_DICT_MAX
#define MPX_FUSE_DICT_MAX 2048
#endif
#define MPX_FUSE_PACK(N) ((N) > 0 && (N) <= MPX_FUSE_DICT_MAX ? (N) : 0)

// mpx_fuse_info = {lanes per workgroup, U of the first-order kernels, U of the Hessian kernel, ELL width of the multi-term rows
// of the first-order pass, of the Hessian pass, dictionary capacity of the packed single-term rows of the two passes (0: unpacked)}
// (U == 0: that kernel does not exist: one evaluation point does not fit the budgets); the host reads it from the code object.
#define MPX_INSTANTIATE_FUSED(NF)                                                                                              \
  extern "C" __device__ __attribute__((used)) const int mpx_fuse_info[9] = {MPX_FUSE_NT, MPX_FUSE_U_FGJ, MPX_FUSE_U_HES,       \
                                                                              MPX_FUSE_MT_FGJ, MPX_FUSE_MT_HES,                \
                                                                              MPX_FUSE_PACK(MPX_FUSE_NDICT_FGJ), MPX_FUSE_PACK(MPX_FUSE_NDICT_HES), MPX_FUSE_NLD, MPX_FUSE_NMD}; \
  extern "C" __global__ __launch_bounds__(MPX_FUSE_NT, MPX_FUSE_MIN_WAVES) void mpx_asm_fg(const MpxFusedArgs A) {                                \
    if constexpr (MPX_FUSE_U_FGJ > 0)                                                                                          \
      mpxk::fused_body<MPX_MODE_FG, NF, MPX_FUSE_NT, (MPX_FUSE_U_FGJ > 0 ? MPX_FUSE_U_FGJ : 1), MPX_FUSE_RAW_N, MPX_FUSE_NZ, 1, \
                       MPX_FUSE_NG, MPX_FUSE_NZ, MPX_FUSE_NNZJ, MPX_FUSE_MT_FGJ, MPX_FUSE_CEIL(MPX_FUSE_NMULTI_FGJ), MPX_FUSE_RL_OK(MPX_FUSE_NLONG_FGJ, MPX_FUSE_LT_FGJ), \
                       MPX_FUSE_TL(MPX_FUSE_LT_FGJ), MPX_FUSE_PACK(MPX_FUSE_NDICT_FGJ)>(A);                                    \
  }                                                                                                                            \
  extern "C" __global__ __launch_bounds__(MPX_FUSE_NT, MPX_FUSE_MIN_WAVES) void mpx_asm_fgj(const MpxFusedArgs A) {                               \
    if constexpr (MPX_FUSE_U_FGJ > 0)                                                                                          \
      mpxk::fused_body<MPX_MODE_FGJ, NF, MPX_FUSE_NT, (MPX_FUSE_U_FGJ > 0 ? MPX_FUSE_U_FGJ : 1), MPX_FUSE_RAW_N, MPX_FUSE_NZ, 1, \
                       MPX_FUSE_NG, MPX_FUSE_NZ, MPX_FUSE_NNZJ, MPX_FUSE_MT_FGJ, MPX_FUSE_CEIL(MPX_FUSE_NMULTI_FGJ), MPX_FUSE_RL_OK(MPX_FUSE_NLONG_FGJ, MPX_FUSE_LT_FGJ), \
                       MPX_FUSE_TL(MPX_FUSE_LT_FGJ), MPX_FUSE_PACK(MPX_FUSE_NDICT_FGJ)>(A);                                    \
  }                                                                                                                            \
  extern "C" __global__ __launch_bounds__(MPX_FUSE_NT, MPX_FUSE_MIN_WAVES) void mpx_asm_hes(const MpxFusedArgs A) {                               \
    if constexpr (MPX_FUSE_U_HES > 0)                                                                                          \
      mpxk::fused_body<MPX_MODE_HESS, NF, MPX_FUSE_NT, (MPX_FUSE_U_HES > 0 ? MPX_FUSE_U_HES : 1), MPX_FUSE_RAWH_N, MPX_FUSE_NZ, \
                       MPX_FUSE_NNZH, 0, 0, 0, MPX_FUSE_MT_HES, MPX_FUSE_CEIL(MPX_FUSE_NMULTI_HES), MPX_FUSE_RL_OK(MPX_FUSE_NLONG_HES, MPX_FUSE_LT_HES),          \
                       MPX_FUSE_TL(MPX_FUSE_LT_HES), MPX_FUSE_PACK(MPX_FUSE_NDICT_HES)>(A);                                    \
  }
