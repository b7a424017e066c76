// mpx_equal_area.cpp -- libmpx: the equal-area width update of the h-adaptive loop on the device (SURVEY 8(f) rank 2, 8(d) config 5;
// include/mpx.h, mpx_equal_area_widths_device).  Split out of mpx_host.cpp in round 5; shares the context definition and the helpers
// of mpx_internal.h and the scans of mpx_scan.h with the prefix kernel of the widths.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mpx.h"
#include "mpx_device.h"

#include "mpx_internal.h"
#include "mpx_scan.h"

using namespace mpxi;

// ---- h-adaptive width update on the device ---------------------------------------------------
// Equal-area rule of the reference (mpopt_h_adaptive.get_roots_wrt_equal_area, mpopt.py:2636-2659, fed with the per-point
// 2-norms of the dynamics residuals, mpopt.py:2620-2633) and its damped update (mpopt.py:2587-2590), batched: one workgroup
// per evaluation point, fixed-order block scan of the trapezoid areas, one binary search per new segment boundary.
namespace {
// (1024 lanes per workgroup: the cumulative areas of one evaluation point fill most of a compute unit's LDS, so a workgroup is
// alone on its CU and its own 16 wavefronts are all there is to hide load and LDS latency: 4 wavefronts measured 2.3x slower)
#define MPX_EA_THREADS 1024
#define MPX_EA_PF 12  // residual samples a lane can prefetch for the next evaluation point (n <= 12 * 1024)
#define MPX_EA_WR 4   // new segment boundaries per lane in the fast kernel (S <= 4 * 1024)
#ifndef MPX_EA_SLICES
#define MPX_EA_SLICES 4  // 1 ... 4: slices the prefetch of the next point is requested in (fast kernel, scalar residuals)
#endif
#ifndef MPX_EA_COND_PREFETCH
#define MPX_EA_COND_PREFETCH 0  // 1: the guarded prefetch of round 3 (A/B: MPX_LIB_HIPCC_FLAGS=-DMPX_EA_COND_PREFETCH=1)
#endif
// Generic kernel: vector residuals, sample lists of any length (cumulative areas in LDS when they fit, else in HBM scratch), any
// number of phases.  One workgroup per evaluation point.
__global__ __launch_bounds__(MPX_EA_THREADS) void mpx_equal_area_kernel(const double* __restrict__ resid, int64_t n, int nx, const double* __restrict__ p_in,
                                                             double* __restrict__ p_out, int64_t p_stride_in, int64_t p_stride_out, int S, int seg_off,
                                                             double damping, double* __restrict__ cum_all, int cum_in_lds, int B) {
  extern __shared__ double s_dyn[];  // cum_in_lds: [n] cumulative areas, then [S + 1] boundaries; else only the boundaries
  constexpr int NT = MPX_EA_THREADS;
  __shared__ double wave_tot[NT / 64];
  __shared__ double total;
  const int l = threadIdx.x;
  const int64_t pos_off = cum_in_lds ? n : 0;
  double* __restrict__ pos = s_dyn + pos_off;
  const int64_t m = n - 1, chunk = (m + NT - 1) / NT;  // m trapezoids; lane l owns the trapezoids [i0, i1)
  const int64_t i0 = l * chunk < m ? l * chunk : m, i1 = i0 + chunk < m ? i0 + chunk : m;
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    const double* __restrict__ r = resid + (int64_t)b * n * nx;
    double* __restrict__ cum = cum_in_lds ? s_dyn : cum_all + (int64_t)b * n;  // cum[i] = area of the first i trapezoids
    auto norm2 = [&](int64_t i) {
      if (nx == 1) return fabs(r[i]);
      double q = 0;
      for (int a = 0; a < nx; ++a) q = fma(r[i * nx + a], r[i * nx + a], q);
      return sqrt(q);
    };
    if (cum_in_lds) {  // the residual norms enter LDS with coalesced loads; the scan below turns them into cumulative areas in place
      for (int64_t i = l; i < n; i += 8 * NT) {
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = i + k * NT < n ? norm2(i + k * NT) : 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (i + k * NT < n) cum[i + k * NT] = v[k];
      }
      __syncthreads();
    }
    auto sample = [&](int64_t i) { return cum_in_lds ? cum[i] : norm2(i); };
    const double first = i0 < i1 ? sample(i0) : 0.0;  // (read before the in-place pass of the neighbouring lane overwrites it)
    double tot = 0, prev = first;
    for (int64_t i = i0; i < i1; ++i) {
      const double nxt = sample(i + 1);
      tot += 0.5 * (prev + nxt);
      prev = nxt;
    }
    const double inc = wave_scan_inclusive(tot);
    if ((l & 63) == 63) wave_tot[l >> 6] = inc;
    __syncthreads();
    double off = wave_shift_up_1(inc);
    for (int q = 0; q < (l >> 6); ++q) off += wave_tot[q];
    if (l == NT - 1) total = off + tot;
    __syncthreads();
    const double inv = 1.0 / total;
    if (l == 0) cum[0] = 0.0;  // (lane 0 holds sample 0 in `first`)
    prev = first;
    for (int64_t i = i0; i < i1; ++i) {
      const double nxt = sample(i + 1);  // position i + 1 is overwritten two lines down, by this lane only
      off += 0.5 * (prev + nxt);
      prev = nxt;
      cum[i + 1] = i + 1 == m ? 1.0 : off * inv;  // (the reference divides by the last entry: exactly 1 there)
    }
    __syncthreads();
    if (l == 0) pos[0] = 0.0;
    // lane l owns a contiguous run of boundaries: one binary search for the first, then a forward walk (targets are monotone)
    const int per = (S + NT - 1) / NT, s0 = l * per < S ? l * per : S, s1 = s0 + per < S ? s0 + per : S;
    int64_t j = 1;
    for (int s = s0; s < s1; ++s) {
      const double target = (double)(s + 1) / (double)S;
      if (s == s0) {
        int64_t lo = 0, hi = m;  // first j with cum[j] >= target
        while (lo < hi) {
          const int64_t mid = (lo + hi) >> 1;
          if (cum[mid] >= target) hi = mid; else lo = mid + 1;
        }
        j = lo < 1 ? 1 : lo;
      } else {
        while (j < m && cum[j] < target) ++j;
      }
      pos[s + 1] = ((double)(j - 1) + (target - cum[j - 1]) / (cum[j] - cum[j - 1])) / (double)m;
    }
    __syncthreads();
    const double* __restrict__ pi = p_in + (int64_t)b * p_stride_in + seg_off;
    double* __restrict__ po = p_out + (int64_t)b * p_stride_out + seg_off;
    for (int s = l; s < S; s += 8 * NT) {
      double v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = s + k * NT < S ? pi[s + k * NT] : 0.0;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (s + k * NT < S) po[s + k * NT] = damping * (pos[s + k * NT + 1] - pos[s + k * NT]) + (1.0 - damping) * v[k];
    }
    __syncthreads();  // LDS is rewritten by the next point
  }
}

// Fast kernel: n <= 12 * 1024 samples, S <= 4 * 1024 segments per phase (the config-5 protocol and everything of its size; scalar or
// vector residuals -- the 2-norms are formed while the next point is fetched --, one call per phase).
//  * persistent: one workgroup per compute unit; the NEXT point's samples are fetched into registers (coalesced) before the current
//    point is scanned and searched, so the load phase disappears behind the rest;
//  * the samples live in LDS in rows of `chunk` (= the trapezoids of one lane) padded to an odd number of doubles: a lane reads its
//    row into registers and writes the cumulative areas back over it without bank conflicts (the unpadded layout of the generic
//    kernel serialises every access four-fold at chunk = 12), one pass over LDS instead of two;
//  * every new boundary is found by its own branch-free binary search, the lane's four searches interleaved (targets l, l + 1024,
//    ...): the walk of the generic kernel is as long as the flattest stretch of the residual curve -- the slowest lane set the pace;
//  * the exclusive prefix sums of the new widths -- what mpx_prefix_kernel would compute from p_out, same additions in the same
//    order (prefix_scan_block) -- are left in `wcum`, so that the next evaluation needs no prefix launch (MPX_WIDTHS_UNCHANGED).
// Same rule, same searches (first j with cum[j] >= target) as the generic kernel; the cumulative sums associate differently.
template <bool SCALAR>
__global__ __launch_bounds__(MPX_EA_THREADS) void mpx_equal_area_fast_kernel(const double* __restrict__ resid, int n, const double* __restrict__ p_in,
                                                                  double* __restrict__ p_out, int64_t p_stride_in, int64_t p_stride_out, int S,
                                                                  double damping, double* __restrict__ wcum, int64_t wcum_stride, int B, int chunk,
                                                                  unsigned magic, int pad, int pos_off, int nx, int seg_off, int want_prefix, long long* dbg) {
#ifdef MPX_EA_STAMPS  // phase stamps of the second point of workgroup 0 (-DMPX_EA_STAMPS + MPX_EA_DEBUG=1)
#define MPX_EA_STAMP(k) if (dbg && threadIdx.x == 0 && blockIdx.x == 0 && b == (int)gridDim.x * ((B - 1) / (int)gridDim.x > 0 ? 1 : 0)) dbg[k] = wall_clock64()
#else
#define MPX_EA_STAMP(k)
#endif
  extern __shared__ double s_dyn[];  // padded samples / cumulative areas, then [S + 1] boundaries at pos_off
  constexpr int NT = MPX_EA_THREADS, PF = MPX_EA_PF, WR = MPX_EA_WR;
  __shared__ double wave_tot[NT / 64];
  __shared__ double pre_tot[MPX_PREFIX_THREADS / 64];
  const int m = n - 1;
  const double inv_m = 1.0 / (double)m;
  auto phys = [&](int i) { return i + (pad ? (int)__umulhi((unsigned)i, magic) : 0); };  // i + i / chunk (exact for i < 2^32 / chunk)
  double* __restrict__ cum = s_dyn;
  double* __restrict__ pos = s_dyn + pos_off;
  int* __restrict__ jmap = reinterpret_cast<int*>(pos);  // [WR * NT] first-target marks: live between the area scan and the boundaries
  __shared__ int wave_j[NT / 64];
  double pf[PF];
  // The prefetch must reach its use without a control-flow merge in between: with `if (next point exists) fetch(...)`, a run-time
  // `nx == 1` and a guard per load the fetched values met the old ones in phi nodes, the register allocator resolved those with
  // copies right behind the loads, and a copy reads its source -- s_waitcnt vmcnt(0) two instructions after the last load was
  // issued: the whole fetch was exposed at every point of a batch (16.8 us per point against 9.8 us of phases).  So: SCALAR is a
  // template parameter, every lane issues all PF loads (indices clamped to the last sample: the surplus ones of a short sample
  // list hit one line), and the last point of a workgroup fetches itself again.
  auto fetch = [&](int b, int l, int k0 = 0, int k1 = PF) {  // (k0, k1: literals at the call sites)
    const double* __restrict__ r = resid + (int64_t)b * n * nx;
    if constexpr (SCALAR) {
#pragma unroll
      for (int k = 0; k < PF; ++k)
        if (k >= k0 && k < k1 && (!MPX_EA_COND_PREFETCH || k * NT < n)) pf[k] = r[min(k * NT + l, m)];
    } else {  // vector residuals: the 2-norm of a sample, accumulated like the generic kernel's (fma over the components, then sqrt)
#pragma unroll
      for (int k = 0; k < PF; ++k)
        if (k * NT < n) {
          const double* __restrict__ ri = r + (int64_t)min(k * NT + l, m) * nx;
          double q = 0;
          for (int a = 0; a < nx; ++a) q = fma(ri[a], ri[a], q);
          pf[k] = sqrt(q);
        }
    }
  };
  fetch(min((int)blockIdx.x, B - 1), threadIdx.x);  // (the host launches at most B workgroups)
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    // (the lane id is opaque per point: everything derived from it is recomputed here with a few integer operations instead of
    // being hoisted out of the loop into registers the 128-VGPR budget of a 1024-lane workgroup does not have)
    int l = threadIdx.x;
    asm volatile("" : "+v"(l));
    MPX_EA_STAMP(0);
#pragma unroll
    for (int k = 0; k < PF; ++k)
      if (k * NT < n) cum[phys(k * NT + l)] = fabs(pf[k]);  // (slots past sample m are never read: the host sized the rows for them)
    reinterpret_cast<int4*>(jmap)[l] = make_int4(0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff);  // (the boundaries of the previous point are spent)
    __syncthreads();
    // Scalar residuals: the next point's samples are requested in MPX_EA_SLICES slices, one behind each of the first phases.  All
    // workgroups of the launch pass through the same phase at the same time: requested at once, the 25 MB of a round of 256 points
    // met an idle memory system, filled the compute units' request queues and held every wavefront at its load instructions until
    // HBM had delivered (stamps at B = 2048: the phase behind the fetch 2.2 -> 5.5 us).
    constexpr int SL = SCALAR && !MPX_EA_COND_PREFETCH ? MPX_EA_SLICES : 1, PS = (PF + SL - 1) / SL;
    const int b_next = min(b + (int)gridDim.x, B - 1);
    if constexpr (SCALAR && !MPX_EA_COND_PREFETCH) fetch(b_next, l, 0, PS);  // in flight during the scan and the search of this point
    else if (b + (int)gridDim.x < B) fetch(b + gridDim.x, l);
    double pin[WR];  // the lane's old widths: requested now, used after the search
    {
      const double* __restrict__ pi_ = p_in + (int64_t)b * p_stride_in + seg_off;
#pragma unroll
      for (int k = 0; k < WR; ++k) pin[k] = pi_[min(l + k * NT, S - 1)];
    }
    MPX_EA_STAMP(1);
    // lane l owns the trapezoids [i0, i0 + cnt) = its row of the padded layout; its samples i0 ... i0 + cnt are read twice (row sums,
    // then cumulative areas written back over them); the last one is the next lane's first -- the first slot of the next row,
    // overwritten by THIS lane only, so the next lane keeps its copy (`first`)
    const int i0 = l * chunk < m ? l * chunk : m, cnt = (i0 + chunk < m ? i0 + chunk : m) - i0;
    const int row = cnt > 0 ? i0 + (pad ? l : 0) : 0;  // phys(i0); lanes without trapezoids read row 0 and use nothing of it
    const int last = row + chunk + pad;                // slot of sample i0 + chunk
    // (the row is fetched whole, then used: thirteen LDS reads in flight instead of a read, a wait and an addition thirteen times;
    // slots past the row's end repeat `last` and are masked by t < cnt)
    double a[PF + 1];
    auto load_row = [&]() {
      a[0] = cum[row];
#pragma unroll
      for (int t = 0; t < PF; ++t) a[t + 1] = cum[t + 1 < chunk ? row + t + 1 : last];
    };
    load_row();
    const double first = a[0];
    double tot = 0;
#pragma unroll
    for (int t = 0; t < PF; ++t) tot += t < cnt ? 0.5 * (a[t] + a[t + 1]) : 0.0;
    const double inc = wave_scan_inclusive(tot);
    if ((l & 63) == 63) wave_tot[l >> 6] = inc;
    __syncthreads();  // (also: every lane has read its `first`)
    if constexpr (SL > 1) fetch(b_next, l, PS, 2 * PS);
    double off = wave_shift_up_1(inc);
    double total = 0;
#pragma unroll
    for (int q = 0; q < NT / 64; ++q) {
      if (q == (l >> 6)) off += total;
      total += wave_tot[q];
    }
    MPX_EA_STAMP(2);
    const double inv = 1.0 / total;
    load_row();
    a[0] = first;
#pragma unroll
    for (int t = 0; t < PF; ++t) {
      off += 0.5 * (a[t] + a[t + 1]);
      if (t < cnt) cum[t + 1 < chunk ? row + t + 1 : last] = off * inv;
    }
    if (l == 0) cum[0] = 0.0;
    if (cnt > 0 && i0 + cnt == m) cum[cnt < chunk ? row + cnt : last] = 1.0;  // (the reference divides by the last entry: exactly 1 there)
    __syncthreads();
    if constexpr (SL > 2) fetch(b_next, l, 2 * PS, 3 * PS);
    MPX_EA_STAMP(3);
    // New boundaries without a search.  kc(j) = number of targets T_s = (s + 1) / S not above cum[j] is a product and a floor
    // (an fma and a division settle the rare products within rounding of an integer); sample j is the first one at or above T_s
    // exactly for s in [kc(j - 1), kc(j)), so every lane marks, for its own samples, the FIRST such target (jmap, aliased on the
    // boundaries, cleared above) and a running maximum over the targets -- four per lane, a DPP scan per wavefront, sixteen
    // wavefront totals -- fills in the rest.  (14 dependent, divergent LDS probes per target, conflicts included, were 6 of the
    // 13.6 us of a point: profiles/r3_config5_loop.)  Same answer as the generic kernel's searches: first j with
    // cum[j] >= fl((s + 1) / S), j >= 1.
    {
      const double Sd = (double)S;
      auto not_above = [&](double c) {  // #{s in [0, S): fl((s + 1) / S) <= c}, c in [0, 1]
        const double x = c * Sd, xf = floor(x), d = x - xf;
        int q = (int)xf;
        if (d == 0.0 || d > 1.0 - 2e-12) {  // (rare: the rounded product is an integer, or within rounding below the next one)
          if (fma(c, Sd, -xf) < 0.0) --q;   // the product was rounded up to an integer: q = floor(c S) exactly now
          // s + 1 <= q: (s + 1) / S <= c before rounding, hence after.  s + 1 = q + 1 is above c, but the quotient may round down to it
          if (q < S && (double)(q + 1) / Sd <= c) ++q;
        }
        return q < S ? q : S;
      };
      load_row();
      int kprev = not_above(a[0]);
#pragma unroll
      for (int t = 0; t < PF; ++t) {
        const int kc = not_above(a[t + 1]);
        if (t < cnt && kc > kprev) atomicMin(&jmap[kprev], i0 + t + 1);
        kprev = kc;
      }
    }
    MPX_EA_STAMP(7);
    __syncthreads();
    if constexpr (SL > 3) fetch(b_next, l, 3 * PS, PF);
    MPX_EA_STAMP(8);
    int jj[WR];
    {
      const int4 mk = reinterpret_cast<const int4*>(jmap)[l];  // the marks of targets 4 l ... 4 l + 3 (0x7fffffff: none)
      jj[0] = mk.x, jj[1] = mk.y, jj[2] = mk.z, jj[3] = mk.w;
      static_assert(WR == 4, "one 16-byte read per lane");
#pragma unroll
      for (int k = 0; k < WR; ++k) jj[k] = jj[k] == 0x7fffffff ? 0 : jj[k];
#pragma unroll
      for (int k = 1; k < WR; ++k) jj[k] = max(jj[k], jj[k - 1]);
      const int inc_j = wave_max_scan_inclusive(jj[WR - 1]);
      if ((l & 63) == 63) wave_j[l >> 6] = inc_j;
      int before = __builtin_amdgcn_update_dpp(0, inc_j, 0x138, 0xf, 0xf, false);  // wave_shr:1
      __syncthreads();  // (also: every lane has read its marks; the boundaries may overwrite them)
#pragma unroll
      for (int q = 0; q < NT / 64; ++q)
        if (q < (l >> 6)) before = max(before, wave_j[q]);
#pragma unroll
      for (int k = 0; k < WR; ++k) jj[k] = max(jj[k], before);
    }
    MPX_EA_STAMP(9);
#pragma unroll
    for (int k = 0; k < WR; ++k) {
      const int s = WR * l + k;
      const int j = jj[k] < 1 ? 1 : jj[k];
      const double target = (double)(WR * (int)threadIdx.x + k + 1) / (double)S;  // (of the lane, not of the point: hoisted)
      const double c0 = cum[phys(j - 1)], c1 = cum[phys(j)];
      if (s < S) pos[s + 1] = ((double)(j - 1) + (target - c0) / (c1 - c0)) * inv_m;
    }
    if (l == 0) pos[0] = 0.0;
    __syncthreads();
    MPX_EA_STAMP(4);
    double* __restrict__ po = p_out + (int64_t)b * p_stride_out + seg_off;
    // the new widths also replace the boundaries in LDS (registers first: a lane's pos[s + 1] is its neighbour's pos[s]), then
    // the workgroup scans them exactly as mpx_prefix_kernel scans p_out (MPX_PREFIX_THREADS == MPX_EA_THREADS)
    double wn[WR];
#pragma unroll
    for (int k = 0; k < WR; ++k) {
      const int s = min(l + k * NT, S - 1);
      wn[k] = damping * (pos[s + 1] - pos[s]) + (1.0 - damping) * pin[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < WR; ++k) {
      const int s = l + k * NT;
      if (s < S) pos[s] = wn[k], po[s] = wn[k];
    }
    __syncthreads();
    MPX_EA_STAMP(5);
    // (only for problems with a node function that uses the node time: nothing else reads the prefix sums -- 1.8 of 11.5 us per point)
    if (want_prefix) prefix_scan_block([&](int s) { return pos[s]; }, wcum + (int64_t)b * wcum_stride + seg_off, S, l, pre_tot);
    MPX_EA_STAMP(6);
    __syncthreads();  // LDS is rewritten by the next point
  }
#undef MPX_EA_STAMP
}
}  // namespace

extern "C" int mpx_equal_area_widths_device(mpx_ctx* c, int phase, int64_t batch, int64_t n_pts, const double* resid, const double* p_in,
                                            int p_in_per_point, double* p_out, double damping) {
  if (!c || !resid || !p_in || !p_out || batch < 1 || n_pts < 2) return fail(c, MPX_ERR_INVALID, "mpx_equal_area_widths_device: bad arguments");
  if (c->kind != 0 || phase < 0 || phase >= c->n_phases) return fail(c, MPX_ERR_INVALID, "mpx_equal_area_widths_device: phase out of range");
  if (!c->has_device) return fail(c, MPX_ERR_NO_DEVICE, "mpx_equal_area_widths_device: context has no device code; there is no CPU fallback");
  HIPCHK(c, hipSetDevice(c->device));
  // cumulative areas in LDS when they fit next to the S + 1 boundaries (150 of the 160 KB of a compute unit), else in HBM scratch
  const size_t lds_all = (size_t)(n_pts + c->S + 1) * 8, lds_pos = (size_t)(c->S + 1) * 8;
  const int in_lds = lds_all <= 150 * 1024;
  if (lds_pos > 150 * 1024) return fail(c, MPX_ERR_UNSUPPORTED, "mpx_equal_area_widths_device: more than 19199 segments per phase");
  // the fast kernel (it leaves the prefix sums of the phase's new widths for the next evaluation; MPX_WIDTHS_UNCHANGED is the caller's
  // word that every phase has been updated): rows of `chunk` samples padded to an odd stride
  const int chunk = (int)((n_pts - 1 + MPX_EA_THREADS - 1) / MPX_EA_THREADS), pad = chunk % 2 == 0;
  // (rows for every staged slot: the lanes stage ceil(n / 1024) * 1024 samples, the ones past the last sample are never read)
  const int64_t staged = (n_pts + MPX_EA_THREADS - 1) / MPX_EA_THREADS * MPX_EA_THREADS;
  const int64_t pos_off = (staged + (pad ? staged / chunk : 0) + 2) & ~(int64_t)1;
  const size_t lds_fast = (size_t)(pos_off + std::max<int64_t>(c->S + 1, MPX_EA_WR * MPX_EA_THREADS / 2)) * 8;  // (boundaries, or the marks they alias)
  const bool fast = n_pts <= (int64_t)MPX_EA_PF * MPX_EA_THREADS && c->S <= MPX_EA_WR * MPX_EA_THREADS &&
                    lds_fast <= 150 * 1024 && !mpx_knob(MPX_K_EA_GENERIC);
  int rc;
  if (!fast && !in_lds && (rc = reserve(c, c->ea_scratch, (size_t)(batch * n_pts)))) return rc;
  const size_t lds = fast ? lds_fast : in_lds ? lds_all : lds_pos;
  long long*& dbg = c->ea_dbg;  // per context: freed in mpx_destroy
  if (!dbg && mpx_knob(MPX_K_EA_DEBUG)) HIPCHK(c, hipHostMalloc((void**)&dbg, 128, hipHostMallocMapped));
  if (lds > 48 * 1024 && lds > c->ea_lds_allowed) {  // the attribute is per device: remembered per context, not per process
    HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(mpx_equal_area_fast_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(mpx_equal_area_fast_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(mpx_equal_area_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    c->ea_lds_allowed = 150 * 1024;
  }
  if (fast) {
    const size_t wcap = c->wcum.cap;
    if ((rc = reserve_wcum(c, (size_t)(batch * c->n_p)))) return rc;
    // the prefix sums of this phase's new widths are left in wcum: they describe p_out (per point, this batch)
    if (wcap != c->wcum.cap || c->wcum_p != p_out || c->wcum_batch != batch || c->wcum_ppp != 1) c->wcum_phases = 0;
    // (the kernel writes the prefix sums only for time-dependent problems -- want_prefix below; otherwise the phase's bit is CLEARED:
    // the bookkeeping must never say the buffer holds sums it does not hold)
    c->wcum_p = p_out, c->wcum_batch = batch, c->wcum_ppp = 1;
    if (c->time_dep) c->wcum_phases |= 1u << phase;
    else c->wcum_phases &= ~(1u << phase);
    int n_cu = 256;
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->device);
    const unsigned grid = (unsigned)std::min<int64_t>(batch, n_cu);  // persistent: a workgroup owns its compute unit's LDS
    const unsigned magic = (unsigned)((((uint64_t)1 << 32) + chunk - 1) / chunk);  // i / chunk = umulhi(i, magic) for i < 2^32 / chunk
    // (scalar residuals -- one state -- have their own instantiation: the prefetch of the next point must not pass a run-time branch)
    hipLaunchKernelGGL(c->nx == 1 ? mpx_equal_area_fast_kernel<true> : mpx_equal_area_fast_kernel<false>, dim3(grid), dim3(MPX_EA_THREADS), lds, c->stream,
                       resid, (int)n_pts, p_in, p_out, (int64_t)(p_in_per_point ? c->n_p : 0), (int64_t)c->n_p, c->S, damping, c->wcum.p, (int64_t)c->n_p,
                       (int)batch, chunk, magic, pad, (int)pos_off, c->nx, phase * c->S, c->time_dep ? 1 : 0, dbg);
  } else {
    if (c->wcum_p == p_out) c->wcum_phases &= ~(1u << phase);  // the generic kernel changes the widths and leaves no prefix sums
    hipLaunchKernelGGL(mpx_equal_area_kernel, dim3((unsigned)batch), dim3(MPX_EA_THREADS), lds, c->stream, resid, n_pts, c->nx, p_in, p_out,
                       (int64_t)(p_in_per_point ? c->n_p : 0), (int64_t)c->n_p, c->S, phase * c->S, damping, c->ea_scratch.p, in_lds, (int)batch);
  }
  HIPCHK(c, hipGetLastError());
  // the context's prefix sums now belong to p_out: a following mpx_eval_device(... | MPX_WIDTHS_UNCHANGED, p = p_out, per point,
  // same batch) may use them (the caller's assertion, as always with that flag)
  c->wcum_valid = false;
  if (dbg) {  // MPX_EA_DEBUG: phase stamps of the last workgroup (wall_clock64, 100 MHz)
    HIPCHK(c, hipStreamSynchronize(c->stream));
    fprintf(stderr, "equal_area phases (us): stage %.2f  row sums %.2f  cumulative areas %.2f  search %.2f  widths %.2f  prefix %.2f\n", (dbg[1] - dbg[0]) / 100.0,
            (dbg[2] - dbg[1]) / 100.0, (dbg[3] - dbg[2]) / 100.0, (dbg[4] - dbg[3]) / 100.0, (dbg[5] - dbg[4]) / 100.0, (dbg[6] - dbg[5]) / 100.0);
    fprintf(stderr, "  search = marks %.2f  barrier %.2f  running maximum %.2f  boundaries %.2f\n", (dbg[7] - dbg[3]) / 100.0, (dbg[8] - dbg[7]) / 100.0,
            (dbg[9] - dbg[8]) / 100.0, (dbg[4] - dbg[9]) / 100.0);
  }
  return MPX_OK;
}
