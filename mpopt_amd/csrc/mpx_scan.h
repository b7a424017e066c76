// mpx_scan.h -- wavefront / workgroup scans on the DPP path, shared by the prefix kernel of the segment widths (mpx_host.cpp) and the
// equal-area kernels (mpx_equal_area.cpp): ONE definition of the additions and their order, so that the prefix sums the equal-area
// update leaves behind are the bits mpx_prefix_kernel would compute (MPX_WIDTHS_UNCHANGED, tested).  Device code only.
#ifndef MPX_SCAN_H
#define MPX_SCAN_H
#include <hip/hip_runtime.h>

namespace mpxi {
// exclusive prefix sums of the segment widths (the reference's running t_seg0, mpopt.py:192): one workgroup per
// (width vector, phase).  Each of the sixteen wavefronts owns a contiguous share and walks it 64 elements at a time with
// coalesced loads: first the quarter totals (so that every wavefront knows its starting offset), then the scan proper --
// shuffle scan inside the 64 elements, running carry across them.  Fixed order: results do not depend on anything else.
// (the scan itself is a device function: mpx_equal_area_kernel runs it on the widths it has just produced, with the same
// additions in the same order, so that the prefix sums it leaves behind are the ones this kernel would compute)
// Wavefront scans on the DPP path (row shifts inside the 16-lane rows, then the row broadcasts 15 / 31): six v_mov_dpp pairs + six
// additions, no LDS crossbar (__shfl_up costs two ds_bpermute per level and their latency six times in a row: the scans of the
// equal-area kernel spent most of their time there, profiles/r3_config5_loop).  Lanes without a source add +0.0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_or_zero(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_scan_inclusive(double x) {
  x += dpp_or_zero<0x111, 0xf>(x);  // row_shr:1
  x += dpp_or_zero<0x112, 0xf>(x);  // row_shr:2
  x += dpp_or_zero<0x114, 0xf>(x);  // row_shr:4
  x += dpp_or_zero<0x118, 0xf>(x);  // row_shr:8
  x += dpp_or_zero<0x142, 0xa>(x);  // row_bcast:15 into rows 1 and 3
  x += dpp_or_zero<0x143, 0xc>(x);  // row_bcast:31 into rows 2 and 3
  return x;
}
__device__ __forceinline__ int wave_max_scan_inclusive(int x) {  // x >= 0; lanes without a source contribute 0
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false));
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false));
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false));
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false));
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false));
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false));
  return x;
}
__device__ __forceinline__ double wave_shift_up_1(double x) { return dpp_or_zero<0x138, 0xf>(x); }  // wave_shr:1 (lane 0: +0.0)
__device__ __forceinline__ double wave_last(double x) {  // lane 63's value, in every lane
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 63), __builtin_amdgcn_readlane(__double2loint(x), 63));
}
#define MPX_PREFIX_THREADS 1024
template <class Load>  // a(s): the s-th width (global memory in mpx_prefix_kernel, LDS in mpx_equal_area_kernel)
__device__ __forceinline__ void prefix_scan_block(Load a, double* __restrict__ o, int S, int tid, double* wave_tot) {
  constexpr int NWV = MPX_PREFIX_THREADS / 64;
  const bool active = true;
  const int lane = tid & 63, wave = tid >> 6;
  const int quarter = ((S + NWV - 1) / NWV + 63) / 64 * 64;  // a wavefront's share: a multiple of 64, every step is one aligned run
  const int q0 = wave * quarter, q1 = active ? min(S, q0 + quarter) : 0;
  double tot = 0;
  for (int s0 = q0 + lane; s0 < q1; s0 += 64 * 8) {  // eight loads in flight, added in index order
    double v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = s0 + k * 64 < q1 ? a(s0 + k * 64) : 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (s0 + k * 64 < q1) tot += v[k];
  }
  tot = wave_last(wave_scan_inclusive(tot));
  if (active && lane == 0) wave_tot[wave] = tot;
  __syncthreads();
  double carry = 0;
  if (active)
    for (int q = 0; q < wave; ++q) carry += wave_tot[q];
  // eight 64-element steps at a time: their loads are in flight together, the scan itself (same additions, same order as one
  // step at a time) runs on registers
  for (int s0 = q0; s0 < q1; s0 += 64 * 8) {
    double v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int s = s0 + k * 64 + lane;
      v[k] = s < q1 ? a(s) : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int s = s0 + k * 64 + lane;
      const double inc = wave_scan_inclusive(v[k]);
      if (s < q1) o[s] = carry + wave_shift_up_1(inc);
      carry += wave_last(inc);
    }
  }
}

}  // namespace mpxi
#endif
