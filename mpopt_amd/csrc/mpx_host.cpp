// mpx_host.cpp -- libmpx host runtime (the planner -- structure -> tiles, index maps, COO patterns -- is mpx_layout.cpp):
// device tables; kernel launches; the C ABI of include/mpx.h.
//
// Reference behaviour restated here (structure only -- all arithmetic is in mpx_kernels.h):
//   decision vector layout          mpopt.py:537-543, 627   (state-major, phases concatenated)
//   constraint row order            mpopt.py:458, 617-621   ([F;C;DU;mU;dU;TC] per phase, events)
//   composite D / W / interpolation mpopt.py:4015-4131      (never formed densely: per-degree
//                                                            tables + per-node (segment, point))
//   node ownership                  mpopt.py:189-195, 208   (shared node belongs to the earlier
//                                                            segment; later segments drop w_0)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <mutex>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <numeric>
#include <string>
#include <vector>

#include "mpx.h"
#include "mpx_device.h"

#include "mpx_internal.h"
#include "mpx_scan.h"

using namespace mpxi;

namespace mpxi {
std::string& create_error() {
  thread_local std::string e;
  return e;
}
}  // namespace mpxi

// ---- knobs of the evaluation path (mpx_internal.h) -------------------------------------------------------------------------------------
namespace {
const char* const kKnobNames[MPX_K_COUNT] = {
    "MPX_BPB", "MPX_NO_LIGHT", "MPX_LIGHT_LONG_SPANS", "MPX_NO_PACKED_G", "MPX_NO_PHASE_MERGE", "MPX_LIGHT_DEBUG", "MPX_LIGHT_PER_CU",
    "MPX_RESIDENT", "MPX_NO_RESIDENT", "MPX_GRADL_GENERIC", "MPX_NO_FUSE", "MPX_FUSE_MIN_BATCH", "MPX_FUSE_PT_STAMPS", "MPX_NO_LANES",
    "MPX_LANES_MIN_BATCH", "MPX_LANES_ORDER", "MPX_ASM_PASS_MB", "MPX_EA_GENERIC", "MPX_EA_DEBUG", "MPX_GRADL_BPB"};
std::atomic<int> g_env_dynamic{0};
std::string g_knob_val[MPX_K_COUNT];
bool g_knob_set[MPX_K_COUNT];
std::once_flag g_knob_once;
void knob_snapshot() {
  for (int k = 0; k < MPX_K_COUNT; ++k) {
    const char* v = getenv(kKnobNames[k]);
    g_knob_set[k] = v != nullptr;
    g_knob_val[k] = v ? v : "";
  }
}
}  // namespace

const char* mpx_knob(MpxKnob k) {
  if (g_env_dynamic.load(std::memory_order_relaxed)) return getenv(kKnobNames[k]);
  std::call_once(g_knob_once, [] {
    if (getenv("MPX_ENV_DYNAMIC")) g_env_dynamic.store(1);
    else knob_snapshot();
  });
  if (g_env_dynamic.load(std::memory_order_relaxed)) return getenv(kKnobNames[k]);
  return g_knob_set[k] ? g_knob_val[k].c_str() : nullptr;
}

extern "C" const char* mpx_env_knob(const char* name) {
  for (int k = 0; name && k < MPX_K_COUNT; ++k)
    if (!strcmp(name, kKnobNames[k])) return mpx_knob((MpxKnob)k);
  return nullptr;
}

extern "C" int mpx_env_dynamic(int on) {
  std::call_once(g_knob_once, [] {});
  if (!on) knob_snapshot();  // (single-threaded moment by contract: no evaluation in flight)
  g_env_dynamic.store(on ? 1 : 0);
  return MPX_OK;
}
#define g_create_error (mpxi::create_error())

namespace {

MpxNodeArgs node_args_static(const mpx_ctx* c, const Bucket& B, bool absorber);
MpxBoundArgs bound_args_static(const mpx_ctx* c);

int load_device(mpx_ctx* c, const mpx_problem* prob) {
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipModuleLoadData(&c->module, prob->code_object));
  {  // does any node function use the node time?  (code objects generated before the symbol existed: assume yes)
    hipDeviceptr_t sym = nullptr;
    size_t bytes = 0;
    int v = 1;
    if (hipModuleGetGlobal(&sym, &bytes, c->module, "mpx_time_dependent") == hipSuccess && bytes == sizeof(int) && hipMemcpyDtoH(&v, sym, sizeof(int)) == hipSuccess)
      c->time_dep = v != 0 || getenv("MPX_ALWAYS_PREFIX") != nullptr;
    (void)hipGetLastError();
  }
  {  // the degree from which the node kernels stream their tables is a build-time constant of the code object: the host lays the
     // tables out for it (transposed copies) and sized the LDS plans with its own value -- the two must agree
    hipDeviceptr_t sym = nullptr;
    size_t bytes = 0;
    int v = -1;
    if (hipModuleGetGlobal(&sym, &bytes, c->module, "mpx_tables_stream_above") != hipSuccess || bytes != sizeof(int) || hipMemcpyDtoH(&v, sym, sizeof(int)) != hipSuccess)
      v = -1;
    (void)hipGetLastError();
    if (v != c->stream_above)
      return fail(c, MPX_ERR_INVALID, "code object built with MPX_TABLES_STREAM_ABOVE=%d, the context expects %d (set the environment variable for both, "
                  "mpx_device.h)", v, c->stream_above);
  }
  static const char* modes[3] = {"fg", "fgj", "hess"};
  for (auto& B : c->buckets)
    for (int m = 0; m < 3; ++m) {
      char name[96];
      snprintf(name, sizeof name, "mpx_node_%s_%d_%d", modes[m], B.phase, B.deg);
      hipError_t e = hipModuleGetFunction(&B.fn[m], c->module, name);
      if (e != hipSuccess) return fail(c, MPX_ERR_INVALID, "code object lacks kernel %s (%s)", name, hipGetErrorString(e));
    }
  for (int m = 0; m < 3; ++m) {
    char name[64];
    snprintf(name, sizeof name, "mpx_boundary_%s", modes[m]);
    hipError_t e = hipModuleGetFunction(&c->fn_bound[m], c->module, name);
    if (e != hipSuccess) return fail(c, MPX_ERR_INVALID, "code object lacks kernel %s", name);
  }
  if (c->n_phases > 1 && c->degs.size() == 1) {  // all phases in one launch (optional kernels)
    static const char* lm[2] = {"fg", "fgq"};
    char name[96];
    for (int m = 0; m < 3; ++m) {
      snprintf(name, sizeof name, "mpx_node_%s_all_%d", modes[m], c->degs[0].deg);
      if (hipModuleGetFunction(&c->fn_node_all[m], c->module, name) != hipSuccess) c->fn_node_all[m] = nullptr, (void)hipGetLastError();
    }
    for (int m = 0; m < 2 && c->lplan.low; ++m) {
      snprintf(name, sizeof name, "mpx_lightlow_%s_all_%d", lm[m], c->degs[0].deg);
      if (hipModuleGetFunction(&c->fn_lightlow_all[m], c->module, name) != hipSuccess) c->fn_lightlow_all[m] = nullptr, (void)hipGetLastError();
      snprintf(name, sizeof name, "mpx_lightlows_%s_all_%d", lm[m], c->degs[0].deg);
      if (hipModuleGetFunction(&c->fn_lightlows_all[m], c->module, name) != hipSuccess) c->fn_lightlows_all[m] = nullptr, (void)hipGetLastError();
    }
  }
  // row spans of more than the default 64 KB of LDS per workgroup: raise the dynamic shared memory limit of the absorbing kernels
  for (auto& B : c->buckets) {
    const int64_t dyn = (int64_t)B.abs_cap * B.abs_slots * 8;
    if (!c->absorb || B.abs_cap <= 0 || B.abs_lds_static + dyn <= 65536) continue;
    for (int m = 0; m < 2; ++m)
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(B.fn[m]), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess) {
        (void)hipGetLastError();
        c->notes += "mixed-degree grid: the runtime refused " + std::to_string(dyn / 1024) + " KB of dynamic LDS for the row spans: unpack pass instead\n";
        c->absorb = false;
        break;
      }
  }
  // nlp_grad kernels (code objects generated before they existed lack them: mpx_eval_grad_gamma then says so)
  for (auto& B : c->buckets) {
    char name[96];
    snprintf(name, sizeof name, "mpx_node_gradl_%d_%d", B.phase, B.deg);
    if (hipModuleGetFunction(&B.fn_gradl, c->module, name) != hipSuccess) B.fn_gradl = nullptr, (void)hipGetLastError();
    static const char* lm[2] = {"fg", "fgq"};
    for (int m = 0; m < 2 && ((B.deg > 12 && B.deg <= 31) || c->lplan.low || c->lplan.high); ++m) {  // light passes (mpx_kernels.h: light_body / light_low_body / light_high_body)
      snprintf(name, sizeof name, "mpx_light%s_%s_%d_%d", c->lplan.low ? "low" : (c->lplan.high ? "high" : ""), lm[m], B.phase, B.deg);
      if (hipModuleGetFunction(&B.fn_light[m], c->module, name) != hipSuccess) B.fn_light[m] = nullptr, (void)hipGetLastError();
      if (c->lplan.high && B.fn_light[m]) {  // the input tile of a workgroup: dynamic LDS, past the 64 KB a launch gets by default from degree ~160
        const int dyn = (int)((int64_t)(c->nx + c->nu) * c->lplan.span_cap * 17 * 8);
        if (dyn > 60 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(B.fn_light[m]), hipFuncAttributeMaxDynamicSharedMemorySize, dyn) != hipSuccess)
          B.fn_light[m] = nullptr, (void)hipGetLastError();
      }
      snprintf(name, sizeof name, "mpx_lightlows_%s_%d_%d", lm[m], B.phase, B.deg);  // one chunk per wavefront: small batches
      if (!c->lplan.low || hipModuleGetFunction(&B.fn_light_small[m], c->module, name) != hipSuccess) B.fn_light_small[m] = nullptr, (void)hipGetLastError();
    }
  }
  if (hipModuleGetFunction(&c->fn_gradl_fin, c->module, "mpx_gradl_finish") != hipSuccess) c->fn_gradl_fin = nullptr, (void)hipGetLastError();
  int rc;
  // (light_body reads the descriptor foreign[f_first + 0] of a group also when the group has NO low-degree node -- every lane fetches one
  // before it knows the count -- and takes a segment index from it for the width loads: a zero descriptor closes the list, so that a
  // group without low-degree nodes at the end of the list, or a single-degree grid with an empty list, reads segment 0 and not whatever
  // the allocation held: a memory access fault on one placement in some thousand, found by the round-6 soak of random grids)
  std::vector<MpxLightForeign> lforeign = c->lplan.foreign;
  lforeign.push_back(MpxLightForeign{});
  if (c->lplan.ok && ((rc = upload(c, &c->d_lgroups, c->lplan.groups)) || (rc = upload(c, &c->d_lforeign, lforeign)) || (rc = upload(c, &c->d_lftab, c->lplan.ftab)))) return rc;
  if ((rc = upload(c, &c->d_lt_ptr, c->lt_ptr)) || (rc = upload(c, &c->d_lt_col, c->lt_col)) || (rc = upload(c, &c->d_lt_row, c->lt_row)) ||
      (rc = upload(c, &c->d_lt_coef, c->lt_coef)))
    return rc;
  if (c->hess_by_node) {
    for (int p = 0; p < c->n_phases; ++p) {
      char name[64];
      snprintf(name, sizeof name, "mpx_node_hessn_%d", p);
      if (hipModuleGetFunction(&c->fn_hessn[p], c->module, name) != hipSuccess)
        return fail(c, MPX_ERR_INVALID, "code object lacks kernel %s (mixed-degree grid: node-ordered hess_l tiles)", name);
    }
    if ((rc = upload(c, &c->d_htiles, c->htiles)) || (rc = upload(c, &c->d_node_seg, c->node_seg)) || (rc = upload(c, &c->d_node_tk, c->node_tk))) return rc;
  }
  for (auto& t : c->degs) {
    if ((rc = upload(c, &t.d_D, t.D))) return rc;
    if ((rc = upload(c, &t.d_Cmid, t.Cmid))) return rc;
    if (t.deg > c->stream_above || (c->lplan.ok && c->lplan.high)) {  // streamed degrees and the high-degree light kernels: DT[j][k] = D[k][j], CT[j][k - 1] = C_mid[k - 1][j] (mpx_kernels.h: node_body TAB_GLB, light_high_body)
      const size_t n1 = (size_t)t.deg + 1, nm = (size_t)t.deg;
      std::vector<double> DT(n1 * n1), CT(n1 * nm);
      for (size_t k = 0; k < n1; ++k)
        for (size_t j = 0; j < n1; ++j) DT[j * n1 + k] = t.D[k * n1 + j];
      for (size_t k = 0; k < nm; ++k)
        for (size_t j = 0; j < n1; ++j) CT[j * nm + k] = t.Cmid[k * n1 + j];
      if ((rc = upload(c, &t.d_DT, DT)) || (rc = upload(c, &t.d_CT, CT))) return rc;
    }
    if ((rc = upload(c, &t.d_tk, t.tk))) return rc;
    if ((rc = upload(c, &t.d_Dmid, t.Dmid)) || (rc = upload(c, &t.d_tkm, t.tkm)) || (rc = upload(c, &t.d_w, t.w))) return rc;
  }
  for (auto& B : c->buckets) {
    if ((rc = upload(c, &B.d_node_i, B.node_i))) return rc;
    if ((rc = upload(c, &B.d_node_sk, B.node_sk))) return rc;
  }
  if ((rc = upload(c, &c->d_tiles, c->tiles))) return rc;
  if ((rc = upload(c, &c->d_Wnode, c->compW))) return rc;
  if ((rc = upload(c, &c->d_seg_start, c->seg_start))) return rc;
  {  // nlp_grad: the segments whose column-0 sums travel through `halo` (mpx_kernels.h: gradl_body halo_out) -- the first lane of the segment is
     // the first lane of its tile, or the lane before it belongs to another segment than s - 1 (mixed-degree grids)
    c->gl_halo_seg.clear();
    c->gl_halo_off.assign((size_t)c->n_phases + 1, 0);
    for (int p_ = 0; p_ < c->n_phases; ++p_) {
      std::vector<int32_t> segs;
      for (auto& B : c->buckets) {
        if (B.phase != p_) continue;
        for (int t = B.tile_first; t < B.tile_first + B.tile_count; ++t) {
          const MpxTile& T = c->tiles[(size_t)t];
          if (T.node0) continue;
          for (int j = 0; j < T.n; ++j) {
            const int sk = B.node_sk[(size_t)(T.m0 + j)], s_ = sk >> 8;
            if ((sk & 255) != 1 || s_ < 1) continue;
            if (!(j >= 1 && (B.node_sk[(size_t)(T.m0 + j - 1)] >> 8) == s_ - 1)) segs.push_back(s_);
          }
        }
      }
      std::sort(segs.begin(), segs.end());
      c->gl_halo_seg.insert(c->gl_halo_seg.end(), segs.begin(), segs.end());
      c->gl_halo_off[(size_t)p_ + 1] = (int32_t)c->gl_halo_seg.size();
    }
    if ((rc = upload(c, &c->d_gl_halo_seg, c->gl_halo_seg))) return rc;
  }
  if ((rc = upload(c, &c->d_lin_ptr, c->lin_ptr))) return rc;
  if ((rc = upload(c, &c->d_lin_idx, c->lin_idx))) return rc;
  if ((rc = upload(c, &c->d_lin_row, c->lin_row))) return rc;
  if ((rc = upload(c, &c->d_lin_coef, c->lin_coef))) return rc;
  if ((rc = upload(c, &c->d_mg_dst, c->mg_dst))) return rc;
  if ((rc = upload(c, &c->d_hc_dst, c->hc_dst))) return rc;
  if ((rc = upload(c, &c->d_th_dst, c->th_dst))) return rc;
  if ((rc = upload(c, &c->d_gmap, c->gmap)) || (rc = upload(c, &c->d_qmap, c->qmap))) return rc;
  if (c->absorb && ((rc = upload(c, &c->d_abs_fpos, c->abs_fpos)) || (rc = upload(c, &c->d_abs_fstage, c->abs_fstage)) || (rc = upload(c, &c->d_abs_fn, c->abs_fn)))) return rc;
  HIPCHK(c, hipEventCreate(&c->ev0));
  HIPCHK(c, hipEventCreate(&c->ev1));
  c->has_device = true;
  // resident kernel of single evaluations (generated for single-degree grids of low degree, codegen.py): one workgroup per tile, all
  // of them on the device at once (two fit a compute unit)
  {
    mpx_ctx::Resident& R = c->res;
    int n_cu = 256;
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->device);
    if (hipModuleGetFunction(&R.fn, c->module, "mpx_resident") != hipSuccess) R.fn = nullptr, (void)hipGetLastError();
    if (R.fn && !c->g_packed && (int64_t)c->tiles.size() <= n_cu) {
      std::vector<MpxNodeArgs> ba;
      std::vector<int32_t> tb(c->tiles.size(), 0);
      for (size_t k = 0; k < c->buckets.size(); ++k) {
        ba.push_back(node_args_static(c, c->buckets[k], false));
        for (int t = c->buckets[k].tile_first; t < c->buckets[k].tile_first + c->buckets[k].tile_count; ++t) tb[t] = (int32_t)k;
      }
      std::vector<MpxBoundArgs> bo(1, bound_args_static(c));
      std::vector<MpxResRequest> rq(MPX_RES_SLOTS);
      std::vector<unsigned long long> sc(1, 0);
      memset(rq.data(), 0, sizeof(MpxResRequest) * MPX_RES_SLOTS);
      if ((rc = upload(c, &R.d_buckets, ba)) || (rc = upload(c, &R.d_tile_bucket, tb)) || (rc = upload(c, &R.d_bound, bo)) || (rc = upload(c, &R.d_slots, rq)) ||
          (rc = upload(c, &R.d_sync, sc)) || (rc = upload(c, &R.d_seq, sc)))
        return rc;
      HIPCHK(c, hipHostMalloc((void**)&R.box, sizeof(MpxMailbox), hipHostMallocMapped));
      HIPCHK(c, hipHostGetDevicePointer((void**)&R.box_dev, R.box, 0));
      memset(R.box, 0, sizeof(MpxMailbox));
      HIPCHK(c, hipStreamCreateWithFlags(&R.stream, hipStreamNonBlocking));
      R.ok = true;
    }
  }
  return MPX_OK;
}

// (wavefront / workgroup scans of the widths: mpx_scan.h, shared with mpx_equal_area.cpp)
__global__ __launch_bounds__(MPX_PREFIX_THREADS) void mpx_prefix_kernel(const double* __restrict__ w, double* __restrict__ wcum, int S) {
  __shared__ double wave_tot[MPX_PREFIX_THREADS / 64];
  const double* __restrict__ a = w + (int64_t)blockIdx.x * S;
  prefix_scan_block([&](int s) { return a[s]; }, wcum + (int64_t)blockIdx.x * S, S, threadIdx.x, wave_tot);
}

// exclusive prefix sums of n_w width vectors at device address p into c->wcum (all phases), with the bookkeeping MPX_WIDTHS_UNCHANGED relies on
int launch_prefix(mpx_ctx* c, const double* p, int64_t n_w, int p_per_point) {
  hipLaunchKernelGGL(mpx_prefix_kernel, dim3((unsigned)(n_w * c->n_phases)), dim3(MPX_PREFIX_THREADS), 0, c->stream, p, c->wcum.p, c->S);
  HIPCHK(c, hipGetLastError());
  c->wcum_p = p, c->wcum_batch = n_w, c->wcum_ppp = p_per_point ? 1 : 0, c->wcum_phases = all_phases(c);
  return MPX_OK;
}

int launch(mpx_ctx* c, hipFunction_t fn, dim3 grid, dim3 block, void* args, size_t size, unsigned lds_bytes = 0, bool any_order = false) {
  void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  if (any_order) {  // (experiment, MPX_BOUNDARY_ANYORDER: no barrier against the previous kernel of the stream; sizes in work-items)
    HIPCHK(c, hipExtModuleLaunchKernel(fn, grid.x * block.x, grid.y * block.y, grid.z * block.z, block.x, block.y, block.z, lds_bytes, c->stream, nullptr, cfg, nullptr, nullptr,
                                       1u /* hipExtAnyOrderLaunch */));
    return MPX_OK;
  }
  HIPCHK(c, hipModuleLaunchKernel(fn, grid.x, grid.y, grid.z, block.x, block.y, block.z, lds_bytes, c->stream, nullptr, cfg));
  return MPX_OK;
}

// Move the packed g / grad_f values of the node kernels to their rows: lane <-> output entry, so the stores are fully
// coalesced; entries written by the boundary kernel (map < 0) are left alone.  A lane keeps its map entry for MPX_UNPACK_PTS
// evaluation points (one index load, that many independent value loads in flight).
#define MPX_UNPACK_PTS 8
__global__ __launch_bounds__(256) void mpx_unpack_kernel(const double* __restrict__ tmp, int64_t tmp_stride, double* __restrict__ g, int64_t g_stride,
                                                         const int64_t* __restrict__ gmap, int64_t n_g, double* __restrict__ grad,
                                                         int64_t grad_stride, const int64_t* __restrict__ qmap, int64_t n_z, int B) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int b0 = blockIdx.y * MPX_UNPACK_PTS, nb = min(MPX_UNPACK_PTS, B - b0);
  double* __restrict__ out;
  int64_t m, stride;
  if (r < n_g) {
    if (!g) return;
    m = gmap[r], out = g + r, stride = g_stride;
  } else {
    r -= n_g;
    if (r >= n_z || !grad) return;
    m = qmap[r], out = grad + r, stride = grad_stride;
  }
  if (m < 0) return;
  const double* __restrict__ src = tmp + (int64_t)b0 * tmp_stride + m;
  out += (int64_t)b0 * stride;
  double v[MPX_UNPACK_PTS];
#pragma unroll
  for (int k = 0; k < MPX_UNPACK_PTS; ++k) v[k] = k < nb ? src[(int64_t)k * tmp_stride] : 0.0;
#pragma unroll
  for (int k = 0; k < MPX_UNPACK_PTS; ++k)
    if (k < nb) out[(int64_t)k * stride] = v[k];
}


// Evaluation points per workgroup.
//  * Round 1 picked 4-8 for large batches (best case of the software-pipelined loop).  Round 2 measured both over physical
//    placements of the output buffers on five boxes (tools/placement_ab.py, profiles/r2_headline): with 5 points per workgroup the
//    headline kernel ranges 878 ... 1167 us, with ONE point 916 ... 1016 us -- every XCD then advances through one contiguous
//    window of the outputs (~9 points deep) instead of 45 points at once, which the HBM controllers serve evenly wherever the
//    pages lie (slow placements -13 %, the fastest +4 %); config 3: +15 % on a slow-state box.
//  * Which of the two wins is a property of where the driver put THESE buffers, so for large batches the library measures it:
//    after two unmeasured ones the next four passes that write a given output array run the two geometries (order A B B A) between HIP events on the context's
//    stream (no synchronisation: the timings are read with hipEventQuery once they exist), after which the faster one is used
//    for that array.  Results do not depend on the geometry (fixed-order reductions; tests/test_gpu_parity.py).
//  * The Hessian kernels take ONE point per workgroup as a compile-time fact (round 3: half the registers without the batch loop's
//    state, mpx_kernels.h); small batches 1.  MPX_BPB overrides (first-order passes), MPX_NO_TUNE=1 pins 1.
struct GeomPick {
  int bpb;
  hipEvent_t begin, end;  // non-null: bracket the node launches of this pass
};

GeomPick pick_geometry(mpx_ctx* c, int64_t B, int mode, const void* key, int sig = 0) {
  const bool light = !(sig & 8);
  const char* env = mpx_knob(MPX_K_BPB);  // tuning / test override
  if (env && atoi(env) > 0 && mode != MPX_MODE_HESS) return {atoi(env), nullptr, nullptr};
  // (the hess_l node kernels take one evaluation point per workgroup as a compile-time fact, mpx_kernels.h: MPX_HESS_ONE_POINT --
  // a host-side override could only leave points unevaluated, so there is none; a code object built with -DMPX_HESS_ONE_POINT=0
  // runs its batch loop over one point per workgroup as well)
  if (mode == MPX_MODE_HESS) return {1, nullptr, nullptr};
  const int64_t work = B * (c->tile_end - c->tile_begin);
  static const bool no_tune = getenv("MPX_NO_TUNE") != nullptr;
  if (no_tune || work < 8192 || !key || c->shard_world > 1) return {1, nullptr, nullptr};
  {  // no event records / queries inside a stream capture (the caller is building a hipGraph): the robust geometry, no measuring
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(c->stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
      (void)hipGetLastError();
      return {1, nullptr, nullptr};
    }
  }
  mpx_ctx::GeomTune* T = nullptr;
  for (auto& t : c->tune)
    if (t.key == key && t.B == B && t.mode == mode && t.sig == sig) T = &t;
  if (!T) {
    if (c->tune.size() < 8) {
      c->tune.emplace_back();
      T = &c->tune.back();
    } else {  // recycle the least recently used entry whose measurement is not in flight (its events are reused)
      for (auto& t : c->tune) {
        const bool idle = t.stage == 0 || t.stage == 5 || (t.ev[7] && hipEventQuery(t.ev[7]) == hipSuccess);
        if (idle && (!T || t.last_use < T->last_use)) T = &t;
      }
      (void)hipGetLastError();  // hipErrorNotReady of the queries
      if (!T) return {1, nullptr, nullptr};  // every entry is mid-measurement: this pass is not tuned
    }
    T->key = key, T->B = B, T->mode = mode, T->sig = sig, T->stage = -2, T->best = 1, T->uses = 0;
    T->cand[0] = 1;
    // (passes without the Jacobian values write little: they are bound by the lifetime of a workgroup, not by HBM -- f alone 272 ->
    // 156 us, grad_f alone 456 -> 271 us at 16 points per workgroup, config 2, B = 4096; tools/r3_single_oracle_bpb.py)
    T->cand[1] = light ? (int)std::min<int64_t>(std::max<int64_t>(work / 2048, 2), 16) : (int)std::min<int64_t>(std::max<int64_t>(work / 16384, 2), 8);
    for (auto& e : T->ev)
      if (!e && hipEventCreate(&e) != hipSuccess) return {1, nullptr, nullptr};
  }
  T->last_use = ++c->tune_clock;
  if (T->stage == 5 && ++T->uses >= 512) T->stage = 0, T->uses = 0;  // placements can change behind the same address: look again now and then
  if (T->stage < 0) {  // two unmeasured passes first, one per geometry: the first kernels to touch freshly allocated arrays run slow
    const int k = T->stage++;  // (measured from the first pass on, the first candidate paid for that and lost: config 4, +8 % left behind)
    return {T->cand[k == -2 ? 0 : 1], nullptr, nullptr};
  }
  if (T->stage < 4) {  // measurement passes in the order 0, 1, 1, 0: a clock that is still ramping up cancels out of the sums
    const int k = T->stage++;
    return {T->cand[(k == 1 || k == 2) ? 1 : 0], T->ev[2 * k], T->ev[2 * k + 1]};
  }
  if (T->stage == 4) {
    // not there yet (the host is running ahead of the device): the prior -- light passes want several points per workgroup, and so
    // do passes whose workgroups write little per point (low degrees: 14 KB of Jacobian values per tile at 4000 x 3 against 46 KB at
    // 1000 x 5 -- the workgroup's prologue then weighs more than the placement effect one point per workgroup is robust against)
    if (hipEventQuery(T->ev[7]) != hipSuccess) {
      const bool small_tiles = c->nnz_j * 8 < (int64_t)24576 * (int64_t)std::max<size_t>(c->tiles.size(), 1);
      return {T->cand[(light || small_tiles) ? 1 : 0], nullptr, nullptr};
    }
    float t[4] = {0, 0, 0, 0};
    for (int k = 0; k < 4; ++k)
      if (hipEventElapsedTime(&t[k], T->ev[2 * k], T->ev[2 * k + 1]) != hipSuccess) t[k] = 1e30f;
    T->best = (t[1] + t[2]) < 0.98f * (t[0] + t[3]) ? T->cand[1] : T->cand[0];  // the robust geometry unless the other wins clearly
    T->stage = 5;
  }
  return {T->best, nullptr, nullptr};
}

// Arguments of a bucket's node kernels that do not depend on the call (io, the tile sub-range and the absorbing lists come on top).
MpxNodeArgs node_args_static(const mpx_ctx* c, const Bucket& B, bool absorber) {
  const PhaseStruct& P = c->ph[B.phase];
  const DegTable& t = c->degs[B.dt];
  MpxNodeArgs A{};
  A.tiles = c->d_tiles;
  A.node_i = B.d_node_i;
  A.node_sk = B.d_node_sk;
  A.Dmat = t.deg > c->stream_above ? t.d_DT : t.d_D;  // (streamed degrees: the transposed tables)
  A.Cmid = t.deg > c->stream_above ? t.d_CT : t.d_Cmid;
  A.tk = t.d_tk;
  A.Dmid = t.d_Dmid;
  A.tkm = t.d_tkm;
  A.phase = B.phase;
  A.deg = B.deg;
  A.Wnode = c->d_Wnode;
  A.inv_dtau = 1.0 / (c->tau1 - c->tau0);
  A.z_off = P.z_off;
  A.g_off_F = P.g_off_F;
  A.g_off_C = P.g_off_C;
  A.g_off_DU = P.g_off_DU;
  A.g_off_mU = P.g_off_mU;
  A.N = (int32_t)c->N;
  A.seg_off = B.phase * c->S;
  A.tile_first = B.tile_first;
  A.tile_count = B.tile_count;
  // regular bucket (every segment of the phase has this degree, tiles = [node 0][full ...][last]): descriptors by arithmetic
  {
    const int nt = B.tile_count;
    bool reg = (int64_t)B.node_i.size() == c->N && nt >= 2 && c->tiles[B.tile_first].node0 && !absorber;
    const int lanes = reg ? c->tiles[B.tile_first + 1].n : 0;
    for (int t = 1; reg && t < nt; ++t) {
      const MpxTile& T = c->tiles[B.tile_first + t];
      const MpxTile& F = c->tiles[B.tile_first + 1];
      reg = T.m0 == 1 + (t - 1) * lanes && !T.node0 && T.n == T.n_own && (t == nt - 1 ? T.n <= lanes : T.n == lanes) && lanes % B.deg == 0;
      if (reg && t < nt - 1 && t > 1)  // full tiles: equal blocks at equal strides
        reg = T.jac_base - F.jac_base == (int64_t)(t - 1) * c->tile_jac_size[B.tile_first + 1] &&
              T.hess_base - F.hess_base == (int64_t)(t - 1) * c->tile_hess_size[B.tile_first + 1] &&
              T.g_base - F.g_base == (int64_t)(t - 1) * c->tile_g_size[B.tile_first + 1];
    }
    A.regular = reg ? 1 : 0;
    if (reg) {
      const int idx[3] = {B.tile_first, B.tile_first + 1, B.tile_first + nt - 1};
      for (int w = 0; w < 3; ++w) {
        A.reg_jac_base[w] = c->tiles[idx[w]].jac_base, A.reg_hess_base[w] = c->tiles[idx[w]].hess_base, A.reg_g_base[w] = c->tiles[idx[w]].g_base;
      }
      A.reg_first_tile = B.tile_first, A.reg_last = nt - 1, A.reg_lanes = lanes, A.reg_last_lanes = c->tiles[idx[2]].n;
      A.reg_jac_size = c->tile_jac_size[B.tile_first + 1], A.reg_hess_size = c->tile_hess_size[B.tile_first + 1], A.reg_g_size = c->tile_g_size[B.tile_first + 1];
    }
  }
  return A;
}

// ... and of the boundary kernels
MpxBoundArgs bound_args_static(const mpx_ctx* c) {
  MpxBoundArgs G{};
  for (int p = 0; p < c->n_phases; ++p) {
    const PhaseStruct& P = c->ph[p];
    G.ph[p].z_off = P.z_off;
    G.ph[p].N = (int32_t)c->N;
    G.ph[p].tile_first = P.tile_first;
    G.ph[p].tile_count = P.tile_count;
    G.ph[p].tile_count_h = c->hess_by_node ? c->ph_htile_count[p] : P.tile_count;
    G.ph[p].g_off_TC = P.g_off_TC;
    G.ph[p].jac_TC = P.jac_TC;
    G.mg_off[p] = c->mg_off[p];
    G.hc_off[p] = c->hc_off[p];
    G.th_off[p] = c->th_off[p];
  }
  G.mg_dst = c->d_mg_dst;
  G.hc_dst = c->d_hc_dst;
  G.th_dst = c->d_th_dst;
  G.lin_ptr = c->d_lin_ptr;
  G.lin_idx = c->d_lin_idx;
  G.lin_coef = c->d_lin_coef;
  G.lin_row = c->d_lin_row;
  G.lin_jac = c->lin_jac;
  G.n_lin = (int32_t)c->lin_row.size();
  return G;
}

int run_mode(mpx_ctx* c, int mode, const MpxIO& io0, bool nodes = true, bool owner = false) {
  MpxIO io = io0;
  // (the array that identifies the pass for the geometry measurement: its largest output)
  const void* geom_key = mode == MPX_MODE_HESS ? (const void*)io.hess : io.jac ? (const void*)io.jac : io.g ? (const void*)io.g : io.grad ? (const void*)io.grad : (const void*)io.f;
  const GeomPick geom = pick_geometry(c, io.B, mode, geom_key, (io.f ? 1 : 0) | (io.g ? 2 : 0) | (io.grad ? 4 : 0) | (io.jac ? 8 : 0));
  io.b_per_block = geom.bpb;
  // (a workgroup row per chunk of evaluation points: at most 65535 rows per launch -- the first-order passes take more points per
  // workgroup past that, the hess_l passes (one point per workgroup, compile-time) are launched in slices, MpxIO::b_first)
  if (mode != MPX_MODE_HESS)
    while ((io.B + io.b_per_block - 1) / io.b_per_block > 65535) ++io.b_per_block;
  // Packed staging of g / grad_f in tile order: (a) mixed-degree phases, full evaluations (a plain mpx_set_tile_range keeps
  // the direct stores); (b) every segment-sharded evaluation (mpx_shard_setup): a rank's tiles are one contiguous run of the
  // staging block, which is what the ranks exchange; the boundary pass then moves the assembled block to g / grad_f.
  const bool shard = c->shard_world > 1;
  const bool want_g = mode != MPX_MODE_HESS && (io.g || io.grad);
  if (shard && !owner && want_g && io.B > 65535) return fail(c, MPX_ERR_UNSUPPORTED, "segment-sharded evaluation: batch must be <= 65535");
  // (owner-resident sharding, MPX_OWNER_RESIDENT: nothing but the tile partials is exchanged, so the node kernels store their
  // g / grad_f rows directly, like a plain tile sub-range)
  // Light passes (no Jacobian values) of high-degree buckets run on the matrix cores (light_body) with direct stores; the other
  // buckets of such a pass store directly too (no staging block, no row spans).  Any batch size: which kernel evaluates a pass
  // never depends on the batch, so results do not either.  MPX_NO_LIGHT=1: the node kernels (A/B runs).
  // (and only when the boundary pass follows in THIS call: the light kernels write their partial sums in their own slot layout --
  // one slot per group / span or [phase][64-node chunk] --, which a later MPX_BOUNDARY_ONLY call and mpx_get_partials, both laid
  // out per tile, would misread; mpx_set_tile_range(0, n_tiles, run_boundary = 0) therefore keeps the node kernels)
  bool light = c->lplan.ok && mode != MPX_MODE_HESS && !io.jac && !shard && nodes && c->run_boundary && c->tile_begin == 0 &&
               c->tile_end == (int64_t)c->tiles.size() && !mpx_knob(MPX_K_NO_LIGHT);
  for (auto& B : c->buckets)
    if (light && B.deg == c->lplan.deg && (!B.fn_light[mode == MPX_MODE_FGJ ? 1 : 0] || (c->lplan.low && !B.fn_light_small[mode == MPX_MODE_FGJ ? 1 : 0]))) light = false;
  if (light && (c->lplan.low || c->lplan.high)) io.n_tiles_total = c->n_phases * c->lplan.n_low_chunks;  // (partial-sum slots of a light pass: [phase][64-node chunk] / [phase][segment])
  // low-degree plan: fewer long spans than a wavefront per SIMD of the device -> one 64-node chunk per wavefront (same sums)
  const bool light_small = light && c->lplan.low && (int64_t)c->lplan.n_low_groups * io.B < 1024 && !mpx_knob(MPX_K_LIGHT_LONG_SPANS);
  const bool packed = want_g && io.B <= 65535 && !light &&
                      ((shard && !owner) || (c->g_packed && nodes && c->tile_begin == 0 && c->tile_end == (int64_t)c->tiles.size() && !mpx_knob(MPX_K_NO_PACKED_G)));
  if (packed) {
    int rc = reserve(c, c->gtmp, (size_t)(io.B * c->gtmp_n));
    if (rc) return rc;
    io.gtmp = c->gtmp.p;
    io.gtmp_stride = c->gtmp_n;
  }
  const int gy = (io.B + io.b_per_block - 1) / io.b_per_block;
  hipEvent_t pe1 = nullptr;
  {
    int rc = prof_begin(c, &pe1);
    if (rc) return rc;
  }
  if (geom.begin && nodes) HIPCHK(c, hipEventRecord(geom.begin, c->stream));
  // absorbing buckets (see build_layout) go last: their tiles read what the other buckets staged
  const bool absorb = packed && !shard && c->absorb;
  const bool by_node = mode == MPX_MODE_HESS && c->hess_by_node;
  if (light) {  // every phase: one launch of persistent wavefronts over the (group, evaluation point) items (light_body)
    // (every group / span writes its partial-sum slot, phase tile_first + index; the boundary pass of a light evaluation sums
    // exactly those -- see below --, so the other slots need no clearing)
    static int n_cu = 0;
    if (!n_cu && hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess) n_cu = 256;
    // one launch for all phases where the code object has the kernels (single-degree grid, n_phases > 1); MPX_NO_PHASE_MERGE=1: A/B
    const int lmode = mode == MPX_MODE_FGJ ? 1 : 0;
    const bool merged_light = c->lplan.low && c->n_phases > 1 && (light_small ? c->fn_lightlows_all : c->fn_lightlow_all)[lmode] != nullptr &&
                              (int)c->buckets.size() == c->n_phases && !mpx_knob(MPX_K_NO_PHASE_MERGE);
    MpxLightMultiArgs LM{};
    for (auto& B : c->buckets) {
      if (B.deg != c->lplan.deg) continue;
      const PhaseStruct& P = c->ph[B.phase];
      const DegTable& t = c->degs[B.dt];
      MpxLightArgs L{};
      MpxNodeArgs& A = L.node;
      A.io = io, A.tiles = c->d_tiles, A.node_i = B.d_node_i, A.node_sk = B.d_node_sk;
      A.Dmat = t.d_D, A.Cmid = t.d_Cmid, A.tk = t.d_tk, A.Dmid = t.d_Dmid, A.tkm = t.d_tkm, A.phase = B.phase, A.Wnode = c->d_Wnode;
      A.inv_dtau = 1.0 / (c->tau1 - c->tau0);
      A.z_off = P.z_off, A.g_off_F = P.g_off_F, A.g_off_C = P.g_off_C, A.g_off_DU = P.g_off_DU, A.g_off_mU = P.g_off_mU;
      A.N = (int32_t)c->N, A.seg_off = B.phase * c->S;
      L.groups = c->d_lgroups, L.foreign = c->d_lforeign, L.wdeg = t.d_w;
      for (size_t k = 0; k < c->degs.size() && !c->lplan.low && !c->lplan.high; ++k) L.fD_off[k] = c->lplan.fD_off[k], L.fC_off[k] = c->lplan.fC_off[k], L.fdeg[k] = c->degs[k].deg;
      L.ftab = c->d_lftab, L.ftab_n = (int32_t)c->lplan.ftab.size();
      L.n_groups = c->lplan.low ? c->lplan.n_low_groups : (int32_t)c->lplan.groups.size(), L.first_node = c->lplan.first_node, L.span_cap = c->lplan.span_cap, L.slot_first = P.tile_first;
      // low-degree plan: fewer long spans than half the wavefront slots of the device -> one 64-node chunk per wavefront (same sums:
      // the partial-sum slots are per chunk for both span lengths)
      const bool small = light_small;
      if (c->lplan.low || c->lplan.high) L.slot_first = B.phase * c->lplan.n_low_chunks;
      if (small) L.n_groups = c->lplan.n_low_chunks;
      if (c->lplan.high) {  // light_high_body: workgroup = (segment, 16 evaluation points); the TRANSPOSED tables as matrix operands
        A.Dmat = t.d_DT, A.Cmid = t.d_CT;
        const unsigned lds = (unsigned)((int64_t)(c->nx + c->nu) * c->lplan.span_cap * 17 * 8);
        const int64_t nblk = (io.B + 15) / 16;
        for (int64_t y0 = 0; y0 < nblk; y0 += 65535) {
          A.io.b_first = (int32_t)(16 * y0);
          int rc = launch(c, B.fn_light[mode == MPX_MODE_FGJ ? 1 : 0], dim3((unsigned)c->S, (unsigned)std::min<int64_t>(65535, nblk - y0), 1), dim3(256, 1, 1), &L, sizeof L, lds);
          if (rc) return rc;
          if (c->profile) ++c->prof_launches;
        }
        continue;
      }
      static long long* ldbg = nullptr;
      if (!ldbg && mpx_knob(MPX_K_LIGHT_DEBUG)) HIPCHK(c, hipHostMalloc((void**)&ldbg, 128, hipHostMallocMapped));
      L.dbg = ldbg;
      int per_cu = 2;  // resident workgroups per compute unit (the kernels' launch bounds)
      if (const char* e = mpx_knob(MPX_K_LIGHT_PER_CU)) per_cu = std::max(1, atoi(e));
      if (merged_light) {  // collect the phases; the launch follows the last one
        if (B.phase == 0) LM.base = L;
        LM.ph[B.phase] = MpxLightPhase{A.z_off, A.g_off_F, A.g_off_C, A.g_off_DU, A.g_off_mU, A.seg_off, L.slot_first};
        if (++LM.n_ph < c->n_phases) continue;
        const int64_t items = (int64_t)L.n_groups * c->n_phases * io.B;
        const unsigned wgs = (unsigned)std::min<int64_t>((items + MPX_LIGHT_WAVES - 1) / MPX_LIGHT_WAVES, per_cu * (int64_t)n_cu);
        int rc = launch(c, (small ? c->fn_lightlows_all : c->fn_lightlow_all)[lmode], dim3(wgs, 1, 1), dim3(64 * MPX_LIGHT_WAVES, 1, 1), &LM,
                        offsetof(MpxLightMultiArgs, ph) + (size_t)c->n_phases * sizeof(MpxLightPhase));
        if (rc) return rc;
        if (c->profile) ++c->prof_launches;
        continue;
      }
      const int64_t items = (int64_t)L.n_groups * io.B;
      const unsigned wgs = (unsigned)std::min<int64_t>((items + MPX_LIGHT_WAVES - 1) / MPX_LIGHT_WAVES, per_cu * (int64_t)n_cu);
      const unsigned lds = c->lplan.low ? 0u : (unsigned)((MPX_LIGHT_WAVES * (c->nx + c->nu) * c->lplan.span_cap + c->lplan.ftab.size()) * 8);
      int rc = launch(c, (small ? B.fn_light_small : B.fn_light)[mode == MPX_MODE_FGJ ? 1 : 0], dim3(wgs, 1, 1), dim3(64 * MPX_LIGHT_WAVES, 1, 1), &L, sizeof L, lds);
      if (rc) return rc;
      if (c->profile) ++c->prof_launches;
      if (ldbg) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        fprintf(stderr, "light kernel (deg %d) phases of one item (us): span loads %.2f  matrix core + node functions %.2f  outputs %.2f\n", B.deg, (ldbg[1] - ldbg[0]) / 100.0,
                (ldbg[2] - ldbg[1]) / 100.0, (ldbg[3] - ldbg[2]) / 100.0);
      }
    }
  }
  if (by_node && nodes && !light) {  // mixed-degree grid: one launch per phase over node-ordered tiles
    for (int p = 0; p < c->n_phases; ++p) {
      int64_t lo = c->ph_htile_first[p], hi = lo + c->ph_htile_count[p];
      if (shard) {  // the rank's share of every phase's tiles (same fractions in every phase)
        const int64_t cnt = c->ph_htile_count[p];
        const int64_t a0 = c->shard_cuts_h[c->shard_rank] * cnt / c->shard_cuts_h.back(), a1 = c->shard_cuts_h[c->shard_rank + 1] * cnt / c->shard_cuts_h.back();
        hi = lo + a1, lo = lo + a0;
      } else if (c->tile_begin != 0 || c->tile_end != (int64_t)c->tiles.size()) {
        // mpx_set_tile_range: the range [b, e) of the context's tiles maps proportionally onto every phase's node-ordered tiles
        // (ranges that partition the tiles partition these too)
        const int64_t cnt = c->ph_htile_count[p], nt = (int64_t)c->tiles.size();
        hi = lo + c->tile_end * cnt / nt, lo = lo + c->tile_begin * cnt / nt;
      }
      if (hi <= lo) continue;
      const PhaseStruct& P = c->ph[p];
      MpxHessNodeArgs A{};
      A.io = io;
      A.htiles = c->d_htiles, A.node_seg = c->d_node_seg, A.node_tk = c->d_node_tk, A.Wnode = c->d_Wnode;
      A.inv_dtau = 1.0 / (c->tau1 - c->tau0);
      A.z_off = P.z_off, A.g_off_F = P.g_off_F, A.g_off_C = P.g_off_C;
      A.N = (int32_t)c->N, A.seg_off = p * c->S, A.tile_first = (int32_t)lo, A.tile_count = (int32_t)(hi - lo);
      for (int64_t bf = 0; bf < (int64_t)gy * io.b_per_block; bf += (int64_t)65535 * io.b_per_block) {
        A.io.b_first = (int32_t)bf;
        const int gys = (int)std::min<int64_t>(65535, gy - bf / io.b_per_block);
        int rc = launch(c, c->fn_hessn[p], dim3((unsigned)(hi - lo), gys, 1), dim3(MPX_TILE, 1, 1), &A, sizeof A);
        if (rc) return rc;
        if (c->profile) ++c->prof_launches;
      }
    }
  }
  // All phases of a single-degree grid in ONE launch (mpx_kernels.h: node_all): the x dimension of the grid holds the tiles of the
  // phases' ranges one after the other.  MPX_NO_PHASE_MERGE=1 (read per call): one launch per phase, the same bits (tested).
  bool merged_nodes = false;
  if (!by_node && !light && nodes && c->fn_node_all[mode] && c->n_phases > 1 && (int)c->buckets.size() == c->n_phases && !absorb &&
      !mpx_knob(MPX_K_NO_PHASE_MERGE)) {
    MpxNodeMultiArgs M{};
    int64_t total = 0;
    for (auto& B : c->buckets) {
      const int64_t lo = std::max<int64_t>(B.tile_first, c->tile_begin), hi = std::min<int64_t>(B.tile_first + B.tile_count, c->tile_end);
      MpxNodeArgs& A = M.a[B.phase];
      A = node_args_static(c, B, false);
      A.io = io;
      A.tile_first = (int32_t)std::min(lo, hi);
      A.tile_count = (int32_t)std::max<int64_t>(hi - lo, 0);
    }
    bool in_order = true;  // (buckets in phase order: tile_cum below relies on it)
    for (int p = 0; p < c->n_phases; ++p) in_order = in_order && c->buckets[p].phase == p;
    for (int p = 0; p < c->n_phases; ++p) M.tile_cum[p] = (int32_t)total, total += M.a[p].tile_count;
    for (int p = c->n_phases; p <= MPX_MAX_PHASES; ++p) M.tile_cum[p] = (int32_t)total;
    M.n_ph = c->n_phases;
    if (in_order && total > 0) {
      merged_nodes = true;
      for (int64_t bf = 0; bf < (int64_t)gy * io.b_per_block; bf += (int64_t)65535 * io.b_per_block) {
        for (int p = 0; p < c->n_phases; ++p) M.a[p].io.b_first = (int32_t)bf;
        const int gys = (int)std::min<int64_t>(65535, gy - bf / io.b_per_block);
        int rc = launch(c, c->fn_node_all[mode], dim3((unsigned)total, gys, 1), dim3(MPX_TILE, 1, 1), &M, offsetof(MpxNodeMultiArgs, a) + (size_t)c->n_phases * sizeof(MpxNodeArgs));
        if (rc) return rc;
        if (c->profile) ++c->prof_launches;
      }
    } else if (in_order) {
      merged_nodes = true;  // nothing to run in this tile range
    }
  }
  for (int pass = 0; pass < 2 && !by_node && !light && !merged_nodes; ++pass)
  for (auto& B : c->buckets) {
    const bool absorber = absorb && B.abs_cap > 0;
    int64_t lo = std::max<int64_t>(B.tile_first, c->tile_begin), hi = std::min<int64_t>(B.tile_first + B.tile_count, c->tile_end);
    if (absorber) {  // its node-0 mini tile (if any) stages like the other buckets: first pass, the absorbing tiles in the second
      const int64_t split = B.tile_first + (c->tiles[B.tile_first].node0 ? 1 : 0);
      if (pass == 0) hi = std::min(hi, split); else lo = std::max(lo, split);
    } else if (pass == 1) {
      continue;
    }
    if (hi <= lo || !nodes) continue;
    const PhaseStruct& P = c->ph[B.phase];
    const DegTable& t = c->degs[B.dt];
    MpxNodeArgs A = node_args_static(c, B, absorber);
    A.io = io;
    A.tile_first = (int32_t)lo;
    A.tile_count = (int32_t)(hi - lo);
    unsigned lds = 0;
    if (absorber && pass == 1) {
      A.abs_fpos = c->d_abs_fpos, A.abs_fstage = c->d_abs_fstage, A.abs_fn = c->d_abs_fn, A.abs_cap = B.abs_cap;
      lds = (unsigned)B.abs_cap * (unsigned)B.abs_slots * 8u;
    }
    for (int64_t bf = 0; bf < (int64_t)gy * io.b_per_block; bf += (int64_t)65535 * io.b_per_block) {
      A.io.b_first = (int32_t)bf;
      const int gys = (int)std::min<int64_t>(65535, gy - bf / io.b_per_block);
      int rc = launch(c, B.fn[mode], dim3((unsigned)(hi - lo), gys, 1), dim3(MPX_TILE, 1, 1), &A, sizeof A, lds);
      if (rc) return rc;
      if (c->profile) ++c->prof_launches;
    }
  }
  if (packed && (!shard || !nodes) && !absorb) {
    const int64_t rows = c->n_g + c->n_z;
    hipLaunchKernelGGL(mpx_unpack_kernel, dim3((unsigned)((rows + 255) / 256), (unsigned)((io.B + MPX_UNPACK_PTS - 1) / MPX_UNPACK_PTS)), dim3(256), 0,
                       c->stream, c->gtmp.p, c->gtmp_n, io.g, io.g_stride, c->d_gmap, c->n_g, io.grad, io.grad_stride, c->d_qmap, c->n_z, (int)io.B);
    HIPCHK(c, hipGetLastError());
  }
  if (geom.end && nodes) HIPCHK(c, hipEventRecord(geom.end, c->stream));
  {
    int rc = prof_end(c, pe1);
    if (rc) return rc;
  }
  if (shard ? nodes : !c->run_boundary) return MPX_OK;  // sharded: node pass and boundary pass are separate calls
  MpxBoundArgs G = bound_args_static(c);
  G.io = io;
  G.part_group = light_small ? c->lplan.own / 64 : 1;
  if (light)  // the slots the light kernels wrote: one per group in front of the phase's tile slots, or [phase][64-node chunk]
    for (int p = 0; p < c->n_phases; ++p) {
      if (c->lplan.high) G.ph[p].tile_first = p * c->lplan.n_low_chunks, G.ph[p].tile_count = c->lplan.n_low_chunks;
      else if (c->lplan.low) G.ph[p].tile_first = p * c->lplan.n_low_chunks, G.ph[p].tile_count = light_small ? c->lplan.n_low_chunks : c->lplan.n_low_groups;
      else G.ph[p].tile_count = (int32_t)c->lplan.groups.size();
    }
  // (MPX_BOUNDARY_ANYORDER=1, read per call, MPX_BOUNDARY_ONLY calls only: an experiment of round 5 -- the boundary pass of the config-5
  // loop launched without a barrier against the equal-area kernel in front of it, profiles/r5_loop/)
  // Compiled out of production builds: without the barrier the pass reads f and the sums while the kernel in front of it
  // may still be writing them -- only for the measurement it was written for (MPX_LIB_HIPCC_FLAGS=-DMPX_EXPERIMENTS).
#ifdef MPX_EXPERIMENTS
  const bool any_order = !nodes && getenv("MPX_BOUNDARY_ANYORDER") != nullptr;
#else
  const bool any_order = false;
#endif
  return launch(c, c->fn_bound[mode], dim3((unsigned)io.B, 1, 1), dim3(256, 1, 1), &G, sizeof G, 0, any_order);
}

}  // namespace

// ------------------------------------------------------------------------------------------------
extern "C" const char* mpx_last_error(const mpx_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

extern "C" int mpx_create(const mpx_problem* prob, mpx_ctx** out) {
  if (!prob || !out) return fail(nullptr, MPX_ERR_INVALID, "null argument");
  *out = nullptr;
  if (prob->version != MPX_VERSION) return fail(nullptr, MPX_ERR_INVALID, "mpx_problem.version %d != %d", prob->version, MPX_VERSION);
  if (prob->n_phases < 1 || prob->n_phases > MPX_MAX_PHASES) return fail(nullptr, MPX_ERR_UNSUPPORTED, "n_phases must be 1..%d", MPX_MAX_PHASES);
  if (prob->nx < 1 || prob->nu < 0 || prob->na < 0 || prob->n_segments < 1 || !prob->poly_orders)
    return fail(nullptr, MPX_ERR_INVALID, "bad dimensions");
  if (prob->n_segments >= (1 << 23)) return fail(nullptr, MPX_ERR_UNSUPPORTED, "too many segments");
  if (!(prob->tau1 > prob->tau0)) return fail(nullptr, MPX_ERR_INVALID, "tau1 must exceed tau0");
  mpx_ctx* c = new (std::nothrow) mpx_ctx;
  if (!c) return fail(nullptr, MPX_ERR_ALLOC, "out of memory");
  c->n_phases = prob->n_phases;
  c->nx = prob->nx;
  c->nu = prob->nu;
  c->na = prob->na;
  c->S = prob->n_segments;
  c->scheme = prob->scheme;
  c->tau0 = prob->tau0;
  c->tau1 = prob->tau1;
  c->device = prob->device;
  c->orders.assign(prob->poly_orders, prob->poly_orders + prob->n_segments);
  if (const char* e = getenv("MPX_TABLES_STREAM_ABOVE")) c->stream_above = std::max(12, std::min(255, atoi(e)));  // (A/B and the bit-identity tests: with the matching code object)
  if (prob->n_links > 0 && prob->links) c->links.assign(prob->links, prob->links + 2 * prob->n_links);
  for (size_t l = 0; l < c->links.size(); ++l)
    if (c->links[l] < 0 || c->links[l] >= c->n_phases) {
      g_create_error = "phase link out of range";
      delete c;
      return MPX_ERR_INVALID;
    }
  c->ph.resize(c->n_phases);
  int rc = parse_structure(c, prob->structure, prob->structure_len);
  if (!rc) rc = build_tables(c);
  if (!rc) rc = build_layout(c);
  if (!rc && (c->n_g >= (1LL << 31) || c->n_z >= (1LL << 31) || c->nnz_j >= (1LL << 31) || c->nnz_h >= (1LL << 31)))
    rc = fail(c, MPX_ERR_UNSUPPORTED, "problem too large for int32 patterns (n_z %lld, n_g %lld, nnz_jac %lld, nnz_hess %lld)", (long long)c->n_z,
              (long long)c->n_g, (long long)c->nnz_j, (long long)c->nnz_h);
  if (!rc && prob->code_object) rc = load_device(c, prob);
  if (rc) {
    g_create_error = c->err;
    mpx_destroy(c);
    return rc;
  }
  *out = c;
  return MPX_OK;
}

extern "C" int mpx_destroy(mpx_ctx* c) {
  if (!c) return MPX_OK;
  if (c->has_device || c->module) {
    (void)hipSetDevice(c->device);
    mpx_asm_release(c);
    if (c->res.box) {  // the resident kernel leaves when asked (or by itself: idle / lifetime limits)
      __atomic_store_n(&c->res.box->stop, 1u, __ATOMIC_SEQ_CST);
      if (c->res.stream) (void)hipStreamSynchronize(c->res.stream), (void)hipStreamDestroy(c->res.stream);
      (void)hipHostFree(c->res.box);
      for (void* q : {(void*)c->res.d_buckets, (void*)c->res.d_tile_bucket, (void*)c->res.d_bound, (void*)c->res.d_slots, (void*)c->res.d_seq, (void*)c->res.d_sync})
        if (q) (void)dev_free(q);
    }
    auto fr = [](void* p) {
      if (p) (void)dev_free(p);
    };
    for (auto& t : c->degs) fr(t.d_D), fr(t.d_Cmid), fr(t.d_tk), fr(t.d_Dmid), fr(t.d_tkm), fr(t.d_w), fr(t.d_DT), fr(t.d_CT);
    for (auto& B : c->buckets) fr(B.d_node_i), fr(B.d_node_sk);
    fr(c->d_htiles), fr(c->d_node_seg), fr(c->d_node_tk);
    fr(c->d_tiles), fr(c->d_Wnode), fr(c->d_seg_start), fr(c->d_lin_ptr), fr(c->d_lin_idx), fr(c->d_lin_row), fr(c->d_lin_coef);
    fr(c->d_mg_dst), fr(c->d_hc_dst), fr(c->d_th_dst);
    fr(c->partial.p), fr(c->wcum.p), fr(c->st_z.p), fr(c->st_p.p), fr(c->st_lam.p), fr(c->st_sig.p), fr(c->st_f.p);
    fr(c->st_g.p), fr(c->st_grad.p), fr(c->st_jac.p), fr(c->st_hess.p);
    fr(c->ccs_j.p), fr(c->ccs_h.p), fr(c->d_perm_j), fr(c->d_perm_h), fr(c->d_var_dst), fr(c->d_var_src);
    fr(c->d_lgroups), fr(c->d_lforeign), fr(c->d_lftab), fr(c->d_gl_halo_seg), fr(c->gl_halo.p), fr(c->gl_pnode.p), fr(c->st_ggx.p), fr(c->st_ggp.p), fr(c->gl_grad.p), fr(c->gl_jac.p);
    fr(c->d_lt_ptr), fr(c->d_lt_col), fr(c->d_lt_row), fr(c->d_lt_coef), fr(c->d_colind_j), fr(c->d_jrow);
    fr(c->d_gmap), fr(c->d_qmap), fr(c->d_abs_fpos), fr(c->d_abs_fstage), fr(c->d_abs_fn), fr(c->gtmp.p), fr(c->d_shard_ent[0]), fr(c->d_shard_ent[1]), fr(c->ea_scratch.p);
    if (c->h_scratch) (void)hipHostFree(c->h_scratch);
    if (c->h_flag) (void)hipHostFree(c->h_flag);
    if (c->ea_dbg) (void)hipHostFree(c->ea_dbg);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    for (auto e : c->prof_ev) (void)hipEventDestroy(e);
    for (auto& t : c->tune)
      for (auto e : t.ev)
        if (e) (void)hipEventDestroy(e);
    if (c->module) (void)hipModuleUnload(c->module);
  }
  delete c;
  return MPX_OK;
}

extern "C" int mpx_get_sizes(const mpx_ctx* c, mpx_sizes* o) {
  if (!c || !o) return MPX_ERR_INVALID;
  o->n_z = c->n_z;
  o->n_p = c->n_p;
  o->n_g = c->n_g;
  o->nnz_jac = c->nnz_j;
  o->nnz_hess = c->nnz_h;
  o->n_nodes = c->N;
  o->n_tiles = (int64_t)c->tiles.size();
  o->bytes_fgj = 8 * (2 * c->n_z + c->n_p + c->n_g + c->nnz_j + 1);
  o->bytes_hess = 8 * (c->n_z + c->n_p + c->n_g + 1 + c->nnz_h);
  return MPX_OK;
}

extern "C" int mpx_pattern_jac(const mpx_ctx* c, int32_t* row, int32_t* col) {
  if (!c || !row || !col) return MPX_ERR_INVALID;
  memcpy(row, c->jrow.data(), c->jrow.size() * sizeof(int32_t));
  memcpy(col, c->jcol.data(), c->jcol.size() * sizeof(int32_t));
  return MPX_OK;
}

extern "C" int mpx_pattern_hess(const mpx_ctx* c, int32_t* row, int32_t* col) {
  if (!c || !row || !col) return MPX_ERR_INVALID;
  memcpy(row, c->hrow.data(), c->hrow.size() * sizeof(int32_t));
  memcpy(col, c->hcol.data(), c->hcol.size() * sizeof(int32_t));
  return MPX_OK;
}

extern "C" int mpx_ccs_perm(const mpx_ctx* c, int which, int64_t* perm, int64_t* colind) {
  if (!c || !perm || !colind || (which != MPX_JAC && which != MPX_HESS)) return MPX_ERR_INVALID;
  const std::vector<int32_t>& r = which == MPX_JAC ? c->jrow : c->hrow;
  const std::vector<int32_t>& cc = which == MPX_JAC ? c->jcol : c->hcol;
  const int64_t nnz = (int64_t)r.size();
  std::iota(perm, perm + nnz, (int64_t)0);
  std::stable_sort(perm, perm + nnz, [&](int64_t a, int64_t b) { return cc[a] != cc[b] ? cc[a] < cc[b] : r[a] < r[b]; });
  std::fill(colind, colind + c->n_z + 1, (int64_t)0);
  for (int64_t k = 0; k < nnz; ++k) colind[cc[k] + 1]++;
  for (int64_t j = 0; j < c->n_z; ++j) colind[j + 1] += colind[j];
  return MPX_OK;
}

extern "C" int mpx_get_comp_weights(const mpx_ctx* c, double* w) {
  if (!c || !w || c->kind != 0) return MPX_ERR_INVALID;
  memcpy(w, c->compW.data(), c->compW.size() * sizeof(double));
  return MPX_OK;
}

extern "C" int mpx_geometry_reset(mpx_ctx* c) {
  if (!c) return MPX_ERR_INVALID;
  for (auto& t : c->tune) t.stage = -2, t.uses = 0;  // (re-allocated arrays: two unmeasured passes first, like a new entry)
  return MPX_OK;
}

extern "C" int mpx_set_mid_resid_output(mpx_ctx* c, double* resid) {
  if (!c) return MPX_ERR_INVALID;
  if (c->kind != 0) return fail(c, MPX_ERR_UNSUPPORTED, "assembled contexts have no mid-point residual pass");
  c->mid_resid_out = resid;
  return MPX_OK;
}

extern "C" int mpx_set_stream(mpx_ctx* c, void* stream) {
  if (!c) return MPX_ERR_INVALID;
  c->stream = (hipStream_t)stream;
  return MPX_OK;
}

extern "C" int mpx_set_tile_range(mpx_ctx* c, int64_t b, int64_t e, int run_boundary) {
  if (!c || b < 0 || e > (int64_t)c->tiles.size() || b > e) return fail(c, MPX_ERR_INVALID, "bad tile range");
  if (c->kind != 0) return fail(c, MPX_ERR_UNSUPPORTED, "assembled contexts have no tiles");
  if (c->shard_world > 1) return fail(c, MPX_ERR_INVALID, "mpx_set_tile_range on a context in segment-sharded mode (mpx_shard_setup(ctx, 1, 0) leaves it)");
  c->tile_begin = b;
  c->tile_end = e;
  c->run_boundary = run_boundary;
  return MPX_OK;
}

extern "C" int mpx_get_tile_jac_range(const mpx_ctx* c, int64_t t, int64_t* b, int64_t* e) {
  if (!c || t < 0 || t >= (int64_t)c->tiles.size() || !b || !e) return MPX_ERR_INVALID;
  // the block of a tile ends where the next block (in base order) starts
  const int64_t base = c->tiles[t].jac_base;
  int64_t end = c->jac_tiles_end;
  for (auto& o : c->tiles)
    if (o.jac_base > base && o.jac_base < end) end = o.jac_base;
  *b = base;
  *e = end;
  return MPX_OK;
}

extern "C" int mpx_get_tile_weights(const mpx_ctx* c, int64_t* w) {
  if (!c || !w) return MPX_ERR_INVALID;
  for (size_t t = 0; t < c->tiles.size(); ++t) {
    int64_t b, e;
    mpx_get_tile_jac_range(c, (int64_t)t, &b, &e);
    w[t] = e - b;
  }
  return MPX_OK;
}

extern "C" int mpx_get_tile_spans(const mpx_ctx* c, int32_t* first, int32_t* len, int32_t* n_foreign) {
  if (!c || !first || !len || !n_foreign) return MPX_ERR_INVALID;
  if (c->kind != 0) return MPX_ERR_UNSUPPORTED;
  for (size_t t = 0; t < c->tiles.size(); ++t) first[t] = c->tiles[t].span_lo, len[t] = c->tiles[t].span_len, n_foreign[t] = c->tiles[t].f_count;
  return MPX_OK;
}

extern "C" const char* mpx_get_notes(const mpx_ctx* c) { return c ? c->notes.c_str() : ""; }

extern "C" int mpx_get_light_plan(const mpx_ctx* c, int32_t* degree, int64_t* n_groups, int64_t* max_span_nodes, int64_t* n_low_degree_nodes) {
  if (!c) return MPX_ERR_INVALID;
  const bool ok = c->kind == 0 && c->lplan.ok;
  if (degree) *degree = ok ? c->lplan.deg : 0;
  if (n_groups) *n_groups = ok ? (c->lplan.low || c->lplan.high ? (int64_t)c->lplan.n_low_groups : (int64_t)c->lplan.groups.size()) : 0;
  if (max_span_nodes) *max_span_nodes = ok ? c->lplan.span_cap : 0;
  if (n_low_degree_nodes) *n_low_degree_nodes = ok ? (int64_t)c->lplan.foreign.size() : 0;
  return MPX_OK;
}

extern "C" int mpx_get_partials(mpx_ctx* c, int64_t batch, double** ptr, int64_t* count) {
  if (!c || !ptr || !count || batch < 1) return MPX_ERR_INVALID;
  if (!c->has_device) return fail(c, MPX_ERR_NO_DEVICE, "context has no device code");
  HIPCHK(c, hipSetDevice(c->device));
  int rc = reserve(c, c->partial, (size_t)(batch * partial_slots(c) * c->nred));
  if (rc) return rc;
  *ptr = c->partial.p;
  *count = batch * (int64_t)c->tiles.size() * c->nred;
  return MPX_OK;
}

extern "C" int mpx_sync(mpx_ctx* c) {
  if (!c) return MPX_ERR_INVALID;
  if (!c->has_device) return fail(c, MPX_ERR_NO_DEVICE, "context has no device code");
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return MPX_OK;
}

// device-side alias of a host array if it lies inside a page-locked range this context knows, else NULL
static void* pin_alias(const mpx_ctx* c, const void* p, size_t bytes) {
  const char* q = static_cast<const char*>(p);
  for (auto& r : c->pins)
    if (q >= r.base && q + bytes <= r.base + r.bytes) return r.dev + (q - r.base);
  return nullptr;
}

extern "C" int mpx_host_alloc(mpx_ctx* c, size_t bytes, void** ptr) {
  if (!c || !ptr) return MPX_ERR_INVALID;
  if (!c->has_device) return fail(c, MPX_ERR_NO_DEVICE, "context has no device code");
  HIPCHK(c, hipSetDevice(c->device));
  const size_t n = bytes ? bytes : 8;
  HIPCHK(c, hipHostMalloc(ptr, n, hipHostMallocMapped));
  if (poison_alloc()) memset(*ptr, poison_byte(), n);
  void* dev = nullptr;
  if (hipHostGetDevicePointer(&dev, *ptr, 0) == hipSuccess && dev) c->pins.push_back({static_cast<char*>(*ptr), n, static_cast<char*>(dev), true});
  return MPX_OK;
}

static void pin_forget(mpx_ctx* c, void* ptr) {
  for (size_t k = 0; k < c->pins.size(); ++k)
    if (c->pins[k].base == ptr) {
      c->pins.erase(c->pins.begin() + k);
      return;
    }
}

extern "C" int mpx_host_free(mpx_ctx* c, void* ptr) {
  if (!c) return MPX_ERR_INVALID;
  if (ptr) {
    pin_forget(c, ptr);
    HIPCHK(c, hipHostFree(ptr));
  }
  return MPX_OK;
}

extern "C" int mpx_host_register(mpx_ctx* c, void* ptr, size_t bytes) {
  if (!c || !ptr || !bytes) return MPX_ERR_INVALID;
  if (!c->has_device) return fail(c, MPX_ERR_NO_DEVICE, "context has no device code");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipHostRegister(ptr, bytes, hipHostRegisterMapped));
  void* dev = nullptr;
  if (hipHostGetDevicePointer(&dev, ptr, 0) == hipSuccess && dev) c->pins.push_back({static_cast<char*>(ptr), bytes, static_cast<char*>(dev), false});
  return MPX_OK;
}

extern "C" int mpx_host_unregister(mpx_ctx* c, void* ptr) {
  if (!c || !ptr) return MPX_ERR_INVALID;
  pin_forget(c, ptr);
  HIPCHK(c, hipHostUnregister(ptr));
  return MPX_OK;
}

extern "C" int mpx_timer_start(mpx_ctx* c) {
  if (!c || !c->has_device) return MPX_ERR_NO_DEVICE;
  HIPCHK(c, hipEventRecord(c->ev0, c->stream));
  return MPX_OK;
}

extern "C" int mpx_timer_stop(mpx_ctx* c, double* ms) {
  if (!c || !c->has_device || !ms) return MPX_ERR_NO_DEVICE;
  HIPCHK(c, hipEventRecord(c->ev1, c->stream));
  HIPCHK(c, hipEventSynchronize(c->ev1));
  float f = 0;
  HIPCHK(c, hipEventElapsedTime(&f, c->ev0, c->ev1));
  *ms = f;
  return MPX_OK;
}

extern "C" int mpx_profile(mpx_ctx* c, int enable) {
  if (!c || !c->has_device) return MPX_ERR_NO_DEVICE;
  c->profile = enable;
  return MPX_OK;
}

extern "C" int mpx_profile_read(mpx_ctx* c, double* ms, int64_t* n) {
  if (!c || !c->has_device || !ms || !n) return MPX_ERR_NO_DEVICE;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  double tot = 0;
  for (size_t k = 0; k + 1 < c->prof_used; k += 2) {
    float f = 0;
    HIPCHK(c, hipEventElapsedTime(&f, c->prof_ev[k], c->prof_ev[k + 1]));
    tot += f;
  }
  *ms = tot;
  *n = c->prof_launches;
  c->prof_used = 0;
  c->prof_launches = 0;
  return MPX_OK;
}

// ---- off-node evaluation ---------------------------------------------------------------------
struct ResidBucket {
  int deg = 0, n = 0;
  std::vector<int32_t> pt_id, pt_seg;
  std::vector<double> pt_tn, Cmat, Dmat;  // Cmat/Dmat: [deg+1][n]
  int32_t *d_id = nullptr, *d_seg = nullptr;
  double *d_tn = nullptr, *d_C = nullptr, *d_D = nullptr;
  hipFunction_t fn = nullptr;
};
struct mpx_resid_plan {
  mpx_ctx* ctx = nullptr;
  int phase = 0;
  int64_t n_pts = 0;
  std::vector<ResidBucket> buckets;
  DevBuf<double> st[7];
};

extern "C" int mpx_resid_plan_destroy(mpx_resid_plan* P) {
  if (!P) return MPX_OK;
  if (P->ctx && P->ctx->has_device) {
    (void)hipSetDevice(P->ctx->device);
    for (auto& B : P->buckets) {
      if (B.d_id) (void)dev_free(B.d_id);
      if (B.d_seg) (void)dev_free(B.d_seg);
      if (B.d_tn) (void)dev_free(B.d_tn);
      if (B.d_C) (void)dev_free(B.d_C);
      if (B.d_D) (void)dev_free(B.d_D);
    }
    for (auto& b : P->st)
      if (b.p) (void)dev_free(b.p);
  }
  delete P;
  return MPX_OK;
}

extern "C" int mpx_resid_plan_create(mpx_ctx* c, int phase, const int64_t* seg_ptr, const double* taus, mpx_resid_plan** out) {
  return mpx_resid_plan_create_order(c, phase, seg_ptr, taus, 1, out);
}

extern "C" int mpx_resid_plan_create_order(mpx_ctx* c, int phase, const int64_t* seg_ptr, const double* taus, int deriv_order,
                                           mpx_resid_plan** out) {
  if (!c || !seg_ptr || !out) return MPX_ERR_INVALID;
  if (deriv_order != 1 && deriv_order != 2) return fail(c, MPX_ERR_INVALID, "residual plan: derivative order must be 1 or 2");
  if (phase < 0 || phase >= c->n_phases) return fail(c, MPX_ERR_INVALID, "residual plan: phase out of range");
  if (seg_ptr[0] != 0) return fail(c, MPX_ERR_INVALID, "residual plan: seg_ptr[0] must be 0");
  for (int s = 0; s < c->S; ++s)
    if (seg_ptr[s + 1] < seg_ptr[s]) return fail(c, MPX_ERR_INVALID, "residual plan: seg_ptr must be non-decreasing");
  if (seg_ptr[c->S] > 0 && !taus) return MPX_ERR_INVALID;
  mpx_resid_plan* P = new (std::nothrow) mpx_resid_plan;
  if (!P) return MPX_ERR_ALLOC;
  P->ctx = c;
  P->phase = phase;
  P->n_pts = seg_ptr[c->S];
  for (auto& t : c->degs) {
    ResidBucket B;
    B.deg = t.deg;
    const int n1 = t.deg + 1;
    std::vector<double> Crows, Drows;  // [n][n1] then transposed
    for (int s = 0; s < c->S; ++s) {
      if (c->orders[s] != t.deg) continue;
      const int64_t a = seg_ptr[s], b = seg_ptr[s + 1];
      if (b == a) continue;
      std::vector<double> Cm((size_t)(b - a) * n1), Dm((size_t)(b - a) * n1);
      mpx_colloc_interp_matrix(t.roots.data(), n1, taus + a, (int)(b - a), Cm.data());
      mpx_colloc_diff_matrix(t.roots.data(), n1, taus + a, (int)(b - a), deriv_order, Dm.data());
      for (int64_t q = a; q < b; ++q) {
        B.pt_id.push_back((int32_t)q);
        B.pt_seg.push_back(s);
        B.pt_tn.push_back((taus[q] - c->tau0) / (c->tau1 - c->tau0));
      }
      Crows.insert(Crows.end(), Cm.begin(), Cm.end());
      Drows.insert(Drows.end(), Dm.begin(), Dm.end());
    }
    B.n = (int)B.pt_id.size();
    if (!B.n) continue;
    B.Cmat.resize((size_t)n1 * B.n);
    B.Dmat.resize((size_t)n1 * B.n);
    for (int m = 0; m < B.n; ++m)
      for (int j = 0; j < n1; ++j) {
        B.Cmat[(size_t)j * B.n + m] = Crows[(size_t)m * n1 + j];
        B.Dmat[(size_t)j * B.n + m] = Drows[(size_t)m * n1 + j];
      }
    P->buckets.push_back(std::move(B));
  }
  if (c->has_device) {
    int rc = MPX_OK;
    if (hipSetDevice(c->device) != hipSuccess) rc = MPX_ERR_HIP;
    for (auto& B : P->buckets) {
      if (rc) break;
      char name[96];
      snprintf(name, sizeof name, "mpx_resid_%d_%d", phase, B.deg);
      if (hipModuleGetFunction(&B.fn, c->module, name) != hipSuccess) {
        rc = fail(c, MPX_ERR_INVALID, "code object lacks kernel %s", name);
        break;
      }
      if ((rc = upload(c, &B.d_id, B.pt_id)) || (rc = upload(c, &B.d_seg, B.pt_seg)) || (rc = upload(c, &B.d_tn, B.pt_tn)) ||
          (rc = upload(c, &B.d_C, B.Cmat)) || (rc = upload(c, &B.d_D, B.Dmat)))
        break;
    }
    if (rc) {
      mpx_resid_plan_destroy(P);
      return rc;
    }
  }
  *out = P;
  return MPX_OK;
}

extern "C" int mpx_resid_eval_device(mpx_ctx* c, mpx_resid_plan* P, int64_t batch, const double* z, const double* p,
                                     int p_per_point, double* ti, double* xi, double* ui, double* dxi, double* dui,
                                     double* dyn, double* resid) {
  if (!c || !P || P->ctx != c) return MPX_ERR_INVALID;
  if (!c->has_device)
    return fail(c, MPX_ERR_NO_DEVICE, "mpx_resid_eval: context was created without a gfx950 code object; there is no CPU fallback");
  if (batch < 1 || !z || !p) return fail(c, MPX_ERR_INVALID, "mpx_resid_eval: batch/z/p invalid");
  HIPCHK(c, hipSetDevice(c->device));
  const int64_t n_w = p_per_point ? batch : 1;
  int rc;
  if ((rc = reserve_wcum(c, (size_t)(n_w * c->n_p)))) return rc;
  c->wcum_valid = false;
  if ((rc = launch_prefix(c, p, n_w, p_per_point))) return rc;
  const PhaseStruct& Ph = c->ph[P->phase];
  for (auto& B : P->buckets) {
    MpxResidArgs A{};
    A.z = z;
    A.z_stride = c->n_z;
    A.w = p;
    A.wcum = c->wcum.p;
    A.w_stride = p_per_point ? c->n_p : 0;
    A.pt_id = B.d_id;
    A.pt_seg = B.d_seg;
    A.pt_tn = B.d_tn;
    A.Cmat = B.d_C;
    A.Dmat = B.d_D;
    A.seg_start = c->d_seg_start;
    A.ti = ti, A.xi = xi, A.ui = ui, A.dxi = dxi, A.dui = dui, A.dyn = dyn, A.resid = resid;
    A.inv_dtau = 1.0 / (c->tau1 - c->tau0);
    A.z_off = Ph.z_off;
    A.n = B.n;
    A.n_pts = (int32_t)P->n_pts;
    A.N = (int32_t)c->N;
    A.seg_off = P->phase * c->S;
    A.B = (int32_t)batch;
    const int gx = (B.n + MPX_TILE - 1) / MPX_TILE;
    A.b_per_block = (int)std::min<int64_t>(std::max<int64_t>(batch * gx / 2048, 1), 16);
    const int gy = (int)((batch + A.b_per_block - 1) / A.b_per_block);
    if ((rc = launch(c, B.fn, dim3(gx, gy, 1), dim3(MPX_TILE, 1, 1), &A, sizeof A))) return rc;
  }
  return MPX_OK;
}

extern "C" int mpx_resid_eval(mpx_ctx* c, mpx_resid_plan* P, int64_t batch, const double* z, const double* p, int p_per_point,
                              double* ti, double* xi, double* ui, double* dxi, double* dui, double* dyn, double* resid) {
  if (!c || !P || P->ctx != c) return MPX_ERR_INVALID;
  if (!c->has_device)
    return fail(c, MPX_ERR_NO_DEVICE, "mpx_resid_eval: context was created without a gfx950 code object; there is no CPU fallback");
  if (batch < 1 || !z || !p) return fail(c, MPX_ERR_INVALID, "mpx_resid_eval: batch/z/p invalid");
  HIPCHK(c, hipSetDevice(c->device));
  int rc;
  const size_t Bz = (size_t)batch, npv = (size_t)(p_per_point ? batch : 1) * c->n_p, np_ = (size_t)std::max<int64_t>(P->n_pts, 1);
  if ((rc = reserve(c, c->st_z, Bz * c->n_z)) || (rc = reserve(c, c->st_p, npv))) return rc;
  HIPCHK(c, hipMemcpyAsync(c->st_z.p, z, Bz * c->n_z * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->st_p.p, p, npv * 8, hipMemcpyHostToDevice, c->stream));
  double* host[7] = {ti, xi, ui, dxi, dui, dyn, resid};
  const size_t width[7] = {1, (size_t)c->nx, (size_t)c->nu, (size_t)c->nx, (size_t)c->nu, (size_t)c->nx, (size_t)c->nx};
  double* dev[7];
  for (int k = 0; k < 7; ++k) {
    dev[k] = nullptr;
    if (host[k] && width[k]) {
      if ((rc = reserve(c, P->st[k], Bz * np_ * width[k]))) return rc;
      dev[k] = P->st[k].p;
    }
  }
  if ((rc = mpx_resid_eval_device(c, P, batch, c->st_z.p, c->st_p.p, p_per_point, dev[0], dev[1], dev[2], dev[3], dev[4], dev[5], dev[6])))
    return rc;
  for (int k = 0; k < 7; ++k)
    if (dev[k] && P->n_pts) HIPCHK(c, hipMemcpyAsync(host[k], dev[k], Bz * P->n_pts * width[k] * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return MPX_OK;
}



static int eval_core(mpx_ctx* c, int mask, int64_t batch, const double* z, const double* p, int p_per_point, const double* lam_g,
                     const double* sigma, double* f, double* g, double* grad_f, double* jac_val, double* hess_val, bool skip_prefix);

extern "C" int mpx_eval_device(mpx_ctx* c, int mask, int64_t batch, const double* z, const double* p, int p_per_point,
                               const double* lam_g, const double* sigma, double* f, double* g, double* grad_f,
                               double* jac_val, double* hess_val) {
  if (c) c->wcum_valid = false;  // caller-owned device widths: cannot be compared cheaply; MPX_WIDTHS_UNCHANGED is the caller's word
  if (c && (mask & MPX_OWNER_RESIDENT) && (c->kind != 0 || c->shard_world <= 1))
    return fail(c, MPX_ERR_INVALID, "MPX_OWNER_RESIDENT without mpx_shard_setup(world > 1)");
  return eval_core(c, mask & ~MPX_WIDTHS_UNCHANGED, batch, z, p, p_per_point, lam_g, sigma, f, g, grad_f, jac_val, hess_val,
                   (mask & MPX_WIDTHS_UNCHANGED) != 0);
}

// out[b][k] = in[b][perm[k]]
__global__ __launch_bounds__(256) void mpx_permute_kernel(const double* __restrict__ in, double* __restrict__ out, const int64_t* __restrict__ perm,
                                                          int64_t n) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) out[(int64_t)blockIdx.y * n + k] = in[(int64_t)blockIdx.y * n + perm[k]];
}

// out[b][dst[i]] = in[b][src[i]]: the (z, p)-dependent entries only (MPX_JAC_VARIABLE_ONLY | MPX_CCS_ORDER)
__global__ __launch_bounds__(256) void mpx_permute_sel_kernel(const double* __restrict__ in, double* __restrict__ out, const int64_t* __restrict__ dst,
                                                              const int64_t* __restrict__ src, int64_t n_sel, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_sel) out[(int64_t)blockIdx.y * n + dst[i]] = in[(int64_t)blockIdx.y * n + src[i]];
}

extern "C" int mpx_pattern_jac_variable(const mpx_ctx* c, uint8_t* is_variable) {
  if (!c || !is_variable) return MPX_ERR_INVALID;
  if (c->jac_var.size() != (size_t)c->nnz_j) {  // (assembled contexts: no classification -- everything counts as variable)
    memset(is_variable, 1, (size_t)c->nnz_j);
    return MPX_OK;
  }
  memcpy(is_variable, c->jac_var.data(), (size_t)c->nnz_j);
  return MPX_OK;
}

static int eval_native(mpx_ctx* c, int mask, int64_t batch, const double* z, const double* p, int p_per_point, const double* lam_g,
                       const double* sigma, double* f, double* g, double* grad_f, double* jac_val, double* hess_val, bool skip_prefix);

// compressed-column positions of the variable entries of jac_g and their native sources
static int upload_var_sel(mpx_ctx* c) {
  if (c->n_var_j >= 0) return MPX_OK;
  std::vector<int64_t> perm((size_t)std::max<int64_t>(c->nnz_j, 1)), colind((size_t)c->n_z + 1), dst, src;
  int rc = mpx_ccs_perm(c, MPX_JAC, perm.data(), colind.data());
  if (rc) return rc;
  for (int64_t k = 0; k < c->nnz_j; ++k)
    if (c->jac_var[(size_t)perm[(size_t)k]]) dst.push_back(k), src.push_back(perm[(size_t)k]);
  if ((rc = upload(c, &c->d_var_dst, dst)) || (rc = upload(c, &c->d_var_src, src))) return rc;
  c->n_var_j = (int64_t)dst.size();
  return MPX_OK;
}

static int upload_ccs_perm(mpx_ctx* c, int which, int64_t** dst) {
  if (*dst) return MPX_OK;
  const int64_t nnz = which == MPX_JAC ? c->nnz_j : c->nnz_h;
  std::vector<int64_t> perm((size_t)std::max<int64_t>(nnz, 1)), colind((size_t)c->n_z + 1);
  int rc = mpx_ccs_perm(c, which, perm.data(), colind.data());
  if (rc) return rc;
  perm.resize((size_t)nnz);
  return upload(c, dst, perm);
}

static int eval_core(mpx_ctx* c, int mask, int64_t batch, const double* z, const double* p, int p_per_point, const double* lam_g,
                     const double* sigma, double* f, double* g, double* grad_f, double* jac_val, double* hess_val, bool skip_prefix) {
  if (!c || !(mask & MPX_CCS_ORDER)) return eval_native(c, mask, batch, z, p, p_per_point, lam_g, sigma, f, g, grad_f, jac_val, hess_val, skip_prefix);
  if (mask & MPX_BOUNDARY_ONLY) return fail(c, MPX_ERR_INVALID, "MPX_CCS_ORDER cannot be combined with MPX_BOUNDARY_ONLY");
  // MPX_JAC_VARIABLE_ONLY | MPX_CCS_ORDER (round 6; what nlp_jac_g uses when the caller's array still holds the constants of a
  // previous call, mpx_casadi.cpp): the node kernels evaluate the WHOLE Jacobian into the device scratch as ever -- on the device
  // that costs nothing to speak of -- and only the (z, p)-dependent entries leave in compressed-column order: a single evaluation
  // through host pointers writes 0.25 instead of 0.96 MB over PCIe at config 2.  Contexts without the classification (assembled
  // ones) write everything.
  const bool var_only = (mask & MPX_JAC_VARIABLE_ONLY) && (mask & MPX_JAC) && c->kind == 0 && c->jac_var.size() == (size_t)c->nnz_j;
  mask &= ~MPX_JAC_VARIABLE_ONLY;
  if (c->shard_world > 1) return fail(c, MPX_ERR_INVALID, "MPX_CCS_ORDER on a context in segment-sharded mode");
  if (!c->has_device) return fail(c, MPX_ERR_NO_DEVICE, "mpx_eval: context was created without a gfx950 code object; there is no CPU fallback");
  if (batch < 1 || batch > 65535) return fail(c, MPX_ERR_INVALID, "MPX_CCS_ORDER: batch must be 1..65535");
  if (((mask & MPX_JAC) && !jac_val) || ((mask & MPX_HESS) && !hess_val)) return fail(c, MPX_ERR_INVALID, "mpx_eval: output array is NULL");
  HIPCHK(c, hipSetDevice(c->device));
  int rc;
  double *tj = jac_val, *th = hess_val;
  if ((mask & MPX_JAC) && c->nnz_j) {
    if ((rc = upload_ccs_perm(c, MPX_JAC, &c->d_perm_j)) || (rc = reserve(c, c->ccs_j, (size_t)(batch * c->nnz_j)))) return rc;
    tj = c->ccs_j.p;
  }
  if ((mask & MPX_HESS) && c->nnz_h) {
    if ((rc = upload_ccs_perm(c, MPX_HESS, &c->d_perm_h)) || (rc = reserve(c, c->ccs_h, (size_t)(batch * c->nnz_h)))) return rc;
    th = c->ccs_h.p;
  }
  if ((rc = eval_native(c, mask & ~MPX_CCS_ORDER, batch, z, p, p_per_point, lam_g, sigma, f, g, grad_f, tj, th, skip_prefix))) return rc;
  if (tj != jac_val && var_only) {
    if ((rc = upload_var_sel(c))) return rc;
    if (c->n_var_j > 0)
      hipLaunchKernelGGL(mpx_permute_sel_kernel, dim3((unsigned)((c->n_var_j + 255) / 256), (unsigned)batch), dim3(256), 0, c->stream, tj, jac_val, c->d_var_dst,
                         c->d_var_src, c->n_var_j, c->nnz_j);
  } else if (tj != jac_val)
    hipLaunchKernelGGL(mpx_permute_kernel, dim3((unsigned)((c->nnz_j + 255) / 256), (unsigned)batch), dim3(256), 0, c->stream, tj, jac_val, c->d_perm_j, c->nnz_j);
  if (th != hess_val) hipLaunchKernelGGL(mpx_permute_kernel, dim3((unsigned)((c->nnz_h + 255) / 256), (unsigned)batch), dim3(256), 0, c->stream, th, hess_val, c->d_perm_h, c->nnz_h);
  HIPCHK(c, hipGetLastError());
  return MPX_OK;
}

// The batched I/O view of one evaluation call (what every kernel of the call sees).
static int make_io(mpx_ctx* c, int mask, int64_t batch, const double* z, const double* p, int p_per_point, const double* lam_g, const double* sigma,
                   double* f, double* g, double* grad_f, double* jac_val, double* hess_val, MpxIO& io) {
  io = MpxIO{};
  io.z = z;
  io.z_stride = c->n_z;
  io.w = p;
  io.wcum = c->wcum.p;
  io.w_stride = p_per_point ? c->n_p : 0;
  io.lam_g = lam_g;
  io.lam_stride = c->n_g;
  io.sigma = sigma;
  io.f = (mask & MPX_F) ? f : nullptr;
  io.g = (mask & MPX_G) ? g : nullptr;
  io.g_stride = c->n_g;
  io.grad = (mask & MPX_GRAD) ? grad_f : nullptr;
  io.grad_stride = c->n_z;
  io.jac = (mask & MPX_JAC) ? jac_val : nullptr;
  io.jac_variable_only = (mask & MPX_JAC_VARIABLE_ONLY) ? 1 : 0;
  io.jac_stride = c->nnz_j;
  io.hess = hess_val;
  io.hess_stride = c->nnz_h;
  if (mask & MPX_MID_RESID) {
    if (!(mask & MPX_HESS) || !c->mid_resid_out) return fail(c, MPX_ERR_INVALID, "MPX_MID_RESID needs MPX_HESS and mpx_set_mid_resid_output");
    for (auto& t : c->degs)
      if (t.deg > 12) return fail(c, MPX_ERR_UNSUPPORTED, "MPX_MID_RESID: polynomial degree %d > 12 (use a residual plan)", t.deg);
    if (c->shard_world > 1) return fail(c, MPX_ERR_UNSUPPORTED, "MPX_MID_RESID on a context in segment-sharded mode");
    if (c->hess_by_node) return fail(c, MPX_ERR_UNSUPPORTED, "MPX_MID_RESID on a mixed-degree grid (its hess_l tiles are node-ordered: use a residual plan)");
    io.mid_resid = c->mid_resid_out;
    io.mid_stride = (int64_t)c->n_phases * (c->N - 1) * c->nx;
  }
  io.partial = c->partial.p;
  io.n_tiles_total = (int32_t)c->tiles.size();
  io.nred = c->nred;
  io.B = (int32_t)batch;
  return MPX_OK;
}

static int eval_native(mpx_ctx* c, int mask, int64_t batch, const double* z, const double* p, int p_per_point, const double* lam_g,
                       const double* sigma, double* f, double* g, double* grad_f, double* jac_val, double* hess_val, bool skip_prefix) {
  if (!c) return MPX_ERR_INVALID;
  if (!c->has_device)
    return fail(c, MPX_ERR_NO_DEVICE, "mpx_eval: context was created without a gfx950 code object; there is no CPU fallback");
  if (batch < 1 || batch > (1 << 30) || !z || (!p && c->n_p > 0)) return fail(c, MPX_ERR_INVALID, "mpx_eval: batch/z/p invalid");
  if ((mask & MPX_HESS) && (!lam_g || !sigma || !hess_val)) return fail(c, MPX_ERR_INVALID, "mpx_eval: HESS needs lam_g, sigma, hess_val");
  if ((mask & MPX_F) && !f) return fail(c, MPX_ERR_INVALID, "mpx_eval: f is NULL");
  if ((mask & MPX_G) && !g) return fail(c, MPX_ERR_INVALID, "mpx_eval: g is NULL");
  if ((mask & MPX_GRAD) && !grad_f) return fail(c, MPX_ERR_INVALID, "mpx_eval: grad_f is NULL");
  if ((mask & MPX_JAC) && !jac_val) return fail(c, MPX_ERR_INVALID, "mpx_eval: jac_val is NULL");
  HIPCHK(c, hipSetDevice(c->device));
  if (c->kind == 1) return mpx_asm_eval_device(c, mask, batch, z, lam_g, sigma, f, g, grad_f, jac_val, hess_val);
  const int64_t n_w = p_per_point ? batch : 1;
  int rc;
  if ((rc = reserve_wcum(c, (size_t)(n_w * c->n_p)))) return rc;
  if ((rc = reserve(c, c->partial, (size_t)(batch * partial_slots(c) * c->nred)))) return rc;
  // MPX_WIDTHS_UNCHANGED (or the host path's "same p"): only if the buffer really holds the prefix sums of THIS p for every phase
  if (skip_prefix && !(c->wcum_p == p && c->wcum_batch == n_w && c->wcum_ppp == (p_per_point ? 1 : 0) && c->wcum_phases == all_phases(c))) skip_prefix = false;
  // (a problem none of whose node functions uses the node time never reads the prefix sums -- th is their only consumer in these
  // kernels --: one launch and one kernel boundary less per pass, 4.7 us of a 12 us single evaluation on device pointers.  The
  // residual pass (times of the off-node points) and nlp_grad launch their own.  MPX_ALWAYS_PREFIX=1: A/B)
  if (!skip_prefix && c->time_dep && (rc = launch_prefix(c, p, n_w, p_per_point))) return rc;
  MpxIO io{};
  if ((rc = make_io(c, mask, batch, z, p, p_per_point, lam_g, sigma, f, g, grad_f, jac_val, hess_val, io))) return rc;
  const bool nodes = !(mask & MPX_BOUNDARY_ONLY), owner = (mask & MPX_OWNER_RESIDENT) != 0;
  if (!nodes && !c->run_boundary && c->shard_world <= 1) return fail(c, MPX_ERR_INVALID, "MPX_BOUNDARY_ONLY with the boundary pass disabled");
  if (mask & (MPX_GRAD | MPX_JAC)) {
    if ((rc = run_mode(c, MPX_MODE_FGJ, io, nodes, owner))) return rc;
  } else if (mask & (MPX_F | MPX_G)) {
    if ((rc = run_mode(c, MPX_MODE_FG, io, nodes, owner))) return rc;
  }
  if (mask & MPX_HESS) {
    if ((rc = run_mode(c, MPX_MODE_HESS, io, nodes, owner))) return rc;
  }
  return MPX_OK;
}

// Completion of a zero-copy evaluation: the last kernel of the stream stores a sequence number into page-locked host memory
// (system-scope release after a system fence) and the host spins on it.  hipStreamSynchronize costs ~5 us more per call on
// this stack (tools/zc_probe.hip: empty kernel + sync 9.6 us, flag kernel + spin 6.0 us; 120 KB written to host memory:
// 15.0 vs 10.0 us; no stale data in 600 checked hand-offs) -- a quarter of a single evaluation.
__global__ void mpx_signal_kernel(unsigned long long* flag, unsigned long long seq) {
  __threadfence_system();
  __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

static int wait_flag(mpx_ctx* c) {
  hipLaunchKernelGGL(mpx_signal_kernel, dim3(1), dim3(64), 0, c->stream, c->h_flag_dev, ++c->flag_seq);
  HIPCHK(c, hipGetLastError());
  const auto t0 = std::chrono::steady_clock::now();
  for (uint64_t spins = 1; __atomic_load_n(c->h_flag, __ATOMIC_ACQUIRE) != c->flag_seq; ++spins) {
    if ((spins & 0xfffff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
      HIPCHK(c, hipStreamSynchronize(c->stream));  // something is wrong (or very slow): let the runtime report it
      break;
    }
  }
  return MPX_OK;
}

// ---- resident kernel: launch, request, completion (mpx_kernels.h: resident_loop) ---------------------------------------------------
static int res_launch(mpx_ctx* c) {
  mpx_ctx::Resident& R = c->res;
  HIPCHK(c, hipStreamSynchronize(R.stream));  // (an earlier instance has left)
  int rc;
  if ((rc = upload_ccs_perm(c, MPX_JAC, &c->d_perm_j)) || (rc = upload_ccs_perm(c, MPX_HESS, &c->d_perm_h))) return rc;
  const unsigned long long zero = 0;
  HIPCHK(c, hipMemcpy(R.d_seq, &R.word, sizeof R.word, hipMemcpyHostToDevice));  // nothing pending for the workgroups that poll the device copy
  HIPCHK(c, hipMemcpy(R.d_sync, &zero, sizeof zero, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(R.d_slots, R.slot, sizeof R.slot, hipMemcpyHostToDevice));  // (the slots the host believes the device to hold)
  R.box->seq = R.word, R.box->done = R.seq, R.box->exited = 0, R.box->stop = 0;
  __atomic_store_n(&R.box->alive, 1u, __ATOMIC_SEQ_CST);
  static const double idle_ms = getenv("MPX_RESIDENT_IDLE_MS") ? atof(getenv("MPX_RESIDENT_IDLE_MS")) : 20.0;
  MpxResidentArgs A{};
  A.box = R.box_dev, A.buckets = R.d_buckets, A.tile_bucket = R.d_tile_bucket, A.bound = R.d_bound, A.dev_slots = R.d_slots, A.dev_seq = R.d_seq, A.sync_count = R.d_sync;
  A.perm_j = c->d_perm_j, A.perm_h = c->d_perm_h, A.nnz_j = c->nnz_j, A.nnz_h = c->nnz_h;
  A.start_seq = R.word, A.idle_ticks = (long long)(idle_ms * 1e5), A.life_ticks = 200000000LL /* 2 s of the 100 MHz clock */, A.n_tiles = (int32_t)c->tiles.size();
  size_t size = sizeof A;
  void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &A, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  HIPCHK(c, hipModuleLaunchKernel(R.fn, (unsigned)c->tiles.size(), 1, 1, MPX_TILE, 1, 1, 0, R.stream, nullptr, cfg));
  R.launched = true;
  ++R.n_launches;
  return MPX_OK;
}

// One pass (mode) of a single evaluation through the resident kernel; blocks until the results are in the caller's arrays.
static int res_request(mpx_ctx* c, int mode, int ccs, const MpxIO& io, double* ccs_out) {
  mpx_ctx::Resident& R = c->res;
  int rc;
  if (!R.launched && (rc = res_launch(c))) return rc;
  // the argument slot: one the device already holds (an NLP solver passes the same work-vector slices to the same function on every
  // call), else the next one round robin, flagged as new
  MpxResRequest rq;
  memset(&rq, 0, sizeof rq);
  rq.mode = mode, rq.ccs = ccs, rq.io = io, rq.ccs_out = ccs_out;
  int sl = -1;
  for (int k = 0; k < R.n_slots && sl < 0; ++k)
    if (memcmp(&R.slot[k], &rq, sizeof rq) == 0) sl = k;
  unsigned long long fresh = 0;
  if (sl < 0) {
    sl = R.next_slot, R.next_slot = (R.next_slot + 1) % MPX_RES_SLOTS, R.n_slots = std::max(R.n_slots, sl + 1);
    R.slot[sl] = rq, R.box->slots[sl] = rq, fresh = 1;
  }
  const unsigned long long q = ++R.seq, word = (q << 8) | ((unsigned long long)sl << 1) | fresh;
  R.word = word;
  __atomic_store_n(&R.box->seq, word, __ATOMIC_SEQ_CST);
  ++R.n_requests;
  const auto t0 = std::chrono::steady_clock::now();
  for (uint64_t spins = 1;; ++spins) {
    if (__atomic_load_n(&R.box->done, __ATOMIC_ACQUIRE) == q) {
      static const bool dbg = getenv("MPX_RES_DEBUG") != nullptr;  // with a code object built with -DMPX_RES_STAMPS
      if (dbg) {
        static double acc[6] = {0, 0, 0, 0, 0, 0};
        static int n = 0;
        const long long* st = reinterpret_cast<const long long*>(&R.box->slots[MPX_RES_SLOTS - 1]);
        const double host_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        for (int k = 0; k < 5; ++k) acc[k] += (st[k + 1] - st[k]) / 100.0;
        acc[5] += host_us;
        if (++n == 500) {
          fprintf(stderr, "resident request (mode %d, us): args %.2f  node pass %.2f  barrier %.2f  boundary %.2f  last barrier (system fence) %.2f | device total %.2f, host round trip %.2f\n", mode,
                  acc[0] / n, acc[1] / n, acc[2] / n, acc[3] / n, acc[4] / n, (acc[0] + acc[1] + acc[2] + acc[3] + acc[4]) / n, acc[5] / n);
          for (double& a : acc) a = 0;
          n = 0;
        }
      }
      return MPX_OK;
    }
    if ((spins & 0x3ff) == 0) {
      if (!__atomic_load_n(&R.box->alive, __ATOMIC_SEQ_CST)) {
        // the kernel is leaving (idle / lifetime limit): it looks at seq once more after clearing `alive` -- either it serves this
        // request after all, or it sets `exited` and a new instance takes it
        while (__atomic_load_n(&R.box->done, __ATOMIC_ACQUIRE) != q && !__atomic_load_n(&R.box->exited, __ATOMIC_SEQ_CST) &&
               std::chrono::steady_clock::now() - t0 < std::chrono::seconds(5)) {}
        if (__atomic_load_n(&R.box->done, __ATOMIC_ACQUIRE) == q) return MPX_OK;
        if (!__atomic_load_n(&R.box->exited, __ATOMIC_SEQ_CST)) break;
        R.word = 0, --R.seq;  // (the new instance starts from "nothing seen", holding every slot the host knows, this request's included)
        rc = res_launch(c);
        R.seq = q, R.word = word;
        if (rc) return rc;
        __atomic_store_n(&R.box->seq, word, __ATOMIC_SEQ_CST);
      } else if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) {
        break;
      }
    }
  }
  __atomic_store_n(&R.box->stop, 1u, __ATOMIC_SEQ_CST);
  R.ok = false;  // never again on this context
  return fail(c, MPX_ERR_HIP, "resident kernel: no completion of request %llu within 5 s", q);
}

// A single evaluation through the resident kernel (zero-copy arrays already resolved to their device aliases).
static int res_eval(mpx_ctx* c, int mask, const double* z, const double* p, const double* lam_g, const double* sigma, double* f, double* g, double* grad_f,
                    double* jac_val, double* hess_val, bool same_p) {
  int rc;
  if ((rc = reserve_wcum(c, (size_t)c->n_p)) || (rc = reserve(c, c->partial, (size_t)(partial_slots(c) * c->nred)))) return rc;
  if (!(same_p && c->wcum_p == p && c->wcum_batch == 1 && c->wcum_ppp == 0 && c->wcum_phases == all_phases(c))) {
    if ((rc = launch_prefix(c, p, 1, 0))) return rc;  // (the widths changed: rare -- an NLP solver keeps p for a whole solve)
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  const bool ccs = (mask & MPX_CCS_ORDER) != 0;
  double *tj = jac_val, *th = hess_val;
  if (ccs && (mask & MPX_JAC) && c->nnz_j) {
    if ((rc = reserve(c, c->ccs_j, (size_t)c->nnz_j))) return rc;
    tj = c->ccs_j.p;
  }
  if (ccs && (mask & MPX_HESS) && c->nnz_h) {
    if ((rc = reserve(c, c->ccs_h, (size_t)c->nnz_h))) return rc;
    th = c->ccs_h.p;
  }
  MpxIO io{};
  if ((rc = make_io(c, mask & ~MPX_CCS_ORDER, 1, z, p, 0, lam_g, sigma, f, g, grad_f, tj, th, io))) return rc;
  io.b_per_block = 1;
  if (mask & (MPX_F | MPX_G | MPX_GRAD | MPX_JAC))
    if ((rc = res_request(c, MPX_MODE_FGJ, tj != jac_val ? 1 : 0, io, jac_val))) return rc;
  if (mask & MPX_HESS)
    if ((rc = res_request(c, MPX_MODE_HESS, th != hess_val ? 2 : 0, io, hess_val))) return rc;
  return MPX_OK;
}

extern "C" int mpx_eval(mpx_ctx* c, int mask, int64_t batch, const double* z, const double* p, int p_per_point,
                        const double* lam_g, const double* sigma, double* f, double* g, double* grad_f, double* jac_val,
                        double* hess_val) {
  if (!c) return MPX_ERR_INVALID;
  if (!c->has_device)
    return fail(c, MPX_ERR_NO_DEVICE, "mpx_eval: context was created without a gfx950 code object; there is no CPU fallback");
  if (batch < 1 || !z || (!p && c->n_p > 0)) return fail(c, MPX_ERR_INVALID, "mpx_eval: batch/z/p invalid");
  if (c->shard_world > 1)  // a sharded context runs node pass / exchange / boundary pass as separate device-pointer calls
    return fail(c, MPX_ERR_INVALID, "mpx_eval on a context in segment-sharded mode: the host-pointer path (and the nlp_* entry points) would return "
                                    "this rank's share only; use the mpx_eval_device / mpx_shard_* sequence, or mpx_shard_setup(ctx, 1, 0) first");
  const double t_entry = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
  HIPCHK(c, hipSetDevice(c->device));
  int rc;
  const size_t B = (size_t)batch;
  const size_t npv = (size_t)(p_per_point ? batch : 1) * c->n_p;
  const size_t cap_p = c->st_p.cap, cap_w = c->wcum.cap;
  if ((mask & MPX_HESS) && (!lam_g || !sigma || !hess_val)) return fail(c, MPX_ERR_INVALID, "mpx_eval: HESS needs lam_g, sigma, hess_val");
  if (((mask & MPX_F) && !f) || ((mask & MPX_G) && !g) || ((mask & MPX_GRAD) && !grad_f) || ((mask & MPX_JAC) && !jac_val))
    return fail(c, MPX_ERR_INVALID, "mpx_eval: a requested output array is NULL");
  if ((rc = reserve(c, c->st_p, npv)) || (rc = reserve_wcum(c, npv))) return rc;
  const bool same_p = npv == 0 || c->wcum_valid && cap_p == c->st_p.cap && cap_w == c->wcum.cap && c->last_p.size() == npv &&
                      memcmp(c->last_p.data(), p, npv * 8) == 0;
  if (!same_p) {
    HIPCHK(c, hipMemcpyAsync(c->st_p.p, p, npv * 8, hipMemcpyHostToDevice, c->stream));
    c->last_p.assign(p, p + npv);
  }
  // ---- zero-copy path (the single-evaluation regime an NLP solver drives): every array lies in page-locked memory this
  // context knows (mpx_host_alloc / mpx_host_register / mpx_current_pin_buffers), so the kernels read z (and lam_g) and
  // write the results straight over PCIe: no staging copies, no copy-engine launches; the widths stay cached on the
  // device.  Measured on MI355X (tools/zc_probe.hip): a kernel storing 1.3 MB into mapped host memory + sync 35 us against
  // 46 us for kernel + one D2H copy + sync (and one copy per output array before); launch + sync floor 10.5 us.
  {
    static const bool zc_off = getenv("MPX_NO_ZERO_COPY") != nullptr;
    size_t total = B * c->n_z * 8;
    if (mask & MPX_G) total += B * c->n_g * 8;
    if (mask & MPX_GRAD) total += B * c->n_z * 8;
    if (mask & MPX_JAC) total += B * c->nnz_j * 8;
    if (mask & MPX_HESS) total += B * (c->nnz_h + c->n_g) * 8;
    void* zd = (!zc_off && total <= (size_t)(16u << 20) && B <= 4096) ? pin_alias(c, z, B * c->n_z * 8) : nullptr;
    void *gd = nullptr, *qd = nullptr, *jd = nullptr, *hd = nullptr, *ld = nullptr;
    bool ok = zd != nullptr;
    if (ok && (mask & MPX_G)) ok = (gd = pin_alias(c, g, B * c->n_g * 8)) != nullptr;
    if (ok && (mask & MPX_GRAD)) ok = (qd = pin_alias(c, grad_f, B * c->n_z * 8)) != nullptr;
    if (ok && (mask & MPX_JAC)) ok = c->nnz_j == 0 || (jd = pin_alias(c, jac_val, B * c->nnz_j * 8)) != nullptr;
    if (ok && (mask & MPX_HESS))
      ok = (c->nnz_h == 0 || (hd = pin_alias(c, hess_val, B * c->nnz_h * 8)) != nullptr) && (ld = pin_alias(c, lam_g, B * c->n_g * 8)) != nullptr;
    if (ok) {
      if (c->h_scratch_cap < 2 * B) {  // page-locked scalars: f out, sigma in
        if (c->h_scratch) (void)hipHostFree(c->h_scratch);  // (the completion flag below does not depend on the batch size)
        c->h_scratch = nullptr, c->h_scratch_cap = 0;
        const size_t cap = std::max<size_t>(2 * B, 64);
        HIPCHK(c, hipHostMalloc((void**)&c->h_scratch, cap * 8, hipHostMallocMapped));
        if (poison_alloc()) memset(c->h_scratch, poison_byte(), cap * 8);
        HIPCHK(c, hipHostGetDevicePointer((void**)&c->h_scratch_dev, c->h_scratch, 0));
        c->h_scratch_cap = cap;
      }
      if (!c->h_flag) {
        HIPCHK(c, hipHostMalloc((void**)&c->h_flag, 64, hipHostMallocMapped));
        HIPCHK(c, hipHostGetDevicePointer((void**)&c->h_flag_dev, c->h_flag, 0));
        *c->h_flag = 0;
      }
      if (mask & MPX_HESS) memcpy(c->h_scratch + B, sigma, B * 8);
      static const bool lat_dbg = getenv("MPX_LAT_DEBUG") != nullptr;  // where a single evaluation spends its host time
      static double acc[3] = {0, 0, 0};
      static int n_acc = 0;
      auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
      const double t1 = lat_dbg ? now() : 0;
      // single evaluations through the resident kernel (no launch, no stream synchronisation) -- OPT-IN (MPX_RESIDENT=1): measured on
      // MI355X it is not faster than the launched kernels (moon lander 20x3: 43.5 against 44.1 us per IPOPT iteration, 1000x5: 127
      // against 101 us; profiles/r4_resident): what the device saves in launches it spends polling and fencing over PCIe
      const bool res_on = mpx_knob(MPX_K_RESIDENT) != nullptr;
      const bool resident = res_on && c->res.ok && B == 1 && c->kind == 0 && !(mask & ~(MPX_F | MPX_G | MPX_GRAD | MPX_JAC | MPX_HESS | MPX_CCS_ORDER)) &&
                            c->tile_begin == 0 && c->tile_end == (int64_t)c->tiles.size() && !mpx_knob(MPX_K_NO_RESIDENT) &&
                            // (passes the span kernels take -- no Jacobian / Hessian values, light plan -- stay with them: their f is
                            // summed per 64-node chunk, the resident kernel's per tile)
                            !(c->lplan.ok && !(mask & (MPX_JAC | MPX_HESS)) && !mpx_knob(MPX_K_NO_LIGHT));
      if (resident) {
        rc = res_eval(c, mask, (const double*)zd, c->st_p.p, (const double*)ld, c->h_scratch_dev + B, c->h_scratch_dev, (double*)gd, (double*)qd, (double*)jd,
                      (double*)hd, same_p);
        c->wcum_valid = rc == MPX_OK;
        if (rc) return rc;
        if (mask & MPX_F) memcpy(f, c->h_scratch, B * 8);
        return MPX_OK;
      }
      rc = eval_core(c, mask, batch, (const double*)zd, c->st_p.p, p_per_point, (const double*)ld, c->h_scratch_dev + B, c->h_scratch_dev, (double*)gd,
                     (double*)qd, (double*)jd, (double*)hd, same_p);
      if (rc) {
        c->wcum_valid = false;
        return rc;
      }
      c->wcum_valid = true;
      const double t2 = lat_dbg ? now() : 0;
      static const bool no_flag = getenv("MPX_NO_FLAG_WAIT") != nullptr;
      if (no_flag)
        HIPCHK(c, hipStreamSynchronize(c->stream));
      else if ((rc = wait_flag(c)))
        return rc;
      if (lat_dbg) {
        const double t3 = now();
        acc[0] += t1 - t_entry, acc[1] += t2 - t1, acc[2] += t3 - t2;
        if (++n_acc == 2000) {
          fprintf(stderr, "mpx_eval zero-copy (mask %d): setup %.2f us, launches %.2f us, sync wait %.2f us\n", mask, acc[0] / n_acc, acc[1] / n_acc, acc[2] / n_acc);
          acc[0] = acc[1] = acc[2] = 0, n_acc = 0;
        }
      }
      if (mask & MPX_F) memcpy(f, c->h_scratch, B * 8);
      return MPX_OK;
    }
  }
  if (mask & MPX_CCS_ORDER) mask &= ~MPX_JAC_VARIABLE_ONLY;  // (staged copies move whole arrays: the full pass)
  if ((rc = reserve(c, c->st_z, B * c->n_z))) return rc;
  HIPCHK(c, hipMemcpyAsync(c->st_z.p, z, B * c->n_z * 8, hipMemcpyHostToDevice, c->stream));
  if (mask & MPX_HESS) {
    if (!lam_g || !sigma || !hess_val) return fail(c, MPX_ERR_INVALID, "mpx_eval: HESS needs lam_g, sigma, hess_val");
    if ((rc = reserve(c, c->st_lam, B * c->n_g)) || (rc = reserve(c, c->st_sig, B)) || (rc = reserve(c, c->st_hess, B * std::max<int64_t>(c->nnz_h, 1))))
      return rc;
    HIPCHK(c, hipMemcpyAsync(c->st_lam.p, lam_g, B * c->n_g * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->st_sig.p, sigma, B * 8, hipMemcpyHostToDevice, c->stream));
  }
  if ((mask & MPX_F) && (rc = reserve(c, c->st_f, B))) return rc;
  if ((mask & MPX_G) && (rc = reserve(c, c->st_g, B * c->n_g))) return rc;
  if ((mask & MPX_GRAD) && (rc = reserve(c, c->st_grad, B * c->n_z))) return rc;
  if ((mask & MPX_JAC) && (rc = reserve(c, c->st_jac, B * std::max<int64_t>(c->nnz_j, 1)))) return rc;
  rc = eval_core(c, mask, batch, c->st_z.p, c->st_p.p, p_per_point, c->st_lam.p, c->st_sig.p, c->st_f.p, c->st_g.p, c->st_grad.p,
                 c->st_jac.p, c->st_hess.p, same_p);
  if (rc) {
    c->wcum_valid = false;
    return rc;
  }
  c->wcum_valid = true;
  if (mask & MPX_F) HIPCHK(c, hipMemcpyAsync(f, c->st_f.p, B * 8, hipMemcpyDeviceToHost, c->stream));
  if (mask & MPX_G) HIPCHK(c, hipMemcpyAsync(g, c->st_g.p, B * c->n_g * 8, hipMemcpyDeviceToHost, c->stream));
  if (mask & MPX_GRAD) HIPCHK(c, hipMemcpyAsync(grad_f, c->st_grad.p, B * c->n_z * 8, hipMemcpyDeviceToHost, c->stream));
  if (mask & MPX_JAC) HIPCHK(c, hipMemcpyAsync(jac_val, c->st_jac.p, B * c->nnz_j * 8, hipMemcpyDeviceToHost, c->stream));
  if (mask & MPX_HESS) HIPCHK(c, hipMemcpyAsync(hess_val, c->st_hess.p, B * c->nnz_h * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return MPX_OK;
}

// ---- nlp_grad: grad_gamma_x, grad_gamma_p (the sixth oracle of ca.nlpsol, mpopt.py:757) ----------------------------------
// out[b][col] = sigma[b] * grad_f[b][col] + sum over the entries of column col of  jac_val[b][e] * lam_g[b][row[e]]  in
// compressed-column order: the generic route (assembled contexts, whose Jacobian comes out of a gather pass; MPX_GRADL_GENERIC=1
// for the tiled contexts as a cross-check of the fused pass).  One lane per column, fixed order.
__global__ __launch_bounds__(256) void mpx_jtvec_kernel(const double* __restrict__ jac, int64_t nnz, const double* __restrict__ grad, const double* __restrict__ lam,
                                                        int64_t n_g, const double* __restrict__ sigma, const int64_t* __restrict__ colind,
                                                        const int64_t* __restrict__ perm, const int32_t* __restrict__ jrow, int64_t n_z,
                                                        double* __restrict__ out) {
  const int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (col >= n_z) return;
  const double* __restrict__ jb = jac + b * nnz;
  const double* __restrict__ lb = lam + b * n_g;
  double s = sigma[b] * grad[b * n_z + col];
  for (int64_t k = colind[col]; k < colind[col + 1]; ++k) {
    const int64_t e = perm[k];
    s = fma(jb[e], lb[jrow[e]], s);
  }
  out[b * n_z + col] = s;
}

static int grad_gamma_generic(mpx_ctx* c, int64_t batch, const double* z, const double* p, int p_per_point, const double* lam_g,
                              const double* sigma, double* ggx) {
  if (batch > 65535) return fail(c, MPX_ERR_UNSUPPORTED, "mpx_eval_grad_gamma: batch must be <= 65535 on this route");
  int rc;
  if (!c->d_colind_j) {
    std::vector<int64_t> perm((size_t)std::max<int64_t>(c->nnz_j, 1)), colind((size_t)c->n_z + 1);
    if ((rc = mpx_ccs_perm(c, MPX_JAC, perm.data(), colind.data()))) return rc;
    if ((rc = upload(c, &c->d_colind_j, colind)) || (rc = upload(c, &c->d_jrow, c->jrow))) return rc;
  }
  if ((rc = upload_ccs_perm(c, MPX_JAC, &c->d_perm_j))) return rc;
  if ((rc = reserve(c, c->gl_grad, (size_t)(batch * c->n_z))) || (rc = reserve(c, c->gl_jac, (size_t)(batch * std::max<int64_t>(c->nnz_j, 1))))) return rc;
  if ((rc = eval_native(c, MPX_GRAD | MPX_JAC, batch, z, p, p_per_point, nullptr, nullptr, nullptr, nullptr, c->gl_grad.p, c->gl_jac.p, nullptr, false))) return rc;
  hipLaunchKernelGGL(mpx_jtvec_kernel, dim3((unsigned)((c->n_z + 255) / 256), (unsigned)batch), dim3(256), 0, c->stream, c->gl_jac.p, c->nnz_j, c->gl_grad.p, lam_g,
                     c->n_g, sigma, c->d_colind_j, c->d_perm_j, c->d_jrow, c->n_z, ggx);
  HIPCHK(c, hipGetLastError());
  return MPX_OK;
}

extern "C" int mpx_eval_grad_gamma_device(mpx_ctx* c, int64_t batch, const double* z, const double* p, int p_per_point, const double* lam_g,
                                          const double* sigma, double* grad_gamma_x, double* grad_gamma_p) {
  if (!c) return MPX_ERR_INVALID;
  if (!c->has_device)
    return fail(c, MPX_ERR_NO_DEVICE, "mpx_eval_grad_gamma: context was created without a gfx950 code object; there is no CPU fallback");
  if (batch < 1 || batch > (1 << 30) || !z || (!p && c->n_p > 0) || !lam_g || !sigma) return fail(c, MPX_ERR_INVALID, "mpx_eval_grad_gamma: batch/z/p/lam_g/sigma invalid");
  if (c->shard_world > 1) return fail(c, MPX_ERR_UNSUPPORTED, "mpx_eval_grad_gamma on a context in segment-sharded mode (mpx_shard_setup(ctx, 1, 0) leaves it)");
  if (c->kind == 0 && (c->tile_begin != 0 || c->tile_end != (int64_t)c->tiles.size())) return fail(c, MPX_ERR_UNSUPPORTED, "mpx_eval_grad_gamma with a tile sub-range");
  HIPCHK(c, hipSetDevice(c->device));
  c->wcum_valid = false;
  if (!grad_gamma_x && (!grad_gamma_p || c->n_p == 0)) return MPX_OK;
  const bool generic = mpx_knob(MPX_K_GRADL_GENERIC) != nullptr;
  if (c->kind == 1) return grad_gamma_x ? grad_gamma_generic(c, batch, z, p, p_per_point, lam_g, sigma, grad_gamma_x) : MPX_OK;  // (n_p == 0 there)
  if (generic && grad_gamma_x) {
    int rc = grad_gamma_generic(c, batch, z, p, p_per_point, lam_g, sigma, grad_gamma_x);
    if (rc) return rc;
    grad_gamma_x = nullptr;
    if (!grad_gamma_p) return MPX_OK;
  }
  if (!c->fn_gradl_fin) return fail(c, MPX_ERR_UNSUPPORTED, "mpx_eval_grad_gamma: the code object has no nlp_grad kernels (generated by an older mpopt_amd)");
  for (auto& B : c->buckets)
    if (!B.fn_gradl) return fail(c, MPX_ERR_UNSUPPORTED, "mpx_eval_grad_gamma: the code object has no nlp_grad kernels (generated by an older mpopt_amd)");
  const int64_t n_w = p_per_point ? batch : 1, nt = (int64_t)c->tiles.size();
  int rc;
  if ((rc = reserve_wcum(c, (size_t)(n_w * c->n_p))) || (rc = reserve(c, c->partial, (size_t)(batch * nt * c->nred))) ||
      (rc = reserve(c, c->gl_pnode, (size_t)(batch * c->n_phases * c->S * 2))) ||
      (rc = reserve(c, c->gl_halo, (size_t)(batch * c->n_phases * c->S * (c->nx + c->nu)))))
    return rc;
  if ((rc = launch_prefix(c, p, n_w, p_per_point))) return rc;
  for (auto& B : c->buckets) {
    const PhaseStruct& P = c->ph[B.phase];
    const DegTable& t = c->degs[B.dt];
    MpxGradlArgs A{};
    A.z = z, A.z_stride = c->n_z;
    A.w = p, A.wcum = c->wcum.p, A.w_stride = p_per_point ? c->n_p : 0;
    A.lam_g = lam_g, A.lam_stride = c->n_g, A.sigma = sigma;
    A.gx = grad_gamma_x, A.gx_stride = c->n_z;
    A.halo = c->gl_halo.p, A.pseg = c->gl_pnode.p, A.partial = c->partial.p;
    A.n_tiles_total = (int32_t)nt, A.nred = c->nred, A.B = (int32_t)batch;
    A.tiles = c->d_tiles, A.node_i = B.d_node_i, A.node_sk = B.d_node_sk;
    A.Dmat = t.d_D, A.Cmid = t.d_Cmid, A.tk = t.d_tk, A.Wnode = c->d_Wnode;
    A.inv_dtau = 1.0 / (c->tau1 - c->tau0);
    A.z_off = P.z_off, A.g_off_F = P.g_off_F, A.g_off_C = P.g_off_C, A.g_off_DU = P.g_off_DU, A.g_off_mU = P.g_off_mU;
    A.N = (int32_t)c->N, A.seg_off = B.phase * c->S, A.tile_first = B.tile_first, A.phase = B.phase, A.S = c->S;
    // evaluation points per workgroup: 1 for small batches (every point its own workgroups: latency), 4 from a few thousand workgroups on
    // (MPX_GRADL_BPB=n: A/B; measured at configs[1], B = 4096: profiles/r6_nlp_grad)
    const char* bpb_knob = mpx_knob(MPX_K_GRADL_BPB);
    const int bpb = bpb_knob ? std::max(1, atoi(bpb_knob)) : (batch * B.tile_count >= 8192 ? 4 : 1);
    A.bpb = bpb;
    for (int64_t bf = 0; bf < batch; bf += (int64_t)65535 * bpb) {
      A.b_first = (int32_t)bf;
      A.B = (int32_t)std::min<int64_t>(batch, bf + (int64_t)65535 * bpb);
      const int64_t rows = (A.B - bf + bpb - 1) / bpb;
      if ((rc = launch(c, B.fn_gradl, dim3((unsigned)B.tile_count, (unsigned)rows, 1), dim3(MPX_TILE, 1, 1), &A, sizeof A))) return rc;
    }
  }
  MpxGradlFinArgs F{};
  F.z = z, F.z_stride = c->n_z, F.lam_g = lam_g, F.lam_stride = c->n_g, F.sigma = sigma;
  F.gx = grad_gamma_x, F.gx_stride = c->n_z;
  F.gp = c->n_p ? grad_gamma_p : nullptr, F.gp_stride = c->n_p;
  F.halo = c->gl_halo.p, F.pseg = c->gl_pnode.p, F.partial = c->partial.p;
  F.halo_seg = c->d_gl_halo_seg;
  for (int p_ = 0; p_ <= c->n_phases; ++p_) F.halo_off[p_] = c->gl_halo_off[(size_t)p_];
  F.n_tiles_total = (int32_t)nt, F.nred = c->nred;
  for (int p_ = 0; p_ < c->n_phases; ++p_) {
    const PhaseStruct& P = c->ph[p_];
    F.ph[p_].z_off = P.z_off, F.ph[p_].N = (int32_t)c->N, F.ph[p_].tile_first = P.tile_first, F.ph[p_].tile_count = P.tile_count;
    F.ph[p_].tile_count_h = P.tile_count, F.ph[p_].g_off_TC = P.g_off_TC, F.ph[p_].jac_TC = P.jac_TC;
  }
  F.seg_start = c->d_seg_start, F.S = c->S, F.n_lt = (int32_t)c->lt_col.size();
  F.lt_ptr = c->d_lt_ptr, F.lt_col = c->d_lt_col, F.lt_row = c->d_lt_row, F.lt_coef = c->d_lt_coef;
  for (int64_t bf = 0; bf < batch; bf += 1 << 20) {  // (one workgroup per evaluation point; slices keep gridDim.x small)
    MpxGradlFinArgs Fs = F;
    const int64_t nb = std::min<int64_t>(1 << 20, batch - bf);
    Fs.z += bf * c->n_z, Fs.lam_g += bf * c->n_g, Fs.sigma += bf;
    if (Fs.gx) Fs.gx += bf * c->n_z;
    if (Fs.gp) Fs.gp += bf * c->n_p;
    Fs.halo += bf * c->n_phases * c->S * (c->nx + c->nu), Fs.pseg += bf * c->n_phases * c->S * 2, Fs.partial += bf * nt * c->nred;
    if ((rc = launch(c, c->fn_gradl_fin, dim3((unsigned)nb, 1, 1), dim3(256, 1, 1), &Fs, sizeof Fs))) return rc;
  }
  return MPX_OK;
}

extern "C" int mpx_eval_grad_gamma(mpx_ctx* c, int64_t batch, const double* z, const double* p, int p_per_point, const double* lam_g,
                                   const double* sigma, double* grad_gamma_x, double* grad_gamma_p) {
  if (!c) return MPX_ERR_INVALID;
  if (!c->has_device)
    return fail(c, MPX_ERR_NO_DEVICE, "mpx_eval_grad_gamma: context was created without a gfx950 code object; there is no CPU fallback");
  if (batch < 1 || batch > (1 << 30) || !z || (!p && c->n_p > 0) || !lam_g || !sigma)  // (bounded BEFORE the size products and reserves below)
    return fail(c, MPX_ERR_INVALID, "mpx_eval_grad_gamma: batch/z/p/lam_g/sigma invalid");
  HIPCHK(c, hipSetDevice(c->device));
  const size_t B = (size_t)batch, npv = (size_t)(p_per_point ? batch : 1) * c->n_p;
  int rc;
  if ((rc = reserve(c, c->st_z, B * c->n_z)) || (rc = reserve(c, c->st_p, std::max<size_t>(npv, 1))) || (rc = reserve(c, c->st_lam, B * std::max<int64_t>(c->n_g, 1))) ||
      (rc = reserve(c, c->st_sig, B)))
    return rc;
  if (grad_gamma_x && (rc = reserve(c, c->st_ggx, B * c->n_z))) return rc;
  if (grad_gamma_p && c->n_p && (rc = reserve(c, c->st_ggp, B * c->n_p))) return rc;
  c->last_p.clear();  // (the staged widths are overwritten: the host path's "same p" shortcut must not trust them)
  HIPCHK(c, hipMemcpyAsync(c->st_z.p, z, B * c->n_z * 8, hipMemcpyHostToDevice, c->stream));
  if (npv) HIPCHK(c, hipMemcpyAsync(c->st_p.p, p, npv * 8, hipMemcpyHostToDevice, c->stream));
  if (c->n_g) HIPCHK(c, hipMemcpyAsync(c->st_lam.p, lam_g, B * c->n_g * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->st_sig.p, sigma, B * 8, hipMemcpyHostToDevice, c->stream));
  if ((rc = mpx_eval_grad_gamma_device(c, batch, c->st_z.p, c->st_p.p, p_per_point, c->st_lam.p, c->st_sig.p, grad_gamma_x ? c->st_ggx.p : nullptr,
                                       grad_gamma_p && c->n_p ? c->st_ggp.p : nullptr)))
    return rc;
  if (grad_gamma_x) HIPCHK(c, hipMemcpyAsync(grad_gamma_x, c->st_ggx.p, B * c->n_z * 8, hipMemcpyDeviceToHost, c->stream));
  if (grad_gamma_p && c->n_p) HIPCHK(c, hipMemcpyAsync(grad_gamma_p, c->st_ggp.p, B * c->n_p * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return MPX_OK;
}
