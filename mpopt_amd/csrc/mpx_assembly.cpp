// mpx_assembly.cpp -- assembled contexts (include/mpx.h: mpx_create_assembled).
//
// Host runtime + the gather kernel of the two-stage evaluation used for transcriptions that do not fit
// the tiled node kernels of mpx_kernels.h -- today mpopt_adaptive (reference mpopt.py:2877-3375), where
// segment widths are decision variables and the mid-point residual rows (mpopt.py:3088-3124) couple all
// nodes of a segment.  The structure (which z entries feed a point, which raw values feed an output
// entry) is expanded once on the host side of the boundary (mpopt_amd/assembly.py); this file only
// uploads it, launches the generated point kernels and the gather kernel, and owns the buffers.
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <map>
#include <vector>

#include "mpx_internal.h"

using namespace mpxi;

#ifndef MPX_GATHER_UNROLL
#define MPX_GATHER_UNROLL 4
#endif

namespace {

struct DevSet {
  int32_t n = 0, n_loc = 0, n_cst = 0, n_out = 0, n_jac = 0, n_hess = 0;
  int32_t *loc_toff = nullptr, *loc_idx = nullptr, *mu_toff = nullptr, *mu_idx = nullptr;
  double *loc_coef = nullptr, *cst = nullptr, *mu_coef = nullptr;
};

struct DevGather {
  int32_t long_threshold = MPX_GATHER_LONG;  // rows with more terms are summed by a wavefront (both paths of the pass use it)
  int64_t n_rows = 0, nnz = 0, n_long = 0;
  int64_t *ptr = nullptr, *long_rows = nullptr;
  int32_t* src = nullptr;
  double* coef = nullptr;
};

}  // namespace

// Tables of the fused kernels (mpx_assembly_fused.h) for one pass: first term and term count per row, all terms with their
// sources remapped to positions in V = [raw | z | 1], the rows with 2 .. MPX_GATHER_LONG terms and the longer ones.
struct DevFused {
  int32_t *r_idx = nullptr, *r_nt = nullptr, *idx = nullptr, *multi = nullptr, *mid = nullptr, *longr = nullptr, *m_idx = nullptr;
  double *r_coef = nullptr, *m_coef = nullptr;
  int32_t n_multi = 0, n_mid = 0, n_long = 0;
  uint32_t* m_pack = nullptr;  // ELL table of the multi-term rows packed the same way, same dictionary
  uint32_t* r_pack = nullptr;  // single-term rows packed (MpxFusedArgs::r_pack); n_dict = 0: not representable (an index or the dictionary past 16 bits)
  double* r_dict = nullptr;
  int32_t n_dict = 0;
};

struct mpx_asm_state {
  DevFused ffgj, fhess;
  hipFunction_t fn_fused[3] = {nullptr, nullptr, nullptr};
  // lane-per-evaluation-point kernels (mpx_assembly_lanes.h), pass 0: hess_l (mpx_asml_hes), pass 1: f, g, grad_f, jac_g (mpx_asml_fgj);
  // fn NULL: not in the code object.  *_global: the second kernel of a pass with global rows (n_global > 0), n_sid scratch slots per block.
  struct Lanes {
    hipFunction_t fn = nullptr, fn_global = nullptr;
    int groups = 0, n_global = 0, n_sid = 0;
  } lanes[2];
  DevBuf<double> lane_scratch;
  hipModule_t lanes_module = nullptr;  // the code object attached later with the lane kernels (mpx_assembled_attach_kernels)
  int fuse_nt = 0, fuse_u[2] = {0, 0};  // lanes per workgroup; evaluation points per workgroup pass (first order, Hessian); 0: no kernel
  int fuse_wg[3] = {0, 0, 0};           // resident workgroups per launch (compute units x occupancy)
  long long* dbg = nullptr;             // MPX_FUSE_DEBUG
  int32_t *d_task_ptr[2] = {nullptr, nullptr}, *d_task_list[2] = {nullptr, nullptr};  // point-phase schedules (first order, Hessian)
  // chained local variables (MpxPtSet::chain_v): shared term lists and the per-point slots
  int32_t *d_ch_ptr = nullptr, *d_ch_idx = nullptr, *d_ch_slot = nullptr;
  double* d_ch_coef = nullptr;
  int32_t n_chains = 0;
  std::vector<int32_t*> d_chain_pos;
  // packed tables of the local variables / multipliers (MpxPtSet::loc_pack, mu_pack) and their dictionaries
  std::vector<uint32_t*> d_loc_pack, d_mu_pack;
  double *d_l_dict = nullptr, *d_m_dict = nullptr;
  int32_t n_ldict = 0, n_mdict = 0;
  std::vector<DevSet> sets;
  MpxPtSet* d_sets = nullptr;  // device copy of the per-set argument blocks
  int n_blocks = 0;            // 64-lane blocks of the fused point launch
  int points_per_lane = 1;     // evaluation points a lane of the generated kernels takes together (read from the code object)
  hipFunction_t fn[3] = {nullptr, nullptr, nullptr};
  DevGather fgj, hess;
  int64_t raw_n = 0, rawh_n = 0;
  DevBuf<double> raw;
};

// Rows with at most MPX_GATHER_LONG terms: one lane per row, terms summed in their stored order (the
// first two terms of a row stay in registers over the batch chunk -- most rows are copies or two-term
// sums).  Longer rows (the objective, gradients and Hessian corners of global variables): one wavefront
// per row, lane j sums terms j, j+64, ... in order, then a fixed shuffle tree.  Either way the result of a
// row does not depend on the launch geometry.
__device__ __forceinline__ double gather_value(int32_t k, const double* __restrict__ rb, const double* __restrict__ zb) {
  return k >= 0 ? rb[k] : (k == -1 ? 1.0 : zb[-2 - k]);
}

__device__ __forceinline__ double* gather_out(const MpxGatherArgs& A, int64_t row, int64_t& local, int64_t& stride) {
  // The segment tables are part of the kernel arguments, i.e. scalar registers as long as every index is a constant.  Indexed by the
  // lane's running `sg` (a search loop) they were fetched from the argument block in memory instead: up to three dependent round
  // trips in front of the first useful load of a kernel whose whole life is a few microseconds (single evaluations).  The begins are
  // non-decreasing, so the last segment that starts at or before the row is the one the search stopped at.
#ifdef MPX_GATHER_OUT_SEARCH  // (the search of rounds 1 - 4: A/B with MPX_LIB_HIPCC_FLAGS=-DMPX_GATHER_OUT_SEARCH)
  int sg = 0;
  while (sg + 1 < A.n_seg && row >= A.seg_begin[sg + 1]) ++sg;
  local = row - A.seg_begin[sg];
  stride = A.seg_stride[sg];
  return A.seg_out[sg];
#endif
  static_assert(sizeof(A.seg_out) / sizeof(A.seg_out[0]) == 4, "gather_out: four output segments");
  int64_t beg = A.seg_begin[0];
  double* out = A.seg_out[0];
  stride = A.seg_stride[0];
#pragma unroll
  for (int k = 1; k < 4; ++k) {
    const bool in = k < A.n_seg && row >= A.seg_begin[k];
    beg = in ? A.seg_begin[k] : beg;
    stride = in ? A.seg_stride[k] : stride;
    out = in ? A.seg_out[k] : out;
  }
  local = row - beg;
  return out;
}

template <int UN>
__global__ __launch_bounds__(256) void mpx_gather_kernel(const MpxGatherArgs A) {
  const int b0 = blockIdx.y * A.b_per_block;
  const int b1 = (b0 + A.b_per_block < A.B) ? b0 + A.b_per_block : A.B;
  // Workgroups go to the 8 XCDs round-robin by linear id.  With gridDim.x a multiple of 8 every XCD would
  // own the same row blocks for ALL evaluation points -- measured 5.3x slower on MI355X (12.4 -> 2.3 M
  // evals/s with one empty block appended to a 31-block grid) -- so the row block rotates with the point.
  const int bx = (int)((blockIdx.x + blockIdx.y) % gridDim.x);
  if (bx >= A.n_short_blocks) {  // one wavefront per long row
    const int64_t w = (int64_t)(bx - A.n_short_blocks) * 4 + (threadIdx.x >> 6);
    if (w >= A.n_long) return;
    const int64_t row = A.long_rows[w];
    int64_t local, stride;
    double* out = gather_out(A, row, local, stride);
    if (!out) return;
    const int lane = threadIdx.x & 63;
    const int64_t e0 = A.ptr[row], e1 = A.ptr[row + 1];
    for (int b = b0; b < b1; ++b) {
      const double* __restrict__ rb = A.raw + (int64_t)b * A.raw_stride;
      const double* __restrict__ zb = A.z + (int64_t)b * A.z_stride;
      double s = 0;
      for (int64_t e = e0 + lane; e < e1; e += 64) s = fma(A.coef[e], gather_value(A.src[e], rb, zb), s);
      s = mpx_wave_total(s);
      if (lane == 0) out[(int64_t)b * stride + local] = s;
    }
    return;
  }
  const int64_t row = (int64_t)bx * blockDim.x + threadIdx.x;
  if (row >= A.n_rows) return;
  const int64_t e0 = A.ptr[row], e1 = A.ptr[row + 1];
  const int nt = (int)(e1 - e0);
  if (nt > A.long_threshold) return;
  int64_t local, stride;
  double* out = gather_out(A, row, local, stride);
  if (!out) return;
  int32_t k0 = -1, k1 = -1;
  double c0 = 0, c1 = 0;
  if (nt >= 1) k0 = A.src[e0], c0 = A.coef[e0];
  if (nt >= 2) k1 = A.src[e0 + 1], c1 = A.coef[e0 + 1];
  int b = b0;
  // The kernel is latency-bound (pointer -> source -> value -> store is a chain of dependent round trips and
  // only 2048 workgroups are resident), so a lane takes UN evaluation points at a time with their
  // loads in flight together.  Per point the terms are added in the same order as in the remainder loop below:
  // results are identical.
  for (; b + UN <= b1; b += UN) {
    const double* __restrict__ rb = A.raw + (int64_t)b * A.raw_stride;
    const double* __restrict__ zb = A.z + (int64_t)b * A.z_stride;
    double s[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) s[u] = 0;
    if (nt >= 1) {
#pragma unroll
      for (int u = 0; u < UN; ++u) s[u] = fma(c0, gather_value(k0, rb + u * A.raw_stride, zb + u * A.z_stride), s[u]);
    }
    if (nt >= 2) {
#pragma unroll
      for (int u = 0; u < UN; ++u) s[u] = fma(c1, gather_value(k1, rb + u * A.raw_stride, zb + u * A.z_stride), s[u]);
    }
    for (int64_t e = e0 + 2; e < e1; ++e) {
      const int32_t k = A.src[e];
      const double cf = A.coef[e];
#pragma unroll
      for (int u = 0; u < UN; ++u) s[u] = fma(cf, gather_value(k, rb + u * A.raw_stride, zb + u * A.z_stride), s[u]);
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) out[(int64_t)(b + u) * stride + local] = s[u];
  }
  for (; b < b1; ++b) {
    const double* __restrict__ rb = A.raw + (int64_t)b * A.raw_stride;
    const double* __restrict__ zb = A.z + (int64_t)b * A.z_stride;
    double s = 0;
    if (nt >= 1) s = fma(c0, gather_value(k0, rb, zb), s);
    if (nt >= 2) s = fma(c1, gather_value(k1, rb, zb), s);
    for (int64_t e = e0 + 2; e < e1; ++e) s = fma(A.coef[e], gather_value(A.src[e], rb, zb), s);
    out[(int64_t)b * stride + local] = s;
  }
}

namespace {

template <class T>
int upload_n(mpx_ctx* c, T** dst, const T* src, size_t n) {
  HIPCHK(c, dev_malloc((void**)dst, std::max<size_t>(n, 1) * sizeof(T)));
  if (n) HIPCHK(c, hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
  else HIPCHK(c, hipMemset(*dst, 0, sizeof(T)));  // (an empty list: one zeroed element, never uninitialised memory)
  return MPX_OK;
}

int upload_gather(mpx_ctx* c, DevGather& d, const mpx_gather& g, int64_t raw_n, int64_t n_z, const char* what, int long_threshold) {
  d.long_threshold = long_threshold;
  if (g.n_rows < 0 || (g.n_rows && (!g.ptr || g.ptr[0] != 0))) return fail(c, MPX_ERR_INVALID, "%s: bad row pointers", what);
  d.n_rows = g.n_rows;
  d.nnz = g.n_rows ? g.ptr[g.n_rows] : 0;
  for (int64_t r = 0; r < g.n_rows; ++r)
    if (g.ptr[r + 1] < g.ptr[r]) return fail(c, MPX_ERR_INVALID, "%s: row pointers decrease", what);
  for (int64_t e = 0; e < d.nnz; ++e) {
    const int64_t k = g.src[e];
    if (k >= raw_n || (k <= -2 && -2 - k >= n_z)) return fail(c, MPX_ERR_INVALID, "%s: source %lld out of range", what, (long long)k);
  }
  int rc;
  std::vector<int64_t> one(1, 0);
  if ((rc = upload_n(c, &d.ptr, g.n_rows ? g.ptr : one.data(), (size_t)g.n_rows + 1))) return rc;
  if ((rc = upload_n(c, &d.src, g.src, (size_t)d.nnz))) return rc;
  std::vector<int64_t> lr;
  for (int64_t r = 0; r < g.n_rows; ++r)
    if (g.ptr[r + 1] - g.ptr[r] > d.long_threshold) lr.push_back(r);
  d.n_long = (int64_t)lr.size();
  if ((rc = upload(c, &d.long_rows, lr))) return rc;
  return upload_n(c, &d.coef, g.coef, (size_t)d.nnz);
}

int upload_fused(mpx_ctx* c, DevFused& f, const mpx_gather& g, int64_t raw_n, int64_t n_z, int mt) {
  const int64_t nnz = g.n_rows ? g.ptr[g.n_rows] : 0;
  std::vector<int32_t> r_idx((size_t)std::max<int64_t>(g.n_rows, 1), 0), r_nt((size_t)std::max<int64_t>(g.n_rows, 1), 0), idx((size_t)std::max<int64_t>(nnz, 1), 0), multi, mid, longr;
  std::vector<double> r_coef((size_t)std::max<int64_t>(g.n_rows, 1), 0.0);
  auto pos = [&](int64_t k) -> int32_t { return (int32_t)(k >= 0 ? k : (k == -1 ? raw_n + n_z : raw_n + (-2 - k))); };
  for (int64_t e = 0; e < nnz; ++e) idx[(size_t)e] = pos(g.src[e]);
  for (int64_t r = 0; r < g.n_rows; ++r) {
    const int64_t nt = g.ptr[r + 1] - g.ptr[r];
    r_nt[(size_t)r] = (int32_t)nt;
    r_idx[(size_t)r] = nt ? idx[(size_t)g.ptr[r]] : (int32_t)(raw_n + n_z);
    r_coef[(size_t)r] = nt ? g.coef[g.ptr[r]] : 0.0;
    if (nt > mt) longr.push_back((int32_t)r);  // (mt = the pass's long-row threshold, DevGather::long_threshold)
    else if (nt >= 2) multi.push_back((int32_t)r);
  }
  f.n_multi = (int32_t)multi.size(), f.n_mid = (int32_t)mid.size(), f.n_long = (int32_t)longr.size();
  // ELL copy of the multi-term rows: [t][row], padded with (the 1.0 slot, 0) -- padding terms are never added (t < nt)
  std::vector<int32_t> m_idx((size_t)std::max<int64_t>((int64_t)mt * f.n_multi, 1), (int32_t)(raw_n + n_z));
  std::vector<double> m_coef((size_t)std::max<int64_t>((int64_t)mt * f.n_multi, 1), 0.0);
  for (int32_t m = 0; m < f.n_multi; ++m)
    for (int64_t e = g.ptr[multi[m]], t = 0; e < g.ptr[multi[m] + 1]; ++e, ++t) m_idx[(size_t)(t * f.n_multi + m)] = idx[(size_t)e], m_coef[(size_t)(t * f.n_multi + m)] = g.coef[e];
  // packed form of the rows with at most one term and of the ELL table of the multi-term rows: position | code << 16, one dictionary
  // of the distinct coefficients (bit patterns) for both
  std::vector<uint32_t> r_pack((size_t)std::max<int64_t>(g.n_rows, 1), 0xffffffffu), m_pack(m_idx.size(), 0u);
  std::vector<double> r_dict;
  {
    std::map<uint64_t, uint32_t> code_of;
    bool ok = raw_n + n_z + 1 <= 65536;
    auto code = [&](double v) -> uint32_t {
      uint64_t bits;
      memcpy(&bits, &v, 8);
      auto it = code_of.find(bits);
      if (it == code_of.end()) {
        ok = ok && r_dict.size() < 65535;
        it = code_of.emplace(bits, (uint32_t)r_dict.size()).first;
        r_dict.push_back(v);
      }
      return it->second;
    };
    for (int64_t r = 0; r < g.n_rows && ok; ++r)
      if (r_nt[(size_t)r] <= 1) r_pack[(size_t)r] = (uint32_t)r_idx[(size_t)r] | (code(r_coef[(size_t)r]) << 16);
    for (int32_t m = 0; m < f.n_multi && ok; ++m)
      for (int64_t e = g.ptr[multi[m]], t = 0; e < g.ptr[multi[m] + 1]; ++e, ++t)
        m_pack[(size_t)(t * f.n_multi + m)] = (uint32_t)idx[(size_t)e] | (code(g.coef[e]) << 16);
    f.n_dict = ok ? (int32_t)r_dict.size() : 0;
  }
  int rc;
  if ((rc = upload(c, &f.r_pack, r_pack)) || (rc = upload(c, &f.r_dict, r_dict)) || (rc = upload(c, &f.m_pack, m_pack))) return rc;
  if ((rc = upload(c, &f.r_idx, r_idx)) || (rc = upload(c, &f.r_nt, r_nt)) || (rc = upload(c, &f.r_coef, r_coef)) || (rc = upload(c, &f.idx, idx)) ||
      (rc = upload(c, &f.multi, multi)) || (rc = upload(c, &f.mid, mid)) || (rc = upload(c, &f.longr, longr)) || (rc = upload(c, &f.m_idx, m_idx)) ||
      (rc = upload(c, &f.m_coef, m_coef)))
    return rc;
  return MPX_OK;
}

int check_terms(mpx_ctx* c, const int32_t* nterm, int nv, const int32_t* idx, int64_t n, int64_t limit, std::vector<int32_t>& toff, const char* what) {
  toff.assign((size_t)nv + 1, 0);
  for (int v = 0; v < nv; ++v) {
    if (nterm[v] < 0) return fail(c, MPX_ERR_INVALID, "%s: negative term count", what);
    toff[v + 1] = toff[v] + nterm[v];
  }
  const int64_t tot = (int64_t)toff[nv] * n;
  for (int64_t e = 0; e < tot; ++e)
    if (idx[e] < 0 || idx[e] >= limit) return fail(c, MPX_ERR_INVALID, "%s: index %d out of range", what, idx[e]);
  return MPX_OK;
}

}  // namespace

void mpx_asm_release(mpx_ctx* c) {
  mpx_asm_state* a = c->assembled;
  if (!a) return;
  auto fr = [](void* p) {
    if (p) (void)dev_free(p);
  };
  for (auto& s : a->sets) fr(s.loc_toff), fr(s.loc_idx), fr(s.mu_toff), fr(s.mu_idx), fr(s.loc_coef), fr(s.cst), fr(s.mu_coef);
  for (DevGather* g : {&a->fgj, &a->hess}) fr(g->ptr), fr(g->src), fr(g->coef), fr(g->long_rows);
  for (DevFused* f : {&a->ffgj, &a->fhess}) fr(f->r_idx), fr(f->r_nt), fr(f->idx), fr(f->multi), fr(f->mid), fr(f->longr), fr(f->m_idx), fr(f->r_coef), fr(f->m_coef), fr(f->r_pack), fr(f->r_dict), fr(f->m_pack);
  fr(a->d_ch_ptr), fr(a->d_ch_idx), fr(a->d_ch_slot), fr(a->d_ch_coef);
  for (auto q : a->d_chain_pos) fr(q);
  for (auto q : a->d_loc_pack) fr(q);
  for (auto q : a->d_mu_pack) fr(q);
  fr(a->d_l_dict), fr(a->d_m_dict);
  fr(a->lane_scratch.p);
  if (a->lanes_module) (void)hipModuleUnload(a->lanes_module);
  fr(a->raw.p), fr(a->d_sets), fr(a->d_task_ptr[0]), fr(a->d_task_ptr[1]), fr(a->d_task_list[0]), fr(a->d_task_list[1]);
  if (a->dbg) (void)hipHostFree(a->dbg);
  delete a;
  c->assembled = nullptr;
}

// The lane-per-point kernels of a code object (generated when the point tasks fall into groups: mpopt_amd/assembly_lanes.py), either the
// context's own or one attached later (mpx_assembled_attach_kernels): mpx_asml_<pass>_info = {groups, tile doubles, nnz of the pass's
// pattern (a check), global rows, scratch slots}.
// (assembly_lanes.py: pattern_hash -- the same arithmetic)
static int lane_pattern_hash(const std::vector<int32_t>& row, const std::vector<int32_t>& col, int64_t n_z, int64_t n_g) {
  uint64_t s = 0;
  for (size_t i = 0; i < row.size(); ++i)
    s += (((uint64_t)(uint32_t)row[i] * 73856093u) & 0xFFFFFFFFu) ^ (((uint64_t)(uint32_t)col[i] * 19349663u) & 0xFFFFFFFFu) ^ (((uint64_t)i * 83492791u) & 0xFFFFFFFFu);
  const uint64_t h = (s + 2654435761ull * ((uint64_t)n_z & 0xFFFFFFFFu) + 40503ull * ((uint64_t)n_g & 0xFFFFFFFFu)) & 0xFFFFFFFFu;
  return (int)(h & 0x7FFFFFFFu);
}

static void load_lanes(mpx_ctx* c, mpx_asm_state* a, hipModule_t mod, int64_t nnz_hess, int64_t nnz_jac) {
  const int want_hash[2] = {lane_pattern_hash(c->hrow, c->hcol, c->n_z, c->n_g), lane_pattern_hash(c->jrow, c->jcol, c->n_z, c->n_g)};
  for (int ps = 0; ps < 2; ++ps) {
    static const char* iname[2] = {"mpx_asml_hes_info", "mpx_asml_fgj_info"};
    static const char* kname2[2] = {"mpx_asml_hes", "mpx_asml_fgj"};
    static const char* gname[2] = {"mpx_asml_hes_global", "mpx_asml_fgj_global"};
    int li[6] = {0, 0, 0, 0, 0, 0};
    hipFunction_t fl = nullptr, fg = nullptr;
    hipDeviceptr_t sym = nullptr;
    size_t bytes = 0;
    if (hipModuleGetGlobal(&sym, &bytes, mod, iname[ps]) == hipSuccess && bytes == sizeof li && hipMemcpyDtoH(li, sym, sizeof li) == hipSuccess &&
        li[0] >= 1 && li[2] == (int)(ps == 0 ? nnz_hess : nnz_jac) && li[5] == want_hash[ps] && hipModuleGetFunction(&fl, mod, kname2[ps]) == hipSuccess &&
        (li[3] == 0 || hipModuleGetFunction(&fg, mod, gname[ps]) == hipSuccess))
      a->lanes[ps].fn = fl, a->lanes[ps].fn_global = fg, a->lanes[ps].groups = li[0], a->lanes[ps].n_global = li[3], a->lanes[ps].n_sid = li[4];
  }
  (void)hipGetLastError();
}

extern "C" int mpx_assembled_attach_kernels(mpx_ctx* c, const void* code_object, size_t size) {
  if (!c || c->kind != 1 || !code_object || !size) return fail(c, MPX_ERR_INVALID, "mpx_assembled_attach_kernels: an assembled context and a code object");
  if (!c->has_device || !c->assembled) return fail(c, MPX_ERR_UNSUPPORTED, "mpx_assembled_attach_kernels: the context has no device");
  mpx_asm_state* a = c->assembled;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (a->lanes_module) {  // (replaces what was attached before)
    a->lanes[0] = a->lanes[1] = mpx_asm_state::Lanes();
    (void)hipModuleUnload(a->lanes_module);
    a->lanes_module = nullptr;
  }
  if (hipModuleLoadData(&a->lanes_module, code_object) != hipSuccess) {
    a->lanes_module = nullptr;
    return fail(c, MPX_ERR_HIP, "mpx_assembled_attach_kernels: hipModuleLoadData failed (is the code object built for this GPU?)");
  }
  load_lanes(c, a, a->lanes_module, c->nnz_h, c->nnz_j);
  return MPX_OK;
}

extern "C" int mpx_create_assembled(const mpx_assembly* D, mpx_ctx** out) {
  if (!D || !out) return fail(nullptr, MPX_ERR_INVALID, "null argument");
  *out = nullptr;
  if (D->version != MPX_VERSION) return fail(nullptr, MPX_ERR_INVALID, "mpx_assembly.version %d != %d", D->version, MPX_VERSION);
  if (D->n_z < 1 || D->n_g < 0 || D->nnz_jac < 0 || D->nnz_hess < 0 || D->n_sets < 1 || !D->sets)
    return fail(nullptr, MPX_ERR_INVALID, "bad dimensions");
  if (D->n_z >= (1LL << 30) || D->n_g >= (1LL << 30)) return fail(nullptr, MPX_ERR_UNSUPPORTED, "problem too large for int32 indices");
  if (D->fgj.n_rows != 1 + D->n_g + D->n_z + D->nnz_jac || D->hess.n_rows != D->nnz_hess)
    return fail(nullptr, MPX_ERR_INVALID, "gather row counts do not match the sizes");
  if ((D->nnz_jac && (!D->jac_row || !D->jac_col)) || (D->nnz_hess && (!D->hess_row || !D->hess_col)))
    return fail(nullptr, MPX_ERR_INVALID, "missing patterns");
  mpx_ctx* c = new (std::nothrow) mpx_ctx;
  if (!c) return fail(nullptr, MPX_ERR_ALLOC, "out of memory");
  c->kind = 1;
  c->device = D->device;
  c->n_z = D->n_z;
  c->n_g = D->n_g;
  c->n_p = 0;
  c->nnz_j = D->nnz_jac;
  c->nnz_h = D->nnz_hess;
  c->jrow.assign(D->jac_row, D->jac_row + D->nnz_jac);
  c->jcol.assign(D->jac_col, D->jac_col + D->nnz_jac);
  c->hrow.assign(D->hess_row, D->hess_row + D->nnz_hess);
  c->hcol.assign(D->hess_col, D->hess_col + D->nnz_hess);
  int rc = MPX_OK;
  for (int64_t k = 0; k < D->nnz_jac && !rc; ++k)
    if (c->jrow[k] < 0 || c->jrow[k] >= D->n_g || c->jcol[k] < 0 || c->jcol[k] >= D->n_z) rc = fail(c, MPX_ERR_INVALID, "jac pattern out of range");
  for (int64_t k = 0; k < D->nnz_hess && !rc; ++k)
    if (c->hrow[k] < 0 || c->hrow[k] > c->hcol[k] || c->hcol[k] >= D->n_z) rc = fail(c, MPX_ERR_INVALID, "hess pattern must be upper triangular and in range");
  auto bail = [&](int code) {
    create_error() = c->err;
    mpx_destroy(c);
    return code;
  };
  if (rc) return bail(rc);
  if (!D->code_object) {  // structure-only context (patterns and sizes); evaluation fails loudly
    *out = c;
    return MPX_OK;
  }
  if (hipSetDevice(c->device) != hipSuccess) return bail(fail(c, MPX_ERR_NO_DEVICE, "no usable HIP device %d", c->device));
  if (hipModuleLoadData(&c->module, D->code_object) != hipSuccess) return bail(fail(c, MPX_ERR_HIP, "hipModuleLoadData failed (is the code object built for this GPU?)"));
  mpx_asm_state* a = c->assembled = new mpx_asm_state;
  a->sets.resize(D->n_sets);
  std::vector<MpxPtSet> hs(D->n_sets);
  for (int k = 0; k < D->n_sets; ++k) {
    const mpx_point_set& S = D->sets[k];
    DevSet& d = a->sets[k];
    if (S.n_points < 1 || S.n_loc < 0 || S.n_cst < 0 || S.n_out < 0 || S.n_jac < 0 || S.n_hess < 0 || S.fid < 0) return bail(fail(c, MPX_ERR_INVALID, "point set %d: bad sizes", k));
    d.n = S.n_points, d.n_loc = S.n_loc, d.n_cst = S.n_cst, d.n_out = S.n_out, d.n_jac = S.n_jac, d.n_hess = S.n_hess;
    std::vector<int32_t> toff, moff;
    if ((rc = check_terms(c, S.loc_nterm, S.n_loc, S.loc_idx, S.n_points, D->n_z, toff, "local variables"))) return bail(rc);
    if ((rc = check_terms(c, S.mu_nterm, S.n_out, S.mu_idx, S.n_points, D->n_g + 1, moff, "multipliers"))) return bail(rc);
    if ((rc = upload(c, &d.loc_toff, toff)) || (rc = upload(c, &d.mu_toff, moff))) return bail(rc);
    if ((rc = upload_n(c, &d.loc_idx, S.loc_idx, (size_t)toff.back() * d.n)) || (rc = upload_n(c, &d.loc_coef, S.loc_coef, (size_t)toff.back() * d.n)) ||
        (rc = upload_n(c, &d.mu_idx, S.mu_idx, (size_t)moff.back() * d.n)) || (rc = upload_n(c, &d.mu_coef, S.mu_coef, (size_t)moff.back() * d.n)) ||
        (rc = upload_n(c, &d.cst, S.cst, (size_t)S.n_cst * d.n)))
      return bail(rc);
    MpxPtSet& h = hs[k];
    h.n = d.n, h.fid = S.fid, h.block_first = a->n_blocks, h.n_slots_fgj = d.n_out + d.n_jac, h.n_hess = d.n_hess;
    h.loc_toff = d.loc_toff, h.loc_idx = d.loc_idx, h.loc_coef = d.loc_coef, h.cst = d.cst;
    h.mu_toff = d.mu_toff, h.mu_idx = d.mu_idx, h.mu_coef = d.mu_coef;
    h.raw_off = a->raw_n, h.rawh_off = a->rawh_n;
    h.n_loc = d.n_loc, h.n_out = d.n_out, h.chain_v = -1, h.chain_pos = nullptr, h.loc_pack = nullptr, h.mu_pack = nullptr;
    a->n_blocks += (d.n + 63) / 64;
    a->raw_n += (int64_t)d.n * (d.n_out + d.n_jac);
    a->rawh_n += (int64_t)d.n * d.n_hess;
  }
  {  // packed copies of the local-variable and multiplier tables of every set (index | code << 16) with context-wide dictionaries
    std::vector<double> ldict, mdict;
    std::map<uint64_t, uint32_t> lcode, mcode;
    bool lok = D->n_z + 1 < 65536, mok = D->n_g + 1 < 65536;
    auto code = [](std::map<uint64_t, uint32_t>& m, std::vector<double>& dict, double v, bool& ok) -> uint32_t {
      uint64_t bits;
      memcpy(&bits, &v, 8);
      auto it = m.find(bits);
      if (it == m.end()) {
        ok = ok && dict.size() < 65535;
        it = m.emplace(bits, (uint32_t)dict.size()).first;
        dict.push_back(v);
      }
      return it->second;
    };
    a->d_loc_pack.assign(D->n_sets, nullptr), a->d_mu_pack.assign(D->n_sets, nullptr);
    for (int k = 0; k < D->n_sets; ++k) {
      const mpx_point_set& S = D->sets[k];
      int64_t lt = 0, mt_ = 0;
      for (int v = 0; v < S.n_loc; ++v) lt += S.loc_nterm[v];
      for (int r = 0; r < S.n_out; ++r) mt_ += S.mu_nterm[r];
      std::vector<uint32_t> lp((size_t)std::max<int64_t>(lt * S.n_points, 1), 0u), mp((size_t)std::max<int64_t>(mt_ * S.n_points, 1), 0u);
      for (int64_t e = 0; e < lt * S.n_points; ++e) lp[(size_t)e] = (uint32_t)S.loc_idx[e] | (code(lcode, ldict, S.loc_coef[e], lok) << 16);
      for (int64_t e = 0; e < mt_ * S.n_points; ++e) mp[(size_t)e] = (uint32_t)S.mu_idx[e] | (code(mcode, mdict, S.mu_coef[e], mok) << 16);
      if ((rc = upload(c, &a->d_loc_pack[k], lp)) || (rc = upload(c, &a->d_mu_pack[k], mp))) return bail(rc);
      hs[k].loc_pack = a->d_loc_pack[k], hs[k].mu_pack = a->d_mu_pack[k];
    }
    a->n_ldict = lok ? (int32_t)ldict.size() : 0, a->n_mdict = mok ? (int32_t)mdict.size() : 0;
    if (ldict.empty()) ldict.push_back(0.0);
    if (mdict.empty()) mdict.push_back(0.0);
    if ((rc = upload(c, &a->d_l_dict, ldict)) || (rc = upload(c, &a->d_m_dict, mdict))) return bail(rc);
  }
  {  // Chained local variables: variable v of set k qualifies when, over its points, the (non-padding) term lists are all prefixes
     // of the longest one -- same z indices, same coefficients, same order.  Chains with identical term lists are shared
     // between sets (the running width sum of the node set and of the mid-point set of a phase).  At most one variable per set,
     // the one with the most table terms; total slots bounded by the LDS array of the kernels (256).
    std::vector<std::vector<std::pair<int32_t, double>>> chains;
    std::vector<int32_t> ch_slot;
    int32_t slots = 0;
    a->d_chain_pos.assign(D->n_sets, nullptr);
    for (int k = 0; k < D->n_sets; ++k) {
      const mpx_point_set& S = D->sets[k];
      hs[k].chain_v = -1, hs[k].chain_pos = nullptr;
      const int64_t n = S.n_points;
      int best_v = -1, best_T = 3;  // (not worth it below four terms)
      std::vector<std::pair<int32_t, double>> best_list;
      std::vector<int32_t> best_len;
      int64_t toff = 0;
      for (int v = 0; v < S.n_loc; toff += S.loc_nterm[v], ++v) {
        const int T = S.loc_nterm[v];
        if (T <= best_T) continue;
        std::vector<int32_t> len((size_t)n, 0);
        int64_t longest = 0;
        bool ok = true;
        for (int64_t p = 0; p < n && ok; ++p) {
          int L = 0;
          while (L < T && S.loc_coef[(toff + L) * n + p] != 0.0) ++L;
          for (int t = L; t < T && ok; ++t) ok = S.loc_coef[(toff + t) * n + p] == 0.0;  // padding trails
          len[(size_t)p] = L;
          if (L > len[(size_t)longest]) longest = p;
        }
        std::vector<std::pair<int32_t, double>> list;
        for (int t = 0; ok && t < len[(size_t)longest]; ++t) list.push_back({S.loc_idx[(toff + t) * n + longest], S.loc_coef[(toff + t) * n + longest]});
        for (int64_t p = 0; p < n && ok; ++p)
          for (int t = 0; t < len[(size_t)p] && ok; ++t)
            ok = S.loc_idx[(toff + t) * n + p] == list[(size_t)t].first && S.loc_coef[(toff + t) * n + p] == list[(size_t)t].second;
        if (ok && !list.empty()) best_v = v, best_T = T, best_list = list, best_len = len;
      }
      if (best_v < 0) continue;
      int cid = -1;
      for (size_t q = 0; q < chains.size(); ++q) {  // an existing chain this list is a prefix of (or equal to)
        if (chains[q].size() >= best_list.size() && std::equal(best_list.begin(), best_list.end(), chains[q].begin())) cid = (int)q;
      }
      if (cid < 0) {
        if (slots + (int32_t)best_list.size() + 1 > 256) continue;
        cid = (int)chains.size();
        chains.push_back(best_list);
        ch_slot.push_back(slots);
        slots += (int32_t)best_list.size() + 1;
      }
      std::vector<int32_t> pos((size_t)n);
      for (int64_t p = 0; p < n; ++p) pos[(size_t)p] = ch_slot[(size_t)cid] + best_len[(size_t)p];
      if ((rc = upload(c, &a->d_chain_pos[k], pos))) return bail(rc);
      hs[k].chain_v = best_v, hs[k].chain_pos = a->d_chain_pos[k];
    }
    std::vector<int32_t> ch_ptr(1, 0), ch_idx;
    std::vector<double> ch_coef;
    for (auto& l : chains) {
      for (auto& t : l) ch_idx.push_back(t.first), ch_coef.push_back(t.second);
      ch_ptr.push_back((int32_t)ch_idx.size());
    }
    a->n_chains = (int32_t)chains.size();
    if ((rc = upload(c, &a->d_ch_ptr, ch_ptr)) || (rc = upload(c, &a->d_ch_idx, ch_idx)) || (rc = upload(c, &a->d_ch_coef, ch_coef)) ||
        (rc = upload(c, &a->d_ch_slot, ch_slot)))
      return bail(rc);
  }
  if ((rc = upload(c, &a->d_sets, hs))) return bail(rc);
  static const char* kname[3] = {"mpx_pts_val", "mpx_pts_jac", "mpx_pts_hes"};
  for (int m = 0; m < 3; ++m)
    if (hipModuleGetFunction(&a->fn[m], c->module, kname[m]) != hipSuccess) return bail(fail(c, MPX_ERR_HIP, "kernel %s missing from the code object", kname[m]));
  {
    hipDeviceptr_t sym = nullptr;
    size_t bytes = 0;
    int v = 1;
    if (hipModuleGetGlobal(&sym, &bytes, c->module, "mpx_pts_points_per_lane") == hipSuccess && bytes == sizeof(int) &&
        hipMemcpyDtoH(&v, sym, sizeof(int)) == hipSuccess && v >= 1 && v <= 64)
      a->points_per_lane = v;
    (void)hipGetLastError();  // an older code object without the symbol is fine: one point per lane
  }
  if (a->raw_n >= (1LL << 31) || a->rawh_n >= (1LL << 31)) return bail(fail(c, MPX_ERR_UNSUPPORTED, "raw buffer too large for int32 sources"));
  {  // fused persistent kernels (mpx_assembly_fused.h): present in code objects generated since round 3
    hipDeviceptr_t sym = nullptr;
    size_t bytes = 0;
    int info[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    static const char* fname[3] = {"mpx_asm_fg", "mpx_asm_fgj", "mpx_asm_hes"};
    const bool have_info = hipModuleGetGlobal(&sym, &bytes, c->module, "mpx_fuse_info") == hipSuccess && bytes == sizeof info &&
                           hipMemcpyDtoH(info, sym, sizeof info) == hipSuccess && info[0] >= 64 && info[0] <= 1024;
    // rows with more terms than the pass's ELL width are summed by a wavefront (lane-strided partial sums + shuffle tree) in BOTH
    // the two-pass gather kernel and the fused kernel -- one definition of every sum, so the two paths agree bit for bit
    const int thr_fgj = have_info && info[3] >= 2 ? info[3] : MPX_GATHER_LONG, thr_hes = have_info && info[4] >= 2 ? info[4] : MPX_GATHER_LONG;
    if ((rc = upload_gather(c, a->fgj, D->fgj, a->raw_n, D->n_z, "fgj gather", thr_fgj)) ||
        (rc = upload_gather(c, a->hess, D->hess, a->rawh_n, D->n_z, "hess gather", thr_hes)))
      return bail(rc);
    if (have_info) {
      bool ok = true;
      for (int m = 0; m < 3 && ok; ++m) ok = hipModuleGetFunction(&a->fn_fused[m], c->module, fname[m]) == hipSuccess;
      for (auto& d : a->sets) ok = ok && d.n_loc < 48 && d.n_out < 48;  // (LDS copies of the offset arrays in the fused kernels)
      ok = ok && a->sets.size() <= 16;
      if (ok && (info[1] > 0 || info[2] > 0)) {
        if ((rc = upload_fused(c, a->ffgj, D->fgj, a->raw_n, D->n_z, thr_fgj)) || (rc = upload_fused(c, a->fhess, D->hess, a->rawh_n, D->n_z, thr_hes))) return bail(rc);
        if ((info[7] > 0 && (a->n_ldict < 1 || a->n_ldict > info[7])) || (info[8] > 0 && (a->n_mdict < 1 || a->n_mdict > info[8])))
          return bail(fail(c, MPX_ERR_INVALID, "fused kernels: dictionaries of the packed local-variable / multiplier tables (%d / %d entries) do not fit the compiled capacity (%d / %d)",
                           a->n_ldict, a->n_mdict, info[7], info[8]));
        // kernels compiled for packed single-term rows: the dictionary the generator counted must be the one built here
        if ((info[5] > 0 && (a->ffgj.n_dict < 1 || a->ffgj.n_dict > info[5])) || (info[6] > 0 && (a->fhess.n_dict < 1 || a->fhess.n_dict > info[6])))
          return bail(fail(c, MPX_ERR_INVALID, "fused kernels: coefficient dictionary of the single-term rows (%d / %d entries) does not fit the compiled capacity (%d / %d)",
                           a->ffgj.n_dict, a->fhess.n_dict, info[5], info[6]));
        a->fuse_nt = info[0], a->fuse_u[0] = info[1], a->fuse_u[1] = info[2];
        // point-phase schedule per pass: tasks (evaluation point u, 64-point block) of one chunk onto the NT / 64 wavefronts, longest
        // first onto the least loaded (cost ~ table terms of the block's set + a constant for the function itself)
        for (int ps = 0; ps < 2; ++ps) {
          const int U = std::max(a->fuse_u[ps], 1), NW = a->fuse_nt / 64;
          std::vector<std::pair<double, int>> tasks;
          for (int u = 0; u < U; ++u)
            for (int k = 0; k < (int)a->sets.size(); ++k) {
              const mpx_point_set& S = D->sets[k];
              double terms = 8;
              for (int v = 0; v < S.n_loc; ++v) terms += S.loc_nterm[v];
              if (ps == 1)
                for (int r = 0; r < S.n_out; ++r) terms += S.mu_nterm[r];
              terms += 0.25 * (ps == 1 ? S.n_hess : S.n_jac + S.n_out);
              for (int blk = 0; blk < (S.n_points + 63) / 64; ++blk) tasks.push_back({terms, u * a->n_blocks + hs[k].block_first + blk});
            }
          std::stable_sort(tasks.begin(), tasks.end(), [](const std::pair<double, int>& x, const std::pair<double, int>& y) { return x.first > y.first; });
          std::vector<double> load((size_t)NW, 0.0);
          std::vector<std::vector<int32_t>> per((size_t)NW);
          for (auto& t : tasks) {
            int w = 0;
            for (int q = 1; q < NW; ++q)
              if (load[q] < load[w]) w = q;
            load[w] += t.first;
            per[w].push_back(t.second);
          }
          std::vector<int32_t> tptr(1, 0), tlist;
          for (int w = 0; w < NW; ++w) {
            tlist.insert(tlist.end(), per[w].begin(), per[w].end());
            tptr.push_back((int32_t)tlist.size());
          }
          if ((rc = upload(c, &a->d_task_ptr[ps], tptr)) || (rc = upload(c, &a->d_task_list[ps], tlist))) return bail(rc);
        }
        int n_cu = 256;
        (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->device);
        for (int m = 0; m < 3; ++m) {
          int per_cu = 1;
          if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, a->fn_fused[m], a->fuse_nt, 0) != hipSuccess || per_cu < 1) per_cu = 1;
          a->fuse_wg[m] = n_cu * per_cu;
        }
      }
    }
    (void)hipGetLastError();
    load_lanes(c, a, c->module, D->nnz_hess, D->nnz_jac);  // (code objects that carry the lane kernels themselves)
    (void)hipGetLastError();
  }
  if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) return bail(fail(c, MPX_ERR_HIP, "hipEventCreate failed"));
  c->has_device = true;
  *out = c;
  return MPX_OK;
}

// Batch elements per workgroup.  Measured on MI355X (tools/adaptive_bench.py + rocprofv3): these kernels are
// latency-bound per lane and a batch loop only serialises the loads of a lane (4 points per workgroup: 1.8x
// slower than 1, 64: 2.5x), so every workgroup takes ONE evaluation point; chunk only past the grid limit.
static int pick_chunk(int64_t batch, int64_t) { return (int)((batch + 65534) / 65535); }

static int launch_points(mpx_ctx* c, int mode, int64_t batch, const double* z, const double* lam, const double* sigma) {
  mpx_asm_state* a = c->assembled;
  if ((mode == MPX_MODE_HESS ? a->rawh_n : a->raw_n) == 0) return MPX_OK;
  MpxPtCall A{};
  A.sets = a->d_sets, A.n_sets = (int32_t)a->sets.size(), A.n_g = (int32_t)c->n_g, A.B = (int32_t)batch;
  A.b_per_block = pick_chunk(batch, a->n_blocks);
  // (threshold measured with tools/r2_asm_ab.sh: 14336 workgroups of the 40x4 hypersensitive case gain 17 % from it, MPX_PTS_MIN_WG overrides)
  static const int64_t min_wg = getenv("MPX_PTS_MIN_WG") ? atoll(getenv("MPX_PTS_MIN_WG")) : 4096;
  if (batch * a->n_blocks >= min_wg) A.b_per_block = std::max(A.b_per_block, a->points_per_lane);  // see point_body
  A.z = z, A.z_stride = c->n_z, A.lam = lam, A.lam_stride = c->n_g, A.sigma = sigma;
  A.raw = a->raw.p;
  A.raw_stride = mode == MPX_MODE_HESS ? a->rawh_n : a->raw_n;
  size_t sz = sizeof(A);
  void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &A, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
  HIPCHK(c, hipModuleLaunchKernel(a->fn[mode], (unsigned)a->n_blocks, (unsigned)((batch + A.b_per_block - 1) / A.b_per_block), 1, 64, 1, 1, 0, c->stream, nullptr, cfg));
  return MPX_OK;
}

static int launch_gather(mpx_ctx* c, const DevGather& g, int64_t batch, const double* z, int64_t raw_stride, int n_seg, const int64_t* seg_begin,
                         double* const* seg_out, const int64_t* seg_stride) {
  if (g.n_rows == 0) return MPX_OK;
  MpxGatherArgs A{};
  A.n_rows = g.n_rows, A.ptr = g.ptr, A.src = g.src, A.coef = g.coef;
  A.raw = c->assembled->raw.p, A.raw_stride = raw_stride, A.z = z, A.z_stride = c->n_z;
  A.n_seg = n_seg;
  for (int k = 0; k < n_seg; ++k) A.seg_begin[k] = seg_begin[k], A.seg_out[k] = seg_out[k], A.seg_stride[k] = seg_stride[k];
  A.seg_begin[n_seg] = g.n_rows;
  A.B = (int32_t)batch;
  A.long_rows = g.long_rows, A.n_long = g.n_long;
  A.long_threshold = g.long_threshold;
  A.n_short_blocks = (int32_t)((g.n_rows + 255) / 256);
  const unsigned gx = (unsigned)(A.n_short_blocks + (g.n_long + 3) / 4);
  A.b_per_block = pick_chunk(batch, gx);
  static const int un = getenv("MPX_GATHER_U") ? atoi(getenv("MPX_GATHER_U")) : MPX_GATHER_UNROLL;
  if (batch * gx >= 16384) A.b_per_block = std::max(A.b_per_block, un);  // several points in flight per lane (see kernel)
  const dim3 grid(gx, (unsigned)((batch + A.b_per_block - 1) / A.b_per_block));
  if (un == 16)
    hipLaunchKernelGGL(mpx_gather_kernel<16>, grid, dim3(256), 0, c->stream, A);
  else if (un == 8)
    hipLaunchKernelGGL(mpx_gather_kernel<8>, grid, dim3(256), 0, c->stream, A);
  else
    hipLaunchKernelGGL(mpx_gather_kernel<4>, grid, dim3(256), 0, c->stream, A);
  HIPCHK(c, hipGetLastError());
  return MPX_OK;
}

// Fused persistent launch (mpx_assembly_fused.h): one kernel per pass, raw values stay in LDS.  Used for batches (every
// workgroup should see several evaluation points for its register-resident row table to pay); single evaluations keep the
// two-pass kernels, whose many small workgroups finish a lone point sooner.  Results are bit-identical either way.
// MPX_NO_FUSE=1 / MPX_FUSE_MIN_BATCH=n (read per call) switch.
static bool use_fused(const mpx_asm_state* a, int mode, int64_t batch) {
  if (!a->fuse_nt || a->fuse_u[mode == MPX_MODE_HESS ? 1 : 0] <= 0 || mpx_knob(MPX_K_NO_FUSE)) return false;
  const char* mb = mpx_knob(MPX_K_FUSE_MIN_BATCH);
  return batch >= (mb ? atoll(mb) : 256);
}

static int launch_fused(mpx_ctx* c, int mode, int64_t batch, const double* z, const double* lam, const double* sigma, double* const* out, const int64_t* stride) {
  mpx_asm_state* a = c->assembled;
  const DevFused& f = mode == MPX_MODE_HESS ? a->fhess : a->ffgj;
  const DevGather& g = mode == MPX_MODE_HESS ? a->hess : a->fgj;
  MpxFusedArgs A{};
  A.sets = a->d_sets, A.n_sets = (int32_t)a->sets.size(), A.n_blocks = a->n_blocks, A.n_g = (int32_t)c->n_g, A.B = (int32_t)batch;
  A.z = z, A.z_stride = c->n_z, A.lam = lam, A.lam_stride = c->n_g, A.sigma = sigma;
  A.r_idx = f.r_idx, A.r_coef = f.r_coef, A.r_nt = f.r_nt, A.ptr = g.ptr, A.idx = f.idx, A.coef = g.coef;
  A.r_pack = f.r_pack, A.r_dict = f.r_dict, A.n_dict = f.n_dict, A.m_pack = f.m_pack;
  A.l_dict = a->d_l_dict, A.m_dict = a->d_m_dict, A.n_ldict = a->n_ldict, A.n_mdict = a->n_mdict;
  A.multi_rows = f.multi, A.m_idx = f.m_idx, A.m_coef = f.m_coef, A.mid_rows = f.mid, A.long_rows = f.longr;
  A.n_multi = f.n_multi, A.n_mid = f.n_mid, A.n_long = f.n_long;
  A.task_ptr = a->d_task_ptr[mode == MPX_MODE_HESS ? 1 : 0], A.task_list = a->d_task_list[mode == MPX_MODE_HESS ? 1 : 0];
  A.ch_ptr = a->d_ch_ptr, A.ch_idx = a->d_ch_idx, A.ch_coef = a->d_ch_coef, A.ch_slot = a->d_ch_slot, A.n_chains = a->n_chains;
  for (int k = 0; k < 4; ++k) A.out[k] = out[k], A.out_stride[k] = stride[k];
  const int U = a->fuse_u[mode == MPX_MODE_HESS ? 1 : 0];
  const int64_t chunks = (batch + U - 1) / U;
  static const int wg_env = getenv("MPX_FUSE_WG") ? atoi(getenv("MPX_FUSE_WG")) : 0;
  const unsigned grid = (unsigned)std::min<int64_t>(chunks, wg_env > 0 ? wg_env : a->fuse_wg[mode]);
  static const bool dbg_on = getenv("MPX_FUSE_DEBUG") != nullptr;
  if (dbg_on && !a->dbg) HIPCHK(c, hipHostMalloc((void**)&a->dbg, 256, hipHostMallocMapped));
  A.dbg = dbg_on ? a->dbg : nullptr;
  size_t sz = sizeof(A);
  void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &A, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
  HIPCHK(c, hipModuleLaunchKernel(a->fn_fused[mode], grid, 1, 1, (unsigned)a->fuse_nt, 1, 1, 0, c->stream, nullptr, cfg));
  if (dbg_on) {  // phase stamps of workgroup 1, third chunk (us): z->LDS, points, barrier, single rows, multi rows, mid rows, long rows, barrier
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const long long* d = a->dbg;
    fprintf(stderr, "fused mode %d grid %u: z %.2f | points %.2f | wait %.2f | rows1 %.2f | multi %.2f | mid %.2f | long %.2f | wait %.2f  (chunk %.2f us)\n", mode, grid,
            (d[1] - d[0]) / 100.0, (d[2] - d[1]) / 100.0, (d[3] - d[2]) / 100.0, (d[4] - d[3]) / 100.0, (d[5] - d[4]) / 100.0, (d[6] - d[5]) / 100.0, (d[7] - d[6]) / 100.0,
            (d[8] - d[7]) / 100.0, (d[8] - d[0]) / 100.0);
    if (mpx_knob(MPX_K_FUSE_PT_STAMPS))
      fprintf(stderr, "  point task of wave 0: start +%.2f | gather %.2f | function %.2f | raw stores %.2f\n", (d[16] - d[1]) / 100.0, (d[17] - d[16]) / 100.0,
              (d[18] - d[17]) / 100.0, (d[19] - d[18]) / 100.0);
  }
  return MPX_OK;
}

// Lane-per-evaluation-point kernels (mpx_assembly_lanes.h): one wavefront per (group of point tasks, block of 64 evaluation points);
// the groups of a block sit on one XCD; a second small kernel for the pass's global rows, if it has any (raw values through a scratch
// array).  Bit-identical to the other two paths, so the host picks by batch size (hess_l: from 64 points on); MPX_NO_LANES=1 /
// MPX_LANES_MIN_BATCH=n (read per call) switch.
static bool use_lanes(const mpx_asm_state* a, int ps, int64_t batch) {
  if (!a->lanes[ps].fn || mpx_knob(MPX_K_NO_LANES)) return false;
  const char* mb = mpx_knob(MPX_K_LANES_MIN_BATCH);
  // (hess_l: faster than the other two paths from one block of 64 points on -- 6.3 against 11.6 us at B = 64, profiles/r5_lanes/ab9;
  // the opt-in first-order pass keeps 512)
  return batch >= std::max<int64_t>(mb ? atoll(mb) : (ps == 0 ? 64 : 512), 64);
}

static int launch_lanes(mpx_ctx* c, int ps, int64_t batch, const double* z, const double* lam, const double* sigma, double* const* out) {
  mpx_asm_state* a = c->assembled;
  const mpx_asm_state::Lanes& L = a->lanes[ps];
  const int64_t n_blocks = (batch + 63) / 64;  // (the last block of a ragged batch starts at batch - 64)
  // (the last block of a ragged batch repeats a few points of its neighbour: two wavefronts store the SAME values to the same
  // addresses -- the lane kernels' outputs must stay pure overwrites, never accumulations)
  if (batch < 64 || batch > INT32_MAX - 64 || 8 * (int64_t)L.groups * ((n_blocks + 7) / 8) > INT32_MAX)
    return fail(c, MPX_ERR_UNSUPPORTED, "lane kernels: a batch of %lld evaluation points x %d groups does not fit one launch", (long long)batch, L.groups);
  int rc;
  if (L.n_sid > 0 && (rc = reserve(c, a->lane_scratch, (size_t)(n_blocks * L.n_sid * 64)))) return rc;
  MpxLaneArgs A{};
  A.z = z, A.lam = lam, A.sigma = sigma;
  for (int k = 0; k < 4; ++k) A.out[k] = out[k];
  A.scratch = L.n_sid > 0 ? a->lane_scratch.p : nullptr;
  A.B = (int32_t)batch, A.n_blocks = (int32_t)n_blocks;
  {
    const char* ord = mpx_knob(MPX_K_LANES_ORDER);  // (A/B)
    A.order = ord ? atoi(ord) : (ps == 1 ? 1 : 0);
  }
  const unsigned grid = (unsigned)(8 * (int64_t)L.groups * ((n_blocks + 7) / 8));
  size_t sz = sizeof(A);
  void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &A, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
  HIPCHK(c, hipModuleLaunchKernel(L.fn, grid, 1, 1, 64, 1, 1, 0, c->stream, nullptr, cfg));
  if (L.n_global > 0) HIPCHK(c, hipModuleLaunchKernel(L.fn_global, (unsigned)n_blocks, (unsigned)L.n_global, 1, 64, 1, 1, 0, c->stream, nullptr, cfg));
  return MPX_OK;
}

// Evaluation points per pass.  The raw point values are written by the point kernels and read back by the gather pass; while a
// pass's raw values + outputs fit the 256 MB Infinity Cache of an MI355X the read-back does not go to HBM.  Measured
// (tools/adaptive_batch_sweep.py, moon lander 20x5): f+g+grad_f+jac_g peaks at 2048 points (179 MB per pass: 31 M evals/s) and
// falls to 24-26 M at 8192+; hess_l peaks at 8192 points (246 MB: 65 M) and falls to 44 M at 16384.  Batches of more than 1.5x
// 256 MB are therefore cut into equal passes of at most 256 MB that reuse ONE raw buffer (smaller passes lose more to the
// shorter launches than the cache gives back); every evaluation point is independent, results unchanged.
static int64_t points_per_pass(int64_t batch, int64_t doubles_per_point) {
  const char* env = mpx_knob(MPX_K_ASM_PASS_MB);
  const int64_t budget = env ? atoll(env) * 1000000 : 256000000;  // 0: one pass
  if (budget <= 0) return batch;
  const int64_t fit = std::max<int64_t>(budget / (8 * std::max<int64_t>(doubles_per_point, 1)), 256);
  if (batch <= fit + fit / 2) return batch;
  const int64_t n_pass = (batch + fit - 1) / fit;
  return ((batch + n_pass - 1) / n_pass + 3) / 4 * 4;
}

extern "C" int mpx_get_assembled_plan(const mpx_ctx* c, int32_t* fused_lanes, int32_t* hess_lane_groups, int32_t* first_order_lane_groups) {
  if (!c || c->kind != 1) return MPX_ERR_INVALID;
  const mpx_asm_state* a = c->has_device ? c->assembled : nullptr;  // (a context without a device has no kernels)
  if (fused_lanes) *fused_lanes = a ? a->fuse_nt : 0;
  if (hess_lane_groups) *hess_lane_groups = a && a->lanes[0].fn ? a->lanes[0].groups : 0;
  if (first_order_lane_groups) *first_order_lane_groups = a && a->lanes[1].fn ? a->lanes[1].groups : 0;
  return MPX_OK;
}

// Device-pointer evaluation of an assembled context (called from eval_core in mpx_host.cpp after the
// argument checks).  All pointers are device pointers.
int mpx_asm_eval_device(mpx_ctx* c, int mask, int64_t batch, const double* z, const double* lam_g, const double* sigma, double* f, double* g,
                        double* grad_f, double* jac_val, double* hess_val) {
  mpx_asm_state* a = c->assembled;
  if (mask & (MPX_BOUNDARY_ONLY | MPX_JAC_VARIABLE_ONLY)) return fail(c, MPX_ERR_UNSUPPORTED, "mask bit not available on assembled contexts");
  int rc;
  if ((mask & (MPX_F | MPX_G | MPX_GRAD | MPX_JAC)) == (MPX_F | MPX_G | MPX_GRAD | MPX_JAC) && use_lanes(a, 1, batch)) {
    hipEvent_t pe = nullptr;  // (the whole first-order pass with one lane per evaluation point; partial masks keep the fused kernels)
    if ((rc = prof_begin(c, &pe))) return rc;
    double* outp[4] = {f, g, grad_f, jac_val};
    if ((rc = launch_lanes(c, 1, batch, z, nullptr, nullptr, outp))) return rc;
    if ((rc = prof_end(c, pe))) return rc;
    if (c->profile) ++c->prof_launches;
  } else if ((mask & (MPX_F | MPX_G | MPX_GRAD | MPX_JAC)) && use_fused(a, (mask & (MPX_GRAD | MPX_JAC)) ? MPX_MODE_FGJ : MPX_MODE_FG, batch)) {
    const int mode = (mask & (MPX_GRAD | MPX_JAC)) ? MPX_MODE_FGJ : MPX_MODE_FG;
    hipEvent_t pe = nullptr;
    if ((rc = prof_begin(c, &pe))) return rc;
    double* outp[4] = {(mask & MPX_F) ? f : nullptr, (mask & MPX_G) ? g : nullptr, (mask & MPX_GRAD) ? grad_f : nullptr, (mask & MPX_JAC) ? jac_val : nullptr};
    const int64_t stride[4] = {1, c->n_g, c->n_z, c->nnz_j};
    if ((rc = launch_fused(c, mode, batch, z, nullptr, nullptr, outp, stride))) return rc;
    if ((rc = prof_end(c, pe))) return rc;
    if (c->profile) ++c->prof_launches;
  } else if (mask & (MPX_F | MPX_G | MPX_GRAD | MPX_JAC)) {
    const int mode = (mask & (MPX_GRAD | MPX_JAC)) ? MPX_MODE_FGJ : MPX_MODE_FG;
    const int64_t per = points_per_pass(batch, a->raw_n + c->n_z + ((mask & MPX_G) ? c->n_g : 0) + ((mask & MPX_GRAD) ? c->n_z : 0) + ((mask & MPX_JAC) ? c->nnz_j : 0));
    if ((rc = reserve(c, a->raw, (size_t)(per * a->raw_n)))) return rc;
    hipEvent_t pe = nullptr;  // mpx_profile: one bracket per evaluation (point kernels + gather of every pass), counted as one launch
    if ((rc = prof_begin(c, &pe))) return rc;
    for (int64_t b0 = 0; b0 < batch; b0 += per) {
      const int64_t nb = std::min(per, batch - b0);
      const double* zb = z + b0 * c->n_z;
      if ((rc = launch_points(c, mode, nb, zb, nullptr, nullptr))) return rc;
      const int64_t begin[4] = {0, 1, 1 + c->n_g, 1 + c->n_g + c->n_z};
      double* outp[4] = {(mask & MPX_F) ? f + b0 : nullptr, (mask & MPX_G) ? g + b0 * c->n_g : nullptr, (mask & MPX_GRAD) ? grad_f + b0 * c->n_z : nullptr,
                         (mask & MPX_JAC) ? jac_val + b0 * c->nnz_j : nullptr};
      const int64_t stride[4] = {1, c->n_g, c->n_z, c->nnz_j};
      if ((rc = launch_gather(c, a->fgj, nb, zb, a->raw_n, 4, begin, outp, stride))) return rc;
    }
    if ((rc = prof_end(c, pe))) return rc;
    if (c->profile) ++c->prof_launches;
  }
  if (mask & MPX_HESS) {
    hipEvent_t pe = nullptr;  // (mpx_profile: one bracket per evaluation, counted as one launch)
    if ((rc = prof_begin(c, &pe))) return rc;
    int64_t done = 0;
    if (use_lanes(a, 0, batch)) {
      double* outp[4] = {hess_val, nullptr, nullptr, nullptr};
      if ((rc = launch_lanes(c, 0, batch, z, lam_g, sigma, outp))) return rc;
      done = batch;
    }
    if (done < batch && use_fused(a, MPX_MODE_HESS, batch - done)) {
      double* outp[4] = {hess_val + done * c->nnz_h, nullptr, nullptr, nullptr};
      const int64_t stride[4] = {c->nnz_h, 0, 0, 0};
      if ((rc = launch_fused(c, MPX_MODE_HESS, batch - done, z + done * c->n_z, lam_g + done * c->n_g, sigma + done, outp, stride))) return rc;
    } else if (done < batch) {
      const int64_t per = points_per_pass(batch - done, a->rawh_n + c->n_z + c->n_g + 1 + c->nnz_h);
      if ((rc = reserve(c, a->raw, (size_t)(per * a->rawh_n)))) return rc;
      for (int64_t b0 = done; b0 < batch; b0 += per) {
        const int64_t nb = std::min(per, batch - b0);
        const double* zb = z + b0 * c->n_z;
        if ((rc = launch_points(c, MPX_MODE_HESS, nb, zb, lam_g + b0 * c->n_g, sigma + b0))) return rc;
        const int64_t begin[1] = {0};
        double* outp[1] = {hess_val + b0 * c->nnz_h};
        const int64_t stride[1] = {c->nnz_h};
        if ((rc = launch_gather(c, a->hess, nb, zb, a->rawh_n, 1, begin, outp, stride))) return rc;
      }
    }
    if ((rc = prof_end(c, pe))) return rc;
    if (c->profile) ++c->prof_launches;
  }
  return MPX_OK;
}
