// mpx_internal.h -- state shared by the translation units of libmpx (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <map>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "mpx.h"
#include "mpx_device.h"

// ---- A/B and test knobs of the EVALUATION path (environment variables) ---------------------------------------------------------------
// Read once per process, at the first evaluation -- not with getenv() on every call (~10 look-ups of ~50 ns on the
// single-evaluation latency path, and getenv is not safe against a concurrent setenv) -- unless the process asks for the old behaviour:
// MPX_ENV_DYNAMIC=1 in the environment when the library is first used, or mpx_env_dynamic(1) (the test suite and the A/B tools switch
// knobs inside one process).  Knobs read at context CREATION (mpx_layout.cpp, load_device, ...) are plain getenv() calls.
enum MpxKnob {
  MPX_K_BPB, MPX_K_NO_LIGHT, MPX_K_LIGHT_LONG_SPANS, MPX_K_NO_PACKED_G, MPX_K_NO_PHASE_MERGE, MPX_K_LIGHT_DEBUG, MPX_K_LIGHT_PER_CU,
  MPX_K_RESIDENT, MPX_K_NO_RESIDENT, MPX_K_GRADL_GENERIC, MPX_K_NO_FUSE, MPX_K_FUSE_MIN_BATCH, MPX_K_FUSE_PT_STAMPS, MPX_K_NO_LANES,
  MPX_K_LANES_MIN_BATCH, MPX_K_LANES_ORDER, MPX_K_ASM_PASS_MB, MPX_K_EA_GENERIC, MPX_K_EA_DEBUG, MPX_K_GRADL_BPB, MPX_K_COUNT
};
const char* mpx_knob(MpxKnob k);  // the variable's value, nullptr if unset

namespace mpxi {

std::string& create_error();  // thread-local message of a failed mpx_create*

struct Entry4 {
  int32_t a, b, c, d;
};
struct PhaseStruct {
  int nc = 0, ntc = 0, diff_u = 0, midu = 0, du_cont = 0;
  std::vector<Entry4> jv, hn, hc, th;
  std::vector<std::pair<int32_t, int32_t>> mg;
  struct TJ {
    int32_t row, kind, comp;
  };
  std::vector<TJ> tj;
  // layout
  int64_t z_off = 0, g_off_F = 0, g_off_C = 0, g_off_DU = 0, g_off_mU = 0, g_off_dU = 0, g_off_TC = 0;
  int64_t jac_TC = 0;
  int tile_first = 0, tile_count = 0;
};

struct DegTable {
  int deg = 0;
  std::vector<double> roots, D, Cmid, w, tk, Dmid, tkm;
  double *d_D = nullptr, *d_Cmid = nullptr, *d_tk = nullptr, *d_Dmid = nullptr, *d_tkm = nullptr, *d_w = nullptr;
  double *d_DT = nullptr, *d_CT = nullptr;  // degrees above mpx_ctx::stream_above: D and C_mid transposed, what node_body's lanes stream
};

struct Bucket {
  int phase = 0, deg = 0, dt = 0;  // dt: index into degree tables
  std::vector<int32_t> node_i, node_sk;
  int tile_first = 0, tile_count = 0;  // global tile ids
  int abs_cap = 0;                     // > 0: absorbing bucket (MpxNodeArgs::abs_cap), row capacity of its LDS span buffer
  int abs_slots = 0;                   // row slots of that buffer (g rows + grad_f rows per node)
  int64_t abs_lds_static = 0;          // static LDS of the bucket's node kernel (the span rows come on top as dynamic LDS)
  int32_t *d_node_i = nullptr, *d_node_sk = nullptr;
  hipFunction_t fn[3] = {nullptr, nullptr, nullptr};
  hipFunction_t fn_gradl = nullptr;  // mpx_node_gradl_<phase>_<deg> (nlp_grad)
  hipFunction_t fn_light[2] = {nullptr, nullptr}, fn_light_small[2] = {nullptr, nullptr};  // mpx_light_fg / _fgq _<phase>_<deg>: light passes on the matrix cores (12 < deg <= 31)
};

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;
};

}  // namespace mpxi

struct mpx_asm_state;  // mpx_assembly.cpp
using namespace mpxi;

struct mpx_ctx {
  // 0: structured collocation context (mpx_create); 1: assembled context (mpx_create_assembled)
  int kind = 0;
  mpx_asm_state* assembled = nullptr;
  std::string err;
  std::string notes;  // decisions worth knowing that are not errors (mpx_get_notes): fallbacks of the layout planner
  // problem
  int n_phases = 0, nx = 0, nu = 0, na = 0, S = 0, scheme = 0, device = 0;
  double tau0 = -1, tau1 = 1;
  std::vector<int32_t> orders, seg_start, links;
  std::vector<PhaseStruct> ph;
  int64_t N = 0, n_zp = 0, n_z = 0, n_g = 0, n_p = 0, nnz_j = 0, nnz_h = 0;
  std::vector<DegTable> degs;
  std::vector<Bucket> buckets;
  std::vector<MpxTile> tiles;  // global, phase-major, bucket-major
  std::vector<double> compW;
  std::vector<int32_t> jrow, jcol, hrow, hcol;
  std::vector<uint8_t> jac_var;  // per entry of jac_g (native order): 1 = depends on (z, p), 0 = a constant of the grid (mpx_pattern_jac_variable)
  // linear rows
  std::vector<int64_t> lin_ptr, lin_idx, lin_row;
  std::vector<double> lin_coef;
  int64_t lin_jac = 0, jac_tiles_end = 0;
  std::vector<int64_t> mg_dst, hc_dst, th_dst;
  std::vector<int32_t> mg_off, hc_off, th_off;
  int nred = 1;
  // linear rows transposed (nlp_grad: J^T lam_g of the control-slope continuity and event rows), CSC over the columns they touch
  std::vector<int64_t> lt_ptr, lt_col, lt_row;
  std::vector<double> lt_coef;
  // device
  bool has_device = false;
  int stream_above = MPX_TABLES_STREAM_ABOVE;  // degrees above it stream their tables (mpx_device.h; MPX_TABLES_STREAM_ABOVE at creation; checked against the code object)
  bool time_dep = true;  // some node function uses the node time (mpx_time_dependent of the code object; true when the symbol is absent): else no prefix sums of the widths
  hipModule_t module = nullptr;
  hipFunction_t fn_bound[3] = {nullptr, nullptr, nullptr};
  // all phases of a single-degree grid in one launch (n_phases > 1; mpx_node_<mode>_all_<deg>, mpx_lightlow[s]_<fg|fgq>_all_<deg>);
  // nullptr: the code object has none (one degree per phase only, or generated before round 5) -- one launch per phase then
  hipFunction_t fn_node_all[3] = {nullptr, nullptr, nullptr};
  hipFunction_t fn_lightlow_all[2] = {nullptr, nullptr}, fn_lightlows_all[2] = {nullptr, nullptr};
  hipStream_t stream = nullptr;
  MpxTile* d_tiles = nullptr;
  double* d_Wnode = nullptr;
  int32_t* d_seg_start = nullptr;
  int64_t *d_lin_ptr = nullptr, *d_lin_idx = nullptr, *d_lin_row = nullptr, *d_mg_dst = nullptr,
          *d_hc_dst = nullptr, *d_th_dst = nullptr;
  double* d_lin_coef = nullptr;
  DevBuf<double> partial, wcum, st_z, st_p, st_lam, st_sig, st_f, st_g, st_grad, st_jac, st_hess;
  // nlp_grad (mpx_eval_grad_gamma*): finishing kernel, staging of the node pass (MpxGradlArgs::halo / pnode), host-path outputs,
  // and the generic J^T lam route (assembled contexts; MPX_GRADL_GENERIC=1): scratch grad_f / jac_val + compressed-column tables
  hipFunction_t fn_gradl_fin = nullptr;
  DevBuf<double> gl_halo, gl_pnode, st_ggx, st_ggp, gl_grad, gl_jac;  // (gl_pnode: the per-segment sums of nlp_grad, MpxGradlArgs::pseg)
  std::vector<int32_t> gl_halo_seg, gl_halo_off;  // nlp_grad: segments whose column-0 sums go through `halo`, per phase (MpxGradlFinArgs)
  int32_t* d_gl_halo_seg = nullptr;
  // light passes on the matrix cores (mpx_light_*, mpx_kernels.h: light_body): grids with ONE high degree (12 < P <= 31) and otherwise
  // degrees <= 12.  Groups of up to 16 high-degree segments + the low-degree segments between them (the same for every phase)
  struct LightPlan {
    bool ok = false, low = false;           // low: single-degree grid of degree <= 12 (light_low_body: spans of `own` nodes)
    bool high = false;                      // high: single-degree grid of degree >= 32 (light_high_body: workgroup = segment x 16 evaluation points; n_low_chunks = S slots per phase, span_cap = padded K)
    int deg = 0, dt = -1, first_node = 0, span_cap = 0, own = 0, n_low_groups = 0, n_low_chunks = 0;
    std::vector<MpxLightGroup> groups;
    std::vector<MpxLightForeign> foreign;
    std::vector<double> ftab;               // D and C_mid of the low degrees, concatenated
    std::vector<int32_t> fD_off, fC_off;    // by degree-table index (-1: the high degree)
  } lplan;
  MpxLightGroup* d_lgroups = nullptr;
  MpxLightForeign* d_lforeign = nullptr;
  double* d_lftab = nullptr;
  int64_t *d_lt_ptr = nullptr, *d_lt_col = nullptr, *d_lt_row = nullptr, *d_colind_j = nullptr;
  double* d_lt_coef = nullptr;
  int32_t* d_jrow = nullptr;
  // mixed-degree phases: packed staging of g / grad_f (MpxIO::gtmp) and the maps of mpx_unpack_kernel
  bool g_packed = false;
  int64_t gtmp_n = 0;
  std::vector<int64_t> gmap, qmap;  // row of g / entry of grad_f -> index in the staging block, -1: written elsewhere
  int64_t *d_gmap = nullptr, *d_qmap = nullptr;
  DevBuf<double> gtmp;
  // ... without the unpack pass where the grid allows it (build_layout: absorbing buckets, MpxTile::span_*)
  bool absorb = false;
  std::vector<int32_t> abs_fpos, abs_fn;
  std::vector<int64_t> abs_fstage;
  int32_t *d_abs_fpos = nullptr, *d_abs_fn = nullptr;
  int64_t* d_abs_fstage = nullptr;
  DevBuf<double> ea_scratch;  // mpx_equal_area_widths_device: cumulative areas + segment boundaries
  size_t ea_lds_allowed = 0;  //   dynamic LDS limit already raised for the kernel on this context's device
  long long* ea_dbg = nullptr;  //   MPX_EA_DEBUG phase stamps (page-locked)
  // MPX_CCS_ORDER: scratch in native order + device copies of the permutations (built on first use)
  DevBuf<double> ccs_j, ccs_h;
  int64_t *d_perm_j = nullptr, *d_perm_h = nullptr;
  int64_t *d_var_dst = nullptr, *d_var_src = nullptr;  // MPX_JAC_VARIABLE_ONLY | MPX_CCS_ORDER: compressed-column positions of the variable entries and where they come from
  int64_t n_var_j = -1;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  double* mid_resid_out = nullptr;  // mpx_set_mid_resid_output: device array the MPX_MID_RESID passes write
  int64_t tile_begin = 0, tile_end = 0;
  int run_boundary = 1;
  // host-side sizes of every tile's value blocks (doubles): jac, hess, packed g / grad_f staging
  std::vector<int64_t> tile_jac_size, tile_hess_size, tile_g_size;
  // mixed-degree grids: node-ordered tiles of the hess_l pass (MpxHTile, mpx_device.h)
  bool hess_by_node = false;
  std::vector<MpxHTile> htiles;                     // all phases, phase-major
  std::vector<int32_t> ph_htile_first, ph_htile_count;
  std::vector<int32_t> node_seg;
  std::vector<double> node_tk;
  MpxHTile* d_htiles = nullptr;
  int32_t* d_node_seg = nullptr;
  double* d_node_tk = nullptr;
  hipFunction_t fn_hessn[MPX_MAX_PHASES] = {};
  std::vector<int64_t> shard_cuts_h;                // [world + 1] ranges of htiles (per phase the same fractions)
  // segment sharding (mpx_shard_*): this context evaluates the tiles [shard_cuts[rank], shard_cuts[rank + 1]) of every point
  int shard_world = 1, shard_rank = 0;
  std::vector<int64_t> shard_cuts;                  // [world + 1]
  std::vector<MpxShardEnt> shard_ent[2];            // pass 0: f/g/grad_f/jac_g, pass 1: hess_l; entries of ALL ranks
  std::vector<int32_t> shard_ent_first[2];          // [world + 1] first entry of every rank
  int64_t shard_len[2] = {0, 0};                    // padded per-rank, per-point length of the exchange buffer (doubles)
  int64_t shard_len_part = 0;                       // the same for the owner-resident exchange (tile partials only)
  MpxShardEnt* d_shard_ent[2] = {nullptr, nullptr};
  // page-locked host ranges this context knows (mpx_host_alloc / mpx_host_register) with their device-side aliases: a single
  // evaluation whose arrays all lie in such ranges runs zero-copy (kernels read z and write the results straight over PCIe)
  struct PinRange {
    char* base;
    size_t bytes;
    char* dev;
    bool owned;  // hipHostMalloc'ed by mpx_host_alloc (else registered caller memory)
  };
  std::vector<PinRange> pins;
  double* h_scratch = nullptr;  // page-locked scalars of the zero-copy path: [f (B) | sigma (B)]
  double* h_scratch_dev = nullptr;
  size_t h_scratch_cap = 0;
  unsigned long long *h_flag = nullptr, *h_flag_dev = nullptr;  // completion flag of the zero-copy path (page-locked, GPU-visible)
  unsigned long long flag_seq = 0;
  // host path: widths of the previous mpx_eval (IPOPT never changes p between oracle calls, so the
  // upload and the prefix-sum launch are skipped while p is unchanged)
  std::vector<double> last_p;
  bool wcum_valid = false;
  // what the device buffer `wcum` holds: the exclusive prefix sums of the width vectors at device address wcum_p (wcum_batch
  // vectors, shared or per point) for the phases in wcum_phases.  Written by every prefix launch (all phases) and by the fast
  // equal-area kernel (the phase it updated); MPX_WIDTHS_UNCHANGED is honoured only when it describes the call's own p -- otherwise
  // the prefix kernel is launched after all (a generic equal-area update, a larger grid, another array: never stale sums)
  const double* wcum_p = nullptr;
  int64_t wcum_batch = 0;
  int wcum_ppp = 0;
  uint32_t wcum_phases = 0;
  // resident kernel of single evaluations (mpx_kernels.h: resident_loop; MpxMailbox / MpxResidentArgs in mpx_device.h)
  struct Resident {
    hipFunction_t fn = nullptr;
    hipStream_t stream = nullptr;
    MpxMailbox *box = nullptr, *box_dev = nullptr;  // page-locked, GPU-visible
    MpxNodeArgs* d_buckets = nullptr;
    int32_t* d_tile_bucket = nullptr;
    MpxBoundArgs* d_bound = nullptr;
    MpxResRequest* d_slots = nullptr;
    unsigned long long *d_seq = nullptr, *d_sync = nullptr;
    unsigned long long seq = 0, word = 0;  // number of the last request; the word it was sent as
    MpxResRequest slot[MPX_RES_SLOTS];     // host copies of the argument slots (what the device holds)
    int n_slots = 0, next_slot = 0;
    bool ok = false, launched = false;
    long long n_launches = 0, n_requests = 0;
  } res;
  // launch-geometry selection for large batches (run_mode): evaluation points per workgroup, measured once per output placement
  struct GeomTune {
    const void* key = nullptr;  // dominant output array of the pass
    int64_t B = 0;
    int mode = 0, stage = 0, best = 1, uses = 0;
    int sig = 0;  // which outputs the pass writes (f 1, g 2, grad_f 4, jac 8): the same array is written by passes of different weight
    int cand[2] = {1, 1};
    hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    uint64_t last_use = 0;
  };
  std::vector<GeomTune> tune;
  uint64_t tune_clock = 0;
  // per-kernel profiling
  int profile = 0;
  std::vector<hipEvent_t> prof_ev;  // pairs
  size_t prof_used = 0;
  int64_t prof_launches = 0;
};

namespace mpxi {

inline int fail(mpx_ctx* c, int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c)
    c->err = buf;
  else
    create_error() = buf;
  return code;
}

#define HIPCHK(ctx, call)                                                                        \
  do {                                                                                           \
    hipError_t e_ = (call);                                                                      \
    if (e_ != hipSuccess)                                                                        \
      return fail(ctx, MPX_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

// Doubles-per-point of the partial-sum buffer, in slots: the tiles, or -- light passes of single-degree low-degree grids -- one slot per
// 64-node chunk of every phase (mpx_kernels.h: light_low_body), whichever is more
inline int64_t partial_slots(const mpx_ctx* c) {
  const int64_t low = c->lplan.ok && (c->lplan.low || c->lplan.high) ? (int64_t)c->n_phases * c->lplan.n_low_chunks : 0;
  return std::max<int64_t>((int64_t)c->tiles.size(), low);
}

// MPX_POISON_ALLOC=1 (debugging aid, round 6): every scratch buffer the library allocates on the device starts out as 0x7f bytes
// (doubles of 1.4e306, indices of 2.1e9) instead of whatever the block held: a kernel that reads scratch memory nobody wrote shows up
// as a wrong result or a fault in the test suite, not once in some thousand placements.
inline bool poison_alloc() {
  static const bool on = getenv("MPX_POISON_ALLOC") != nullptr;
  return on;
}
inline int poison_byte() {  // MPX_POISON_ALLOC=1: 0x7f; MPX_POISON_ALLOC=ff (any two hex digits): that byte (0xff: NaN doubles, indices of -1)
  static const int b = [] {
    const char* e = getenv("MPX_POISON_ALLOC");
    return e && strlen(e) == 2 ? (int)strtol(e, nullptr, 16) & 0xff : 0x7f;
  }();
  return b;
}

// MPX_GUARD_ALLOC=1 | head (debugging aid, round 6): every device buffer of the library is the LAST bytes (head: the first bytes) of its own
// allocation of a multiple of 2 MB, so that a kernel running past the end (before the start) of one of the library's tables or scratch buffers
// leaves the allocation -- a memory access fault -- instead of reading its neighbour.  (The tail pointer keeps 16-byte alignment: an overrun of
// less than 16 bytes can stay inside.)  tools/r6_tail_guard.py does the same for the caller's arrays.
struct GuardAlloc {
  std::mutex mu;
  std::map<void*, void*> base_of;  // pointer handed out -> allocation
  int mode = 0;                    // 0 off, 1 tail, 2 head
  GuardAlloc() {
    const char* e = getenv("MPX_GUARD_ALLOC");
    mode = !e ? 0 : (!strcmp(e, "head") ? 2 : 1);
  }
};
inline GuardAlloc& guard_alloc() {
  static GuardAlloc g;
  return g;
}
inline hipError_t dev_malloc(void** p, size_t bytes) {
  GuardAlloc& g = guard_alloc();
  if (!g.mode) return hipMalloc(p, bytes);
  const size_t gran = size_t(2) << 20, nb = (std::max<size_t>(bytes, 1) + 15) & ~size_t(15), size = (nb + gran - 1) / gran * gran;
  void* base = nullptr;
  hipError_t e = hipMalloc(&base, size);
  if (e != hipSuccess) return e;
  *p = g.mode == 2 ? base : static_cast<char*>(base) + (size - nb);
  std::lock_guard<std::mutex> lk(g.mu);
  g.base_of[*p] = base;
  return hipSuccess;
}
inline hipError_t dev_free(void* p) {
  GuardAlloc& g = guard_alloc();
  if (g.mode) {
    std::lock_guard<std::mutex> lk(g.mu);
    auto it = g.base_of.find(p);
    if (it != g.base_of.end()) {
      p = it->second;
      g.base_of.erase(it);
    }
  }
  return hipFree(p);
}

template <class T>
inline int upload(mpx_ctx* c, T** dst, const std::vector<T>& v) {
  size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
  HIPCHK(c, dev_malloc((void**)dst, bytes));
  if (!v.empty()) HIPCHK(c, hipMemcpy(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  else HIPCHK(c, hipMemset(*dst, 0, bytes));  // (an empty list is one zeroed element on the device, never uninitialised memory)
  return MPX_OK;
}

template <class T>
inline int reserve(mpx_ctx* c, DevBuf<T>& b, size_t n) {
  if (n <= b.cap) return MPX_OK;
  if (b.p) HIPCHK(c, dev_free(b.p));
  b.p = nullptr;
  b.cap = 0;
  HIPCHK(c, dev_malloc((void**)&b.p, n * sizeof(T)));
  if (poison_alloc()) HIPCHK(c, hipMemset(b.p, poison_byte(), n * sizeof(T)));
  b.cap = n;
  return MPX_OK;
}

}  // namespace mpxi

namespace mpxi {
// Per-kernel profiling (mpx_profile): a pair of events on the context stream around the dominant launches of one pass.
inline int prof_begin(mpx_ctx* c, hipEvent_t* end_event) {
  *end_event = nullptr;
  if (!c->profile) return MPX_OK;
  if (c->prof_used + 2 > c->prof_ev.size()) {
    for (int k = 0; k < 2; ++k) {
      hipEvent_t e;
      HIPCHK(c, hipEventCreate(&e));
      c->prof_ev.push_back(e);
    }
  }
  hipEvent_t begin = c->prof_ev[c->prof_used];
  *end_event = c->prof_ev[c->prof_used + 1];
  c->prof_used += 2;
  HIPCHK(c, hipEventRecord(begin, c->stream));
  return MPX_OK;
}
inline int prof_end(mpx_ctx* c, hipEvent_t end_event) {
  if (end_event) HIPCHK(c, hipEventRecord(end_event, c->stream));
  return MPX_OK;
}
// (a re-allocation loses the prefix sums the buffer held)
inline int reserve_wcum(mpx_ctx* c, size_t n) {
  const double* before = c->wcum.p;
  const int rc = reserve(c, c->wcum, n);
  if (c->wcum.p != before) c->wcum_phases = 0;
  return rc;
}
inline uint32_t all_phases(const mpx_ctx* c) { return c->n_phases >= 32 ? ~0u : ((1u << c->n_phases) - 1u); }

// mpx_layout.cpp -- the planner of mpx_create: structure words -> phases; per-degree tables; tiles, index maps, patterns, light plans
int parse_structure(mpx_ctx* c, const int32_t* s, int64_t len);
int build_tables(mpx_ctx* c);
int deg_index(const mpx_ctx* c, int d);
int build_layout(mpx_ctx* c);

}  // namespace mpxi

// mpx_assembly.cpp
void mpx_asm_release(mpx_ctx* c);
int mpx_asm_eval_device(mpx_ctx* c, int mask, int64_t batch, const double* z, const double* lam_g, const double* sigma, double* f, double* g,
                        double* grad_f, double* jac_val, double* hess_val);
